"""does the hardware queue priority of the engine's streams change the composed step?  The step is throughput-bound (the
MIOpen convolutions of the image branch fill the chip while the main chain is a string of small kernels).
Usage: JM_SIDE_PRIO=... python tools/stream_prio_probe.py [main_priority]   (fresh process per setting)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

main_prio = int(sys.argv[1]) if len(sys.argv) > 1 else None
dev = torch.device("cuda:0")
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "n/a")
st = bench.make_detect_state(8, 1236, dev)
st["engine"].overlap = True
st["prefetch"] = True
ms = torch.cuda.Stream(priority=main_prio) if main_prio is not None else torch.cuda.current_stream()
with torch.cuda.stream(ms):
    for _ in range(5):
        bench.detect_step(st)
    torch.cuda.synchronize()
    res = []
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(20):
            bench.detect_step(st)
        torch.cuda.synchronize()
        res.append((time.perf_counter() - t0) / 20 * 1e3)
print(f"JM_SIDE_PRIO={os.environ.get('JM_SIDE_PRIO', '')!r} main={main_prio}: ms/step {['%.3f' % r for r in res]} -> {8e3 / min(res):.1f} frames/s")
