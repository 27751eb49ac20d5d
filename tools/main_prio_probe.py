"""the main chain on a HIGH-priority stream (the image / FPS / detections side streams at normal priority), with the Winograd kernel
persistent or one workgroup per item (tools build: JM_WN_GRID): do the main chain's small launches then get CUs as the convolution's
workgroups retire?    gpurun -- 'JM_WN_GRID=100000 python tools/ab_lib.py tools/bin/libjmodt_hip_tools.so tools/main_prio_probe.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
streams = {"default stream": None, "high-priority stream": torch.cuda.Stream(device=dev, priority=-1), "normal-priority stream": torch.cuda.Stream(device=dev, priority=0)}
for tag, s in list(streams.items()) + [("default stream", None)]:
    def steps(n):
        for _ in range(n):
            if s is None:
                bench.detect_step(st)
            else:
                with torch.cuda.stream(s):
                    bench.detect_step(st)
    steps(6); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps(40); torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 40 * 1e3
    print(f"JM_WN_GRID={os.environ.get('JM_WN_GRID')} main chain on the {tag:24s}: {ms:7.3f} ms/step {8 / ms * 1e3:7.1f} frames/s", flush=True)
    with torch.no_grad():
        if s is None:
            st["engine"](st["xyz"], st["image"], st["pts_xy"])
        else:
            with torch.cuda.stream(s):
                st["engine"](st["xyz"], st["image"], st["pts_xy"])
    torch.cuda.synchronize()
