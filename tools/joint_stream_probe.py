"""the joint-mode step with its MAIN stream = torch's default (legacy NULL) stream against a stream of its own: graph launches on
the NULL stream?   gpurun -- 'JM_JOINT_GRAPHED=... python tools/joint_stream_probe.py'"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
st = bench.make_joint_state(4, 1234, dev)
def run(n, stream):
    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        for _ in range(4):
            bench.train_step(st, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            bench.train_step(st, None)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
if os.environ.get("JM_STEP_TRACE"):
    for _ in range(5):
        bench.train_step(st, None)
    sys.exit(0)
for name, s in (("own stream", torch.cuda.Stream()), ("default stream", None)):
    if name == "own stream":
        s.wait_stream(torch.cuda.current_stream())
    h, t = run(10, s)
    print(f"{os.environ.get('JM_JOINT_GRAPHED', 'all'):40s} {name:15s}: host {h:.2f} ms, step {t:.2f} ms = {4e3 / t:.1f} frames/s", flush=True)
