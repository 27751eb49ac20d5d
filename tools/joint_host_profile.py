"""where does the HOST spend the joint-mode step's enqueue time?  cProfile over N steps + the wall-clock of enqueue vs device:
    gpurun -- 'python tools/joint_host_profile.py 10'"""
import cProfile, os, pstats, sys, io, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10
st = bench.make_joint_state(4, 1234, torch.device("cuda:0"))
for _ in range(4):
    bench.train_step(st, None)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    bench.train_step(st, None)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"un-profiled: host enqueue {(t1 - t0) / N * 1e3:.2f} ms per step, step {(t2 - t0) / N * 1e3:.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bench.train_step(st, None)
pr.disable()
torch.cuda.synchronize()
for key, n in (("tottime", 40), ("cumulative", 45)):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(n)
    txt = s.getvalue()
    print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)
