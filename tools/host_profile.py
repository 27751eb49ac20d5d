"""where does the HOST spend its enqueue time?  cProfile over N composed steps (the GPU runs behind; no synchronisation inside):
    gpurun -- 'python tools/host_profile.py 30'"""
import cProfile, os, pstats, sys, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
st = bench.make_detect_state(8, 1236, torch.device("cuda:0"))
for _ in range(5):
    bench.detect_step(st)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    bench.detect_step(st)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
ps = pstats.Stats(pr, stream=s).sort_stats("tottime")
ps.print_stats(28)
txt = s.getvalue()
print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
txt = s.getvalue()
print(txt[txt.index("ncalls"):] if "ncalls" in txt else txt)
