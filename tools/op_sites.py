"""which aten ops does one composed detect+affinity step issue, and from which line of jmodt_amd?  A TorchDispatchMode logs every
op with the innermost jmodt_amd / bench.py frame (the profiler's python stacks come back empty on this build).
    gpurun -- 'python tools/op_sites.py [detect|joint]'"""
import collections, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode
import bench

VIEW = {"view", "_unsafe_view", "transpose", "t", "expand", "slice", "select", "unsqueeze", "squeeze", "permute", "detach", "alias", "as_strided",
        "reshape", "_reshape_alias", "unbind", "split", "split_with_sizes", "narrow", "empty", "empty_like", "empty_strided", "new_empty",
        "new_empty_strided", "is_pinned", "record_stream", "size", "stride", "numel", "dim", "lift_fresh", "is_contiguous", "sym_size", "sym_stride",
        "sym_numel", "sym_storage_offset", "stride", "_local_scalar_dense", "resize_"}

class Sites(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.count = collections.Counter()
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.overloadpacket.__name__
        if name not in VIEW:
            site = "?"
            for fr in reversed(traceback.extract_stack(limit=30)):
                if ("jmodt_amd" in fr.filename or fr.filename.endswith("bench.py")) and "op_sites" not in fr.filename:
                    site = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.name}"
                    break
            shp = next((tuple(a.shape) for a in args if isinstance(a, torch.Tensor)), ())
            self.count[(name, site, shp)] += 1
        return func(*args, **(kwargs or {}))

what = sys.argv[1] if len(sys.argv) > 1 else "detect"
dev = torch.device("cuda:0")
if what == "detect":
    st = bench.make_detect_state(8, 1236, dev)
    step = lambda: bench.detect_step(st)
else:
    st = bench.make_joint_state(4, 1234, dev)
    step = lambda: bench.train_step(st, None)
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 2
with Sites() as s:
    for _ in range(N):
        step()
torch.cuda.synchronize()
tot = sum(s.count.values()) / N
print(f"{tot:.0f} non-view aten ops per step")
for (name, site, shp), c in sorted(s.count.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    print(f"  x{c / N:4.1f}  {name:28s} {str(shp):26s} {site}")
