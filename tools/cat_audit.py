"""which torch.cat / .contiguous() / copy_ calls does one composed step make (shapes + call sites)?"""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
st = bench.make_detect_state(8, 1236, dev)
for _ in range(2): bench.detect_step(st)
torch.cuda.synchronize()
log = []
def site():
    for f in reversed(traceback.extract_stack()[:-2]):
        if "jmodt_amd" in f.filename: return f"{os.path.basename(f.filename)}:{f.lineno}"
    return "?"
_cat, _contig = torch.cat, torch.Tensor.contiguous
def cat(ts, *a, **k):
    log.append(("cat", [tuple(t.shape) for t in ts], site())); return _cat(ts, *a, **k)
def contig(self, *a, **k):
    if not self.is_contiguous(*a, **k) if not a and not k else not self.is_contiguous(**k):
        log.append(("contiguous", tuple(self.shape), site()))
    return _contig(self, *a, **k)
torch.cat = cat; torch.Tensor.contiguous = contig
bench.detect_step(st); torch.cuda.synchronize()
torch.cat = _cat; torch.Tensor.contiguous = _contig
for r in log: print(r)
