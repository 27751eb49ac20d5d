"""the batched link head (jm_affinity_forward_batched without the start/end head) at the detector's shapes: time, fraction of
the fp32 MFMA peak, error against float64.  Usage: PYTHONPATH=. python tools/affinity_link_time.py"""
import torch

from jmodt_amd.ops.affinity import make_affinity_mlp, pairwise_affinity_batched

torch.manual_seed(0)
link = make_affinity_mlp().cuda().eval()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


for nb, P in ((8, 128), (8, 256), (8, 64), (1, 128), (2, 128), (4, 128)):
    pf = torch.relu(torch.randn(nb, P, 512, device="cuda"))
    df = torch.relu(torch.randn(nb, P, 512, device="cuda"))
    fl = nb * P * P * (2 * 512 * 512 * 2 + 2 * 512)
    t = timeit(lambda: pairwise_affinity_batched(pf, df, link, None, return_raw=True))
    raw = pairwise_affinity_batched(pf, df, link, None, return_raw=True)[-1]
    cor = (pf[:1, :, None, :] - df[:1, None, :, :]).abs().double().view(-1, 512, 1)
    want = link.double()(cor).view(P, P)
    link.float()
    err = (raw[0].double() - want).abs().max().item()
    print(f"{nb} x {P}^2: link head {t * 1e3:7.1f} us ({fl / t / 1e9:6.1f} TF = {fl / t / 1e9 / 157.3:.3f} of peak, incl. softmax)  "
          f"max err vs f64 {err:.2e}", flush=True)
