"""csrc/conv1d_stack.hip vs the library route (batched GEMM + bias broadcast + ReLU per layer) on the composed path's shapes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd.ops.conv1d import PackedConv1dStack

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

dev = "cuda"
cases = [  # name, B, n, c0, c1, xyz1, widths, relus
    ("hoisted SA layer, RPN SA2", 8, 4096, 96, 3, True, [64], [False]),
    ("hoisted SA layer, RCNN SA2", 1024, 128, 128, 3, True, [128], [False]),
    ("RPN heads (both)", 8, 16384, 128, 0, False, [256, 77], [True, False]),
    ("FP1 256+? -> 128,128", 8, 16384, 256, 0, False, [128, 128], [True, True]),
    ("FP2 512+96 -> 256,256", 8, 4096, 512, 96, False, [256, 256], [True, True]),
    ("FP3 512+256 -> 512,512", 8, 1024, 512, 256, False, [512, 512], [True, True]),
    ("FP4 1024+512 -> 512,512 (64 tiles)", 8, 256, 1024, 512, False, [512, 512], [True, True]),
]
for name, B, n, c0, c1, xyz1, widths, relus in cases:
    layers, k = [], c0 + c1
    for w, r in zip(widths, relus):
        layers.append(((torch.randn(w, k) * (2.0 / k) ** 0.5).to(dev), (torch.randn(w) * 0.2).to(dev), r))
        k = w
    x0 = torch.randn(B, c0, n, device=dev)
    x1 = (torch.randn(B, n, 3, device=dev) if xyz1 else torch.randn(B, c1, n, device=dev)) if c1 else None
    st = PackedConv1dStack(layers, c0, c1, xyz1)

    def lib():
        W, b, r = layers[0]
        x = torch.baddbmm(b[None, :, None], W[:, :c0].expand(B, -1, -1), x0)
        if c1:
            x = x.baddbmm_(W[:, c0:].expand(B, -1, -1), x1.transpose(1, 2) if xyz1 else x1)
        if r: x = torch.relu_(x)
        for W, b, r in layers[1:]:
            x = torch.baddbmm(b[None, :, None], W.expand(B, -1, -1), x)
            if r: x = torch.relu_(x)
        return x
    fl = 2 * B * n * sum(a * b for a, b in zip([c0 + c1] + widths[:-1], widths))
    import jmodt_amd.ops.conv1d as C1
    C1.TILE64 = 0
    t_k, t_l = timeit(lambda: st(x0, x1)), timeit(lib)
    C1.TILE64 = 2
    t_64 = timeit(lambda: st(x0, x1)) if st.supported64(B, n) else float("nan")
    print(f"{name:28s} 32-point tiles {t_k:8.1f} us ({fl / t_k / 1e6:6.1f} TF)   64-point tiles {t_64:8.1f} us ({fl / t_64 / 1e6:6.1f} TF)   library {t_l:8.1f} us   "
          f"max diff {(st(x0, x1) - lib()).abs().max().item():.2e}", flush=True)
