"""csrc/points_gemm.hip against the library GEMMs it replaces, at the two places of the composed step that used rocBLAS until
round 5: feature propagation level 4 (8 x 256 points, [1024 | 512] -> 512 -> 512) and the level-4 LI-Fusion attention block
(8 x 64 points: fc1/fc2 -> tanh -> fc3 -> sigmoid; 512 -> 1024 conv * gate; [1024 | 1024] -> 1024).  Event timing over 50 calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from jmodt_amd.ops.conv1d import points_linear
dev = "cuda:0"
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
R = lambda *s: torch.randn(*s, generator=g).to(dev)
# --- FP4
B, n = 8, 256
x1, x2 = R(B, 1024, n), R(B, 512, n)
W1, b1, W2, b2 = R(512, 1536) * 0.02, R(512), R(512, 512) * 0.04, R(512)
def fp_old():
    x = torch.baddbmm(b1[None, :, None], W1[:, :1024].expand(B, -1, -1), x1)
    x = torch.relu_(torch.baddbmm(x, W1[:, 1024:].expand(B, -1, -1), x2))
    return torch.relu_(torch.baddbmm(b2[None, :, None], W2.expand(B, -1, -1), x))
def fp_new():
    return points_linear(points_linear(x1, W1, b1, 1, x2=x2), W2, b2, 1)
print("fp4 max |new - old|", float((fp_new() - fp_old()).abs().max()))
print(f"fp4: library {timeit(fp_old):.1f} us, points_linear {timeit(fp_new):.1f} us")
# --- level-4 attention
n = 64
P, I = R(B, 1024, n), R(B, 512, n)
rc = 256
w12, b12 = R(rc, 1536) * 0.02, R(rc)
w3 = torch.zeros(4, rc, device=dev); w3[0] = R(rc) * 0.05
b3 = torch.zeros(4, device=dev)
Wi, bi = R(1024, 512) * 0.04, R(1024)
Wf, bf = R(1024, 2048) * 0.02, R(1024)
def att_old():
    it, pt = I.transpose(1, 2), P.transpose(1, 2)
    t = torch.tanh(it @ w12[:, :512].t() + pt @ w12[:, 512:].t() + b12)
    gate = torch.sigmoid(t @ w3[:1].t() + b3[0])
    img_new = torch.relu(torch.baddbmm(bi[None, :, None], Wi.expand(B, -1, -1), I)) * gate.transpose(1, 2)
    out = torch.baddbmm(bf[None, :, None], Wf[:, :1024].expand(B, -1, -1), P)
    return torch.relu_(torch.baddbmm(out, Wf[:, 1024:].expand(B, -1, -1), img_new))
def att_new():
    t = points_linear(I, w12, b12, 2, x2=P)
    gate = points_linear(t, w3, b3, 3, out_rows=4)
    img_new = points_linear(I, Wi, bi, 1, rowscale=gate, rowscale_stride=4)
    return points_linear(P, Wf, bf, 1, x2=img_new)
print("attention max |new - old|", float((att_new() - att_old()).abs().max()))
print(f"attention level 4: library {timeit(att_old):.1f} us, points_linear {timeit(att_new):.1f} us")
for name, fn in (("fc12", lambda: points_linear(I, w12, b12, 2, x2=P)), ("conv_img", lambda: points_linear(I, Wi, bi, 1)),
                 ("fuse", lambda: points_linear(P, Wf, bf, 1, x2=P)), ("fp4 layer 1", lambda: points_linear(x1, W1, b1, 1, x2=x2))):
    print(f"   {name}: {timeit(fn):.1f} us")
