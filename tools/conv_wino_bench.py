"""fused Winograd F(2x2,3x3) convolution + bias + ReLU (csrc/conv_wino.hip) against MIOpen conv + bias_relu_ on the image branch's
three stride-1 layers (batch 8): error vs float64, time per layer"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jmodt_amd import _lib
from jmodt_amd.csrc import build as _hip_build
if os.environ.get("JM_WN_G"):            # patches per channel-block group: a switch of the tools build only
    _lib.LIB_PATH = _hip_build.TOOLS_LIB
from jmodt_amd.ops.fusion import bias_relu_, conv3x3_wino_bias_relu, pack_wino_weight

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
B = int(os.environ.get("B", 8))
for cin, cout, H, W in ((64, 128, 192, 640), (128, 256, 96, 320), (256, 512, 48, 160)):
    x = torch.relu(torch.randn(B, cin, H, W, device="cuda")).contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda") * (2.0 / (9 * cin)) ** 0.5
    b = torch.randn(cout, device="cuda") * 0.1
    wcl = w.contiguous(memory_format=torch.channels_last)
    packed = pack_wino_weight(w)
    y = conv3x3_wino_bias_relu(x, packed, b, cout)
    torch.backends.cudnn.benchmark = True
    ref = bias_relu_(F.conv2d(x, wcl, None, padding=1), b)
    # float64 on a slice of the batch
    r64 = torch.relu(F.conv2d(x[:1].double(), w.double(), b.double(), padding=1))
    scale = r64.abs().max().item()
    e_w = (y[:1].double() - r64).abs().max().item() / scale
    e_m = (ref[:1].double() - r64).abs().max().item() / scale
    t_w = timeit(lambda: conv3x3_wino_bias_relu(x, packed, b, cout))
    t_m = timeit(lambda: bias_relu_(F.conv2d(x, wcl, None, padding=1), b))
    t_c = timeit(lambda: F.conv2d(x, wcl, None, padding=1))
    fl = 2 * 9 * cin * cout * H * W * B
    print(f"{cin}->{cout} @{H}x{W}: winograd {t_w:.3f} ms ({fl / t_w / 1e9:.1f} TF direct-equivalent, {fl / 2.25 / t_w / 1e9:.1f} TF MFMA) "
          f"err {e_w:.2e} | MIOpen conv {t_c:.3f} + bias/relu = {t_m:.3f} ms ({fl / t_c / 1e9:.1f} TF) err {e_m:.2e}", flush=True)
