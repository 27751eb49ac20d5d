"""csrc/conv_rgb.hip vs MIOpen convolution + bias/ReLU pass on the first image layer (8 x 3 x 384 x 1280 -> 64 channels)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from jmodt_amd import _lib
if os.environ.get("JM_TOOLS_LIB"): _lib.LIB_PATH = os.environ["JM_TOOLS_LIB"]
from jmodt_amd.ops.fusion import conv3x3_rgb_bias_relu, pack_rgb_weight, bias_relu_

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

img = torch.rand(8, 3, 384, 1280, device="cuda")
W = torch.randn(64, 3, 3, 3, device="cuda") * 0.3
b = torch.randn(64, device="cuda") * 0.1
wt = pack_rgb_weight(W)
Wcl = W.contiguous(memory_format=torch.channels_last)
t1 = timeit(lambda: conv3x3_rgb_bias_relu(img, W, b, wt))
def lib():
    x = img.contiguous(memory_format=torch.channels_last)
    return bias_relu_(F.conv2d(x, Wcl, None, padding=1), b)
t2 = timeit(lib)
gb = 8 * 384 * 1280 * 67 * 4 / 1e9
print(f"conv_rgb {t1:.1f} us ({gb / t1 * 1e6:.0f} GB/s on {gb:.2f} GB)   MIOpen conv + bias_relu {t2:.1f} us")
