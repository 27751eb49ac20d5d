"""does aten::miopen_convolution_relu (conv + bias + ReLU as one MIOpen fusion plan) work for the image branch's fp32
channels-last 3x3 convolutions, and what does it save over conv + BatchNorm + ReLU?"""
import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

torch.manual_seed(0)
for cin, cout, h, w in ((3, 64, 384, 1280), (64, 128, 192, 640), (128, 256, 96, 320), (256, 512, 48, 160)):
    x = torch.randn(8, cin, h, w, device="cuda").contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(cout, device="cuda") * 0.1
    bn = torch.nn.BatchNorm2d(cout).cuda().eval()
    ref = lambda: F.relu(bn(F.conv2d(x, wgt, None, 1, 1)), inplace=True)
    plain = lambda: torch.relu_(F.conv2d(x, wgt, b, 1, 1))
    line = f"{cin}->{cout} @{h}x{w}: conv+bn+relu {timeit(ref):.3f} ms | conv(bias)+relu_ {timeit(plain):.3f} ms"
    try:
        fused = lambda: torch.ops.aten.miopen_convolution_relu(x, wgt, b, [1, 1], [1, 1], [1, 1], 1)
        y = fused()
        err = (y - plain()).abs().max().item()
        line += f" | miopen_convolution_relu {timeit(fused):.3f} ms (max diff {err:.2e}, channels_last out {y.is_contiguous(memory_format=torch.channels_last)})"
    except Exception as ex:
        line += f" | miopen_convolution_relu FAILED: {str(ex)[:120]}"
    conv_only = lambda: F.conv2d(x, wgt, None, 1, 1)
    line += f" | conv only {timeit(conv_only):.3f} ms"
    print(line, flush=True)
