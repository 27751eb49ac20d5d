"""do MIOpen's exhaustive find / forced search pick faster fp32 kernels for the image branch's seven 3x3 convolutions?
Run in a FRESH process per setting (the find results are cached per process and in ~/.config/miopen)."""
import os, sys, time
import torch, torch.nn.functional as F
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
layout = sys.argv[2] if len(sys.argv) > 2 else "nhwc"
fmt = torch.channels_last if layout == "nhwc" else torch.contiguous_format
torch.backends.cudnn.benchmark = mode not in ("default", "deterministic")
torch.backends.cudnn.deterministic = mode.startswith("deterministic")   # MIOpen: excludes the split-K (atomic add) solvers
shapes = [(64, 64, 384, 1280, 2), (64, 128, 192, 640, 1), (128, 128, 192, 640, 2), (128, 256, 96, 320, 1), (256, 256, 96, 320, 2),
          (256, 512, 48, 160, 1), (512, 512, 48, 160, 2)]
tot = 0.0
for cin, cout, H, W, s in shapes:
    x = torch.randn(8, cin, H, W, device="cuda").contiguous(memory_format=fmt)
    w = torch.randn(cout, cin, 3, 3, device="cuda").contiguous(memory_format=fmt)
    t0 = time.time()
    for _ in range(3): F.conv2d(x, w, None, stride=s, padding=1)
    torch.cuda.synchronize(); setup = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): F.conv2d(x, w, None, stride=s, padding=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fl = 2 * 8 * (H // s) * (W // s) * 9 * cin * cout
    tot += ms
    print(f"{mode:10s} {layout} {cin:4d}->{cout:4d} @{H}x{W} s{s}: {ms:.3f} ms  {fl / ms / 1e9:6.1f} TF  (first calls {setup:.1f} s)", flush=True)
print(f"{mode:10s} total {tot:.3f} ms")
