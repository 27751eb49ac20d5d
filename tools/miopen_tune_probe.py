"""does MIOpen's tuning search (MIOPEN_FIND_ENFORCE=SEARCH) find faster fp32 kernels than its find mode for the image branch's four
stride-2 3x3 convolutions?  Run once with the variable unset and once with it set (fresh process each)."""
import os, sys, time
import torch, torch.nn.functional as F
torch.backends.cudnn.benchmark = True
shapes = [(64, 64, 384, 1280), (128, 128, 192, 640), (256, 256, 96, 320), (512, 512, 48, 160)]
tot = 0.0
for cin, cout, H, W in shapes:
    x = torch.randn(8, cin, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, 3, 3, device="cuda").contiguous(memory_format=torch.channels_last)
    t0 = time.time()
    for _ in range(3): F.conv2d(x, w, None, stride=2, padding=1)
    torch.cuda.synchronize(); setup = time.time() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): F.conv2d(x, w, None, stride=2, padding=1)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    tot += ms
    print(f"{os.environ.get('MIOPEN_FIND_ENFORCE', 'find')}: {cin}->{cout} @{H}x{W} s2: {ms:.3f} ms  {2 * 8 * (H // 2) * (W // 2) * 9 * cin * cout / ms / 1e9:6.1f} TF (first calls {setup:.1f} s)", flush=True)
print("total", round(tot, 3))
