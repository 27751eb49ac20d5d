"""GPU probe (run through gpurun): forward + backward time of the image side of the joint-mode training step —
the four BasicBlocks (conv3x3 + folded BN + ReLU + conv3x3 / 2) and the deconvolution pyramid + fusion convolution in its
composed (detector._image_fusion_map) and un-composed (backbone.py:187-193: cat of four deconvolutions + 1x1 conv) forms —
in channels-last memory, with and without MIOpen's find mode (torch.backends.cudnn.benchmark)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
from jmodt_amd.train_rows import BnFold, _image_pyramid, _image_fusion_map

dev = "cuda:0"
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.manual_seed(0)
eng = DetectAffinityEngine(DetectorConfig.survey()).to(dev)
for p in eng.parameters():
    p.requires_grad_(True)
net = eng.rpn.backbone_net
image = torch.randn(B, 3, 384, 1280, device=dev)


def uncomposed(fold, maps):
    de = torch.cat([dc(m) for dc, m in zip(net.DeConv, maps)], dim=1)
    Wf, bf = fold.conv(net.image_fusion_conv, net.image_fusion_bn)
    return F.relu(F.conv2d(de, Wf[:, :, None, None], bf))


def timeit(fn, n=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for bench in (False, True):
    torch.backends.cudnn.benchmark = bench

    def pyr_fb():
        eng.zero_grad(set_to_none=True)
        fold = BnFold(eng)
        maps = _image_pyramid(fold, net, image)
        sum(m.sum() for m in maps).backward()

    def pyr_f():
        with torch.no_grad():
            _image_pyramid(BnFold(eng), net, image)

    def fus(form):
        def run():
            eng.zero_grad(set_to_none=True)
            fold = BnFold(eng)
            with torch.no_grad():
                maps = _image_pyramid(fold, net, image)
            maps = [m.detach().requires_grad_() for m in maps]
            out = form(fold, maps)
            (out * out).sum().backward()
        return run

    def pyr_only():
        with torch.no_grad():
            _image_pyramid(BnFold(eng), net, image)
    tp = timeit(pyr_only)
    print(f"benchmark={bench}: pyramid fwd {timeit(pyr_f):.2f} ms, fwd+bwd {timeit(pyr_fb):.2f} ms; fusion map fwd+bwd (incl. {tp:.2f} ms of pyramid fwd): "
          f"composed {timeit(fus(lambda f, m: _image_fusion_map(f, net, m))):.2f} ms, un-composed {timeit(fus(uncomposed)):.2f} ms", flush=True)
