"""1x1 Conv2d of the un-fused (training) set-abstraction route on (B, C, npoint, nsample) tensors: MIOpen's convolution against the
same product as one broadcast matmul  W (O, C) @ x.view(B, C, npoint * nsample)  (rocBLAS strided-batched GEMM), forward + backward.
    gpurun -- 'python tools/pointwise_conv_bench.py'"""
import torch, time
dev = torch.device("cuda:0")
shapes = [(4, 99, 1024, 32, 128), (4, 128, 1024, 32, 128), (4, 259, 256, 32, 256), (4, 515, 64, 32, 256), (512, 131, 128, 64, 128), (512, 128, 128, 64, 128),
          (512, 259, 32, 64, 256)]
for B, C, P, S, O in shapes:
    conv = torch.nn.Conv2d(C, O, 1, bias=False).to(dev)
    x = torch.randn(B, C, P, S, device=dev, requires_grad=True)
    g = torch.randn(B, O, P, S, device=dev)

    def f_conv():
        y = conv(x); y.backward(g); return y

    def f_mm():
        y = torch.matmul(conv.weight.view(O, C), x.view(B, C, P * S)).view(B, O, P, S); y.backward(g); return y
    res = []
    for f in (f_conv, f_mm):
        for _ in range(3):
            x.grad = None; conv.weight.grad = None; f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            x.grad = None; conv.weight.grad = None; y = f()
        torch.cuda.synchronize(); res.append(((time.perf_counter() - t0) / 10 * 1e3, y.detach(), x.grad.clone(), conv.weight.grad.clone()))
    (tc, yc, gxc, gwc), (tm, ym, gxm, gwm) = res
    fl = 3 * 2.0 * B * C * P * S * O
    print(f"({B},{C},{P},{S}) -> {O}: conv2d fwd+bwd {tc:7.3f} ms ({fl / tc / 1e9:6.1f} TF)   matmul {tm:7.3f} ms ({fl / tm / 1e9:6.1f} TF)   "
          f"max |dy| {(yc - ym).abs().max().item():.1e} |dgx| {(gxc - gxm).abs().max().item():.1e} |dgw| {((gwc - gwm).abs().max() / gwc.abs().max()).item():.1e}", flush=True)
