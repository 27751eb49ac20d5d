"""which piece of a training section does not survive the capture of its BACKWARD graph?  Each case in a subprocess of its own
(a failing capture takes the process down):   gpurun -- 'python tools/graph_capture_bisect.py'"""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CASES = ["full_alone", "full_after_eager", "seq_nograd_then_grad", "seq_sidestream_nograd_then_grad", "seq_grad_grad", "seq_geom_then_grad", "seq_take_then_grad", "fold", "fold_nocat", "fold_kernel_only", "cat_split", "conv_rgb", "conv_wino", "conv2d_s2", "conv_bwd_op", "rows_mlp"]

def run_case(name):
    import torch
    import torch.nn.functional as F
    from jmodt_amd.graphed import GraphedSection
    from jmodt_amd import train_rows as TR, train_joint
    from jmodt_amd.detector import DetectorConfig
    from jmodt_amd.ops import rows as R
    from tests.test_gpu_detector import make_engine
    dev = "cuda:0"
    eng = make_engine(seed=3, cfg=DetectorConfig.tiny()).to(dev)
    train_joint.prepare_rows(eng)
    for p in eng.parameters():
        p.requires_grad_(True)
    net = eng.rpn.backbone_net
    blk = net.Img_Block[1]
    pairs = [(blk.conv1, blk.bn1)]
    x = torch.randn(2, blk.conv1.in_channels, 48, 160, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    if name.startswith("full_"):
        import numpy as np
        from jmodt_amd import synth, train_graphs
        os.environ["JM_GRAPH_TRACE"] = "1"
        for m in eng.modules():
            if isinstance(m, torch.nn.Dropout):
                m.eval()
        xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
        xyz, img, xy = torch.from_numpy(xyz).to(dev), torch.from_numpy(img).to(dev), torch.from_numpy(xy).to(dev)
        K = min(64, eng.cfg.rpn_post_nms_top_n)
        tids = torch.randint(0, 6, (2, K)).float().to(dev)
        if name == "full_after_eager":
            ref = TR.joint_forward_rows(eng, xyz, img, xy, rois_per_frame=K)
            train_joint.thin_loss(eng, ref, tids).backward()
            torch.cuda.synchronize()
            eng.zero_grad(set_to_none=True)
        loss, _ = train_graphs.forward_backward(eng, xyz, img, xy, tids, None, True, K, None)
        torch.cuda.synchronize()
        print("OK", name, float(loss), flush=True)
        return
    if name.startswith("seq_"):
        W = torch.nn.Parameter(torch.randn(64, 32, device=dev) * 0.1)
        xr = torch.randn(500, 32, device=dev, requires_grad=True)
        grad_sec = GraphedSection(lambda xv, w: R.rows_mlp(xv, [(w, None)], [1]), "g1")
        if name == "seq_nograd_then_grad":
            a = GraphedSection(lambda t: t * 2 + 1, "ng")
            a(torch.ones(1000, device=dev))
        elif name == "seq_sidestream_nograd_then_grad":
            a = GraphedSection(lambda t: t * 2 + 1, "ng")
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side), torch.no_grad():
                a(torch.ones(1000, device=dev))
            torch.cuda.current_stream().wait_stream(side)
        elif name == "seq_grad_grad":
            g0 = GraphedSection(lambda xv, w: R.rows_mlp(xv, [(w, None)], [1]), "g0")
            g0(xr, W).sum().backward()
        elif name in ("seq_geom_then_grad", "seq_take_then_grad"):
            import numpy as np
            from jmodt_amd import synth, train_graphs
            xyz, img, xy = synth.frames(2, 2048, 77, H=96, W=320, native=(94, 310))
            xyz, xy = torch.from_numpy(xyz).to(dev), torch.from_numpy(xy).to(dev)
            jg = train_graphs.joint_graphs(eng)
            if name == "seq_geom_then_grad":
                with torch.no_grad():
                    jg.sec_geom(xyz, xy)
            else:
                jg.geometry(xyz, xy)
        torch.cuda.synchronize()
        print("first part done", flush=True)
        grad_sec(xr, W).sum().backward()
        torch.cuda.synchronize()
        print("OK", name, flush=True)
        return
    if name == "fold":
        def fn(w, g, b):
            f = TR.BnFold(None, pairs)
            wf, t = f.conv4d(blk.conv1)
            return wf.sum() + t.sum()
        args = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias]
    elif name == "fold_nocat":
        def fn(w, g, b):
            inv = torch.rsqrt(blk.bn1.running_var + blk.bn1.eps)
            s = g * inv
            (wf,) = R.fold_all(s, [0], [w])
            return wf.sum() + (b - blk.bn1.running_mean * s).sum()
        args = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias]
    elif name == "fold_kernel_only":
        s0 = torch.rand(blk.conv1.out_channels, device=dev)
        def fn(w):
            (wf,) = R.fold_all(s0, [0], [w])
            return wf.sum()
        args = [blk.conv1.weight]
    elif name == "cat_split":
        ps = [torch.nn.Parameter(torch.randn(16, device=dev)) for _ in range(6)]
        def fn(*ts):
            c = torch.cat(ts) * 2.0
            parts = torch.split(c, [16] * 6)
            return sum(p.sum() * (i + 1) for i, p in enumerate(parts))
        args = ps
    elif name in ("conv_rgb", "conv_wino"):
        b2 = net.Img_Block[0] if name == "conv_rgb" else blk
        xx = torch.randn(2, b2.conv1.in_channels, 48, 160, device=dev)
        xx = (xx if name == "conv_rgb" else xx.contiguous(memory_format=torch.channels_last)).requires_grad_(name != "conv_rgb")
        bias = torch.nn.Parameter(torch.randn(b2.conv1.out_channels, device=dev))
        def fn(xv, w, bb):
            return TR._Conv3x3BiasRelu.apply(xv, w, bb)
        args = [xx, b2.conv1.weight, bias]
    elif name == "conv2d_s2":
        xc = torch.randn(2, blk.conv2.in_channels, 48, 160, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
        def fn(xv, w, bb):
            return F.conv2d(xv, w, bb, stride=2, padding=1)
        args = [xc, blk.conv2.weight, blk.conv2.bias]
    elif name == "conv_bwd_op":
        xc = torch.randn(2, blk.conv2.in_channels, 48, 160, device=dev).contiguous(memory_format=torch.channels_last)
        w = blk.conv2.weight
        def fn(dy):
            gx, dw, db = torch.ops.aten.convolution_backward(dy, xc, w.detach(), [w.shape[0]], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
            return gx.sum() + dw.sum() + db.sum()
        args = [torch.randn(2, w.shape[0], 48, 160, device=dev).contiguous(memory_format=torch.channels_last)]
    elif name == "rows_mlp":
        W = torch.nn.Parameter(torch.randn(64, 32, device=dev) * 0.1)
        xr = torch.randn(500, 32, device=dev, requires_grad=True)
        def fn(xv, w):
            return R.rows_mlp(xv, [(w, None)], [1])
        args = [xr, W]
    sec = GraphedSection(fn, name)
    out = sec(*args)
    out = out if isinstance(out, torch.Tensor) else out[0]
    if out.requires_grad:
        out.sum().backward()
    torch.cuda.synchronize()
    print("OK", name, flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        run_case(sys.argv[1])
    else:
        for c in CASES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True)
            tail = [l for l in (r.stdout + r.stderr).splitlines() if l.strip() and "amdgpu.ids" not in l]
            print(f"{c:18s} rc {r.returncode:4d}  {tail[-1][:160] if tail else ''}", flush=True)
            if c.startswith("full_"):
                print("\n".join(l for l in tail if l.startswith("[graphed]") or "Error" in l or "error" in l)[-1500:], flush=True)
