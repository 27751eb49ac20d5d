"""GPU tier: out-of-bounds WRITE detection.  Every output buffer handed to the C ABI sits inside a larger
allocation whose margins hold a sentinel pattern; after the call the margins must be untouched.  Shapes
are chosen off the kernels' vector widths (odd counts, sizes that are not multiples of 4 / 64 / 128)."""
import ctypes

import numpy as np
import pytest
import torch

from jmodt_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PAD = 4096   # elements on each side; the odd variant also misaligns every buffer (scalar store paths)
SENT = {torch.float32: 1.2345678e30, torch.int32: 0x5A5A5A5A, torch.int64: 0x5A5A5A5A5A5A5A5A}


@pytest.fixture(autouse=True, params=[4096, 4099])
def _pad(request):
    global PAD
    PAD = request.param
    yield


class Guard:
    def __init__(self, shape, dtype, fill=None):
        n = int(np.prod(shape))
        self.big = torch.full((n + 2 * PAD,), SENT[dtype], dtype=dtype, device=DEV)
        self.view = self.big[PAD:PAD + n].view(*shape)
        if fill is not None:
            self.view.fill_(fill)
        self.n = n

    def intact(self):
        s = SENT[self.big.dtype]
        lo, hi = self.big[:PAD], self.big[PAD + self.n:]
        return bool((lo == s).all().item() and (hi == s).all().item())


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def test_pointnet2_outputs_stay_in_bounds():
    from jmodt_amd.ext import pointnet2_cuda as pc
    B, N, m, ns, C = 3, 1003, 131, 17, 7
    xyz = T(synth.cloud(B, N, seed=3))
    temp = Guard((B, N), torch.float32, 1e10)
    idx = Guard((B, m), torch.int32, 0)
    pc.farthest_point_sampling_wrapper(B, N, m, xyz, temp.view, idx.view)
    assert temp.intact() and idx.intact()
    new_xyz = Guard((B, 3, m), torch.float32)
    pc.gather_points_wrapper(B, 3, N, m, xyz.transpose(1, 2).contiguous(), idx.view, new_xyz.view)
    assert new_xyz.intact()
    centres = new_xyz.view.transpose(1, 2).contiguous()
    nb = Guard((B, m, ns), torch.int32, 0)
    pc.ball_query_wrapper(B, N, m, 2.5, ns, centres, xyz, nb.view)
    assert nb.intact() and int(nb.view.max()) < N
    feats = torch.randn(B, C, N, device=DEV)
    grouped = Guard((B, C, m, ns), torch.float32)
    pc.group_points_wrapper(B, C, N, m, ns, feats, nb.view, grouped.view)
    assert grouped.intact()
    gp = Guard((B, C, N), torch.float32, 0.0)
    pc.group_points_grad_wrapper(B, C, N, m, ns, torch.randn(B, C, m, ns, device=DEV), nb.view, gp.view)
    assert gp.intact()
    gg = Guard((B, C, N), torch.float32, 0.0)
    pc.gather_points_grad_wrapper(B, C, N, m, torch.randn(B, C, m, device=DEV), idx.view, gg.view)
    assert gg.intact()
    d2, i3 = Guard((B, N, 3), torch.float32), Guard((B, N, 3), torch.int32)
    pc.three_nn_wrapper(B, N, m, xyz, centres, d2.view, i3.view)
    assert d2.intact() and i3.intact()
    w = torch.rand(B, N, 3, device=DEV)
    out = Guard((B, C, N), torch.float32)
    pc.three_interpolate_wrapper(B, C, m, N, torch.randn(B, C, m, device=DEV), i3.view, w, out.view)
    assert out.intact()
    gi = Guard((B, C, m), torch.float32, 0.0)
    pc.three_interpolate_grad_wrapper(B, C, N, m, torch.randn(B, C, N, device=DEV), i3.view, w, gi.view)
    assert gi.intact()


def test_cooperative_fps_outputs_stay_in_bounds():
    from jmodt_amd.ext import pointnet2_cuda as pc
    B, N, m = 2, 20011, 77
    xyz = T(synth.cloud(B, N, seed=4))
    temp, idx = Guard((B, N), torch.float32, 1e10), Guard((B, m), torch.int32, 0)
    pc.farthest_point_sampling_wrapper(B, N, m, xyz, temp.view, idx.view)
    assert temp.intact() and idx.intact()


@pytest.mark.parametrize("S,C", [(129, 5), (512, 130), (64, 0)])
def test_roipool3d_outputs_stay_in_bounds(S, C):
    from jmodt_amd.ext import roipool3d_cuda as rc
    B, N, M = 2, 3001, 9
    pts = synth.dense_cloud(B, N, 8, extent=10.0)
    boxes = T(synth.proposals(pts, M, 9))
    feat = torch.randn(B, N, max(C, 1), device=DEV)[:, :, :C].contiguous()
    for fn in ("forward", "forward_canonical"):
        pooled, flag = Guard((B, M, S, 3 + C), torch.float32, 0.0), Guard((B, M), torch.int32, 0)
        if fn == "forward":
            rc.forward(T(pts), boxes, feat, pooled.view, flag.view, zero_empty=1)
        else:
            rc.forward_canonical(T(pts), boxes, 0.2, feat, pooled.view, flag.view)
        assert pooled.intact() and flag.intact(), fn


def test_iou3d_nms_outputs_stay_in_bounds():
    from jmodt_amd.ext import iou3d_cuda as ic
    from jmodt_amd import _lib as L
    a, sa = synth.bev_boxes(77, 1)
    b, _ = synth.bev_boxes(53, 2)
    ov, iou = Guard((77, 53), torch.float32), Guard((77, 53), torch.float32)
    ic.boxes_overlap_bev_gpu(T(a), T(b), ov.view)
    ic.boxes_iou_bev_gpu(T(a), T(b), iou.view)
    assert ov.intact() and iou.intact()
    lib = L.load()
    n = 1001
    bx, sc = synth.bev_boxes(n, 3)
    bx = T(bx[np.argsort(-sc)])
    keep, num = Guard((n,), torch.int64), Guard((1,), torch.int32)
    wsb = lib.jm_nms_workspace_bytes(n)
    ws = Guard((wsb // 8,), torch.int64)
    for normal in (0, 1):
        L.check(lib.jm_nms(n, L.dev(bx, torch.float32, "boxes"), 0.7, normal, ctypes.c_void_p(keep.view.data_ptr()),
                           ctypes.c_void_p(num.view.data_ptr()), ctypes.c_void_p(ws.view.data_ptr()), wsb, L.stream_ptr()),
                "nms")
        assert keep.intact() and num.intact() and ws.intact()


def test_fused_sa_and_feature_gather_outputs_stay_in_bounds():
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    lib = L.load()
    torch.manual_seed(0)
    sa = PointnetSAModuleMSG(npoint=24, radii=[1.5], nsamples=[16], mlps=[[5, 24, 40]], bn=False).to(DEV).eval()
    B, N = 3, 333
    xyz = T(synth.dense_cloud(B, N, 5, extent=4.0))
    feats = torch.randn(B, 5, N, device=DEV)
    with torch.no_grad():
        idx = pu.farthest_point_sample(xyz, 24)
        new_xyz = pu.gather_operation(xyz.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        nb = pu.ball_query(1.5, 16, xyz, new_xyz)
        layers = fused._packed_layers(sa.mlps[0], xyz.device)
    out = Guard((B, 40, 24), torch.float32)
    widths = (ctypes.c_int * 3)(8, 24, 40)
    warr = (ctypes.c_void_p * 2)(*[l[0].data_ptr() for l in layers])
    barr = (ctypes.c_void_p * 2)(*[l[1].data_ptr() for l in layers])
    L.check(lib.jm_sa_mlp_forward(B, N, 24, 5, 16, L.dev(xyz, torch.float32, "xyz"), L.dev(new_xyz, torch.float32, "c"),
                                  L.dev(feats, torch.float32, "f"), L.dev(nb, torch.int32, "i"), 2, widths, warr, barr,
                                  ctypes.c_void_p(out.view.data_ptr()), L.stream_ptr()), "sa_mlp")
    assert out.intact()
    fm = torch.randn(2, 6, 11, 13, device=DEV)
    xy = torch.rand(2, 101, 2, device=DEV) * 2.4 - 1.2
    g = Guard((2, 6, 101), torch.float32)
    L.check(lib.jm_feature_gather(2, 6, 11, 13, 101, L.dev(fm, torch.float32, "fm"), *[int(s) for s in fm.stride()],
                                  L.dev(xy, torch.float32, "xy"), ctypes.c_void_p(g.view.data_ptr()), L.stream_ptr()), "fg")
    assert g.intact()


def test_round3_entries_stay_in_bounds_and_refuse_bad_arguments():
    """jm_nms_normal_first_k_batched, jm_ball_query_grid_build / _query, the vector-pipe set-abstraction scale (sa_xyz.hip) through
    jm_sa_mlp_forward: outputs and workspaces inside guarded allocations; capacity / workspace violations return an error"""
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    lib = L.load()
    f32, i32 = torch.float32, torch.int32
    # ---- first-K NMS: ragged problems, K below / above the survivor counts
    counts = [1001, 0, 65, 513]
    nmax = 1001
    boxes = np.zeros((len(counts), nmax, 5), np.float32)
    for p, c in enumerate(counts):
        if c:
            b, s = synth.bev_boxes(c, 40 + p)
            boxes[p, :c] = b[np.argsort(-s, kind="stable")]
    tb, tc = T(boxes), T(np.array(counts, np.int32))
    for k in (7, 600):
        keep, num = Guard((len(counts), nmax), torch.int64, fill=-1), Guard((len(counts),), i32)
        L.check(lib.jm_nms_normal_first_k_batched(len(counts), nmax, L.dev(tc, i32, "counts"), L.dev(tb, f32, "boxes"), 0.7, k,
                                                  ctypes.c_void_p(keep.view.data_ptr()), ctypes.c_void_p(num.view.data_ptr()),
                                                  L.stream_ptr()), "first_k")
        assert keep.intact() and num.intact()
        n = num.view.cpu().numpy()
        assert n[1] == 0 and (n <= k).all() and n[0] == min(k, n[0])
        for p in range(len(counts)):
            assert (keep.view[p, n[p]:] == -1).all()            # nothing written behind the kept entries
    assert lib.jm_nms_normal_first_k_batched(1, nmax, L.dev(tc, i32, "counts"), L.dev(tb, f32, "boxes"), 0.7, 4096,
                                             ctypes.c_void_p(keep.view.data_ptr()), ctypes.c_void_p(num.view.data_ptr()),
                                             L.stream_ptr()) != 0 and b"2048" in lib.jm_last_error()
    # ---- hash-grid ball query in two calls: guarded workspace and index outputs, short workspace refused
    B, N, M = 3, 2500, 333
    xyz = T(synth.kitti_like_cloud(B, N, 9))
    cen = xyz[:, :M].contiguous()
    wsb = lib.jm_ball_query_workspace_bytes(B, N)
    assert wsb > 0
    if PAD % 64 == 0:
        ws = Guard((wsb // 4,), i32)
    else:       # the workspace must be 256-byte aligned (header): the misaligning variant keeps it in a plain allocation
        ws = type("Plain", (), {"view": torch.empty((wsb // 4,), dtype=i32, device=DEV), "intact": lambda self: True})()
    i0, i1 = Guard((B, M, 16), i32, fill=0), Guard((B, M, 32), i32, fill=0)
    L.check(lib.jm_ball_query_grid_build(B, N, 0.8, L.dev(xyz, f32, "xyz"), ctypes.c_void_p(ws.view.data_ptr()), wsb, L.stream_ptr()), "build")
    L.check(lib.jm_ball_query_grid_query(B, N, M, 0.8, 0.4, 16, 0.8, 32, L.dev(cen, f32, "c"), ctypes.c_void_p(i0.view.data_ptr()),
                                         ctypes.c_void_p(i1.view.data_ptr()), ctypes.c_void_p(ws.view.data_ptr()), wsb, L.stream_ptr()), "query")
    assert ws.intact() and i0.intact() and i1.intact()
    assert torch.equal(i0.view, pu.ball_query(0.4, 16, xyz, cen)) and torch.equal(i1.view, pu.ball_query(0.8, 32, xyz, cen))
    assert lib.jm_ball_query_grid_build(B, N, 0.8, L.dev(xyz, f32, "xyz"), ctypes.c_void_p(ws.view.data_ptr()), wsb // 2, L.stream_ptr()) != 0
    # ---- xyz-only scale on the vector pipe (npoint a multiple of 64 / 32: jm_sa_mlp_supported == 3)
    torch.manual_seed(1)
    sa = PointnetSAModuleMSG(npoint=192, radii=[0.5, 1.0], nsamples=[16, 32], mlps=[[0, 16, 16, 32], [0, 32, 32, 64]], bn=True).to(DEV).eval()
    pts = T(synth.dense_cloud(2, 1500, 6, extent=5.0))
    with torch.no_grad():
        idx = pu.farthest_point_sample(pts, 192)
        new_xyz = pu.gather_operation(pts.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
        for k, (r, ns, cout) in enumerate(((0.5, 16, 32), (1.0, 32, 64))):
            nb = pu.ball_query(r, ns, pts, new_xyz)
            layers = fused._packed_layers(sa.mlps[k], pts.device)
            widths = (ctypes.c_int * 4)(3, *[l[2] for l in layers])
            assert lib.jm_sa_mlp_supported(2, 1500, 192, 0, ns, 0, 3, widths) == 3
            out = Guard((2, cout, 192), f32)
            warr = (ctypes.c_void_p * 3)(*[l[0].data_ptr() for l in layers])
            barr = (ctypes.c_void_p * 3)(*[l[1].data_ptr() for l in layers])
            L.check(lib.jm_sa_mlp_forward(2, 1500, 192, 0, ns, L.dev(pts, f32, "xyz"), L.dev(new_xyz, f32, "c"), None,
                                          L.dev(nb, i32, "i"), 3, widths, warr, barr, ctypes.c_void_p(out.view.data_ptr()),
                                          L.stream_ptr()), "sa_xyz")
            assert out.intact() and torch.isfinite(out.view).all() and out.view.abs().max() > 0


def test_conv3x3_wino_stays_in_bounds_and_refuses_bad_arguments():
    """jm_conv3x3_wino_pack / jm_conv3x3_wino_bias_relu: odd image sizes (partial 2x2 tiles and partial 8x16 patches at both
    borders), output and packed weight inside guarded allocations; unsupported widths, null pointers and a misaligned packed
    weight return an error"""
    import torch.nn.functional as F
    from jmodt_amd import _lib as L
    lib = L.load()
    f32 = torch.float32
    g = torch.Generator().manual_seed(11)
    for B, cin, cout, H, W in ((2, 16, 64, 7, 19), (1, 32, 128, 9, 33), (1, 16, 64, 1, 1)):
        x = torch.randn(B, cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 3, 3, generator=g) * 0.1).to(DEV)
        b = torch.randn(cout, generator=g).to(DEV)
        if PAD % 4 == 0:
            packed = Guard((16 * cin * cout,), f32)
        else:   # the packed weight must be 16-byte aligned (header): the misaligning variant keeps it in a plain allocation
            packed = type("Plain", (), {"view": torch.empty((16 * cin * cout,), dtype=f32, device=DEV), "intact": lambda self: True})()
        assert lib.jm_conv3x3_wino_packed_elems(cin, cout) == 16 * cin * cout
        L.check(lib.jm_conv3x3_wino_pack(cin, cout, L.dev(w, f32, "w"), ctypes.c_void_p(packed.view.data_ptr()), L.stream_ptr()), "pack")
        out = Guard((B, H, W, cout), f32)
        L.check(lib.jm_conv3x3_wino_bias_relu(B, H, W, cin, cout, ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(packed.view.data_ptr()),
                                              L.dev(b, f32, "b"), 1, ctypes.c_void_p(out.view.data_ptr()), L.stream_ptr()), "wino")
        assert packed.intact() and out.intact()
        want = torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
        assert (out.view.double() - want).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())
    px, po = ctypes.c_void_p(x.data_ptr()), ctypes.c_void_p(out.view.data_ptr())
    pk = ctypes.c_void_p(packed.view.data_ptr())
    assert lib.jm_conv3x3_wino_supported(24, 64) == 0 and lib.jm_conv3x3_wino_supported(16, 96) == 0 and lib.jm_conv3x3_wino_supported(16, 64) == 1
    assert lib.jm_conv3x3_wino_bias_relu(1, 4, 4, 24, 64, px, pk, None, 1, po, L.stream_ptr()) != 0
    assert lib.jm_conv3x3_wino_bias_relu(1, 4, 4, 16, 64, None, pk, None, 1, po, L.stream_ptr()) != 0
    assert lib.jm_conv3x3_wino_bias_relu(1, 4, 4, 16, 64, px, ctypes.c_void_p(packed.view.data_ptr() + 4), None, 1, po, L.stream_ptr()) != 0
    assert lib.jm_conv3x3_wino_bias_relu(0, 4, 4, 16, 64, None, None, None, 1, None, L.stream_ptr()) == 0      # empty batch: nothing to do
    assert lib.jm_conv3x3_wino_pack(16, 64, None, pk, L.stream_ptr()) != 0


def test_sa_outputs_written_into_a_channel_slice_leave_the_rest_alone():
    """jm_sa_mlp_forward_into / _pre_into / _pm_forward_into (the MSG concatenation written in place): the scale's (cout, M) blocks land
    in their channel slice of a wider guarded tensor — equal to the stand-alone call bit for bit — and every other channel and the
    margins keep the sentinel; a frame stride below cout * M is refused"""
    from jmodt_amd import _lib as L
    from jmodt_amd.ops.pointnet2 import fused, pointnet2_utils as pu
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModuleMSG
    torch.manual_seed(2)
    cases = [  # (C, npoint, nsamples, mlps): xyz-only vector-pipe scales, pre-projected narrow / pm scales, wide scales
        (0, 192, [16, 32], [[0, 16, 16, 32], [0, 32, 32, 64]]),
        (32, 128, [16, 32], [[32, 32, 32, 64], [32, 64, 64, 128]]),
        (64, 64, [16, 32], [[64, 128, 160, 256], [64, 128, 192, 256]]),
    ]
    for C, npoint, nsamples, mlps in cases:
        sa = PointnetSAModuleMSG(npoint=npoint, radii=[0.6, 1.2], nsamples=nsamples, mlps=[list(m) for m in mlps], bn=True).to(DEV).eval()
        B, N = 3, 1536
        pts = T(synth.dense_cloud(B, N, 6 + C, extent=5.0))
        feats = torch.randn(B, C, N, device=DEV) if C else None
        with torch.no_grad():
            idx = pu.farthest_point_sample(pts, npoint)
            new_xyz = pu.gather_operation(pts.transpose(1, 2).contiguous(), idx).transpose(1, 2).contiguous()
            widths = [fused.out_width(m) for m in sa.mlps]
            wide = Guard((B, sum(widths) + 5, npoint), torch.float32)          # 5 channels nobody writes, between and behind the slices
            c0 = 2
            for k, (r, ns) in enumerate(zip([0.6, 1.2], nsamples)):
                nb = pu.ball_query(r, ns, pts, new_xyz)
                assert fused.can_fuse(sa.mlps[k], npoint, ns, False, B, N)
                alone = fused.sa_mlp_fused(pts, new_xyz, feats, nb, sa.mlps[k])
                slot = wide.view[:, c0:c0 + widths[k]]
                got = fused.sa_mlp_fused(pts, new_xyz, feats, nb, sa.mlps[k], out=slot)
                assert got.data_ptr() == slot.data_ptr() and torch.equal(slot, alone)
                c0 += widths[k] + 1
            s = SENT[torch.float32]
            mask = torch.ones(sum(widths) + 5, dtype=torch.bool)
            c0 = 2
            for w in widths:
                mask[c0:c0 + w] = False
                c0 += w + 1
            assert wide.intact() and bool((wide.view[:, mask.to(DEV)] == s).all())
            # the module itself: concatenation in place == concatenation of the stand-alone results
            _, full, _ = sa(pts, feats, new_xyz=new_xyz)
            want = torch.cat([fused.sa_mlp_fused(pts, new_xyz, feats, pu.ball_query(r, ns, pts, new_xyz), m)
                              for r, ns, m in zip([0.6, 1.2], nsamples, sa.mlps)], dim=1)
            assert full.is_contiguous() and torch.equal(full, want)
    lib = L.load()
    layers = fused._packed_layers(sa.mlps[0], pts.device)
    nl = len(layers)
    wc = (ctypes.c_int * (nl + 1))(3 + C, *[l[2] for l in layers])
    warr = (ctypes.c_void_p * nl)(*[l[0].data_ptr() for l in layers])
    barr = (ctypes.c_void_p * nl)(*[l[1].data_ptr() for l in layers])
    nb = pu.ball_query(0.6, nsamples[0], pts, new_xyz)
    assert lib.jm_sa_mlp_forward_into(B, N, npoint, C, nsamples[0], L.dev(pts, torch.float32, "x"), L.dev(new_xyz, torch.float32, "c"),
                                      L.dev(feats, torch.float32, "f"), L.dev(nb, torch.int32, "i"), nl, wc, warr, barr,
                                      ctypes.c_void_p(wide.view.data_ptr()), widths[0] * npoint - 1, L.stream_ptr()) != 0
