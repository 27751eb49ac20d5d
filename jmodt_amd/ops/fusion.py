"""LI-Fusion point -> image gather on the gfx950 kernel.

Mirror of `feature_gather` (jmodt/detection/modeling/backbone.py:79-89):
    F.grid_sample(feature_map.float(), xy.unsqueeze(1), align_corners=True).squeeze(2)
bilinear, zero padding.  The feature map is consumed through its strides, so a
`torch.channels_last` map (what MIOpen prefers for the image branch's convolutions anyway) is
gathered with 16-byte tap reads and without a layout copy.
"""
from typing import Optional

import torch
from torch.autograd import Function

from .. import _lib as L

_f32 = torch.float32


def _dense_strides(t: torch.Tensor):
    """a (B,C,H,W) tensor whose memory is a permutation of a dense block (NCHW or channels-last)"""
    if t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last):
        return t
    return t.contiguous()


class _FeatureGather(Function):
    @staticmethod
    def forward(ctx, feature_map: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
        fm = _dense_strides(feature_map.float())
        xy = xy.float().contiguous()
        B, C, H, W = fm.shape
        N = xy.shape[1]
        out = torch.empty((B, C, N), dtype=_f32, device=fm.device)
        if not fm.is_cuda:
            raise RuntimeError("feature_gather: feature_map must be a GPU tensor (no CPU path)")
        sb, sc, sh, sw = fm.stride()
        import ctypes
        L.check(L.load().jm_feature_gather(B, C, H, W, N, ctypes.c_void_p(fm.data_ptr()), sb, sc, sh, sw,
                                           L.dev(xy, _f32, "xy"), L.dev(out, _f32, "out"), L.stream_ptr()),
                "feature_gather")
        ctx.save_for_backward(xy)
        ctx.geom = (B, C, H, W, N, fm.is_contiguous())
        return out

    @staticmethod
    def backward(ctx, grad_out):
        (xy,) = ctx.saved_tensors
        B, C, H, W, N, nchw = ctx.geom
        fmt = torch.contiguous_format if nchw else torch.channels_last
        grad_map = torch.empty((B, C, H, W), dtype=_f32, device=grad_out.device, memory_format=fmt).zero_()
        sb, sc, sh, sw = grad_map.stride()
        g = grad_out.contiguous()
        import ctypes
        L.check(L.load().jm_feature_gather_grad(B, C, H, W, N, L.dev(g, _f32, "grad_out"), L.dev(xy, _f32, "xy"),
                                                ctypes.c_void_p(grad_map.data_ptr()), sb, sc, sh, sw, L.stream_ptr()),
                "feature_gather.backward")
        return grad_map, None


def feature_gather(feature_map: torch.Tensor, xy: torch.Tensor) -> torch.Tensor:
    """feature_map (B, C, H, W), xy (B, N, 2) normalised to [-1, 1] -> (B, C, N)"""
    return _FeatureGather.apply(feature_map, xy)


def _pack(W: torch.Tensor, b):
    """(cout, cin) weight (+ bias) -> the MFMA kernels' device layout (jm_sa_mlp_pack)"""
    import ctypes
    lib = L.load()
    W = W.detach().to(_f32).contiguous()
    cout, cin = W.shape
    wp = torch.empty((lib.jm_sa_mlp_packed_weight_elems(cout, cin, 0),), dtype=_f32, device=W.device)
    bp = torch.empty((lib.jm_sa_mlp_packed_bias_elems(cout),), dtype=_f32, device=W.device)
    bb = b.detach().to(_f32).contiguous() if b is not None else None
    L.check(lib.jm_sa_mlp_pack(cout, cin, 0, L.dev(W, _f32, "W"), L.dev(bb, _f32, "b") if bb is not None else None,
                               ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bp.data_ptr()), L.stream_ptr()), "sa_mlp_pack")
    return wp, bp


class PackedAttentionFusion:
    """weights of one AttentionFusion block (backbone.py:35-81) in the layout of jm_attention_fusion_forward.
    W_img / b_img and W_fuse / b_fuse are the 1x1 convolutions with their eval-mode BatchNorm already folded."""

    def __init__(self, fc1_w, fc1_b, fc2_w, fc2_b, fc3_w, fc3_b, W_img, b_img, W_fuse, b_fuse):
        self.rc, self.ic = fc1_w.shape
        self.pc = fc2_w.shape[1]
        self.oc = W_fuse.shape[0]
        assert W_img.shape == (self.pc, self.ic) and W_fuse.shape[1] == 2 * self.pc
        self.w1, self.b12 = _pack(fc1_w, fc1_b.detach() + fc2_b.detach())
        self.w2, _ = _pack(fc2_w, None)
        self.wi, self.bi = _pack(W_img, b_img)
        self.wfp, self.bf = _pack(W_fuse[:, :self.pc], b_fuse)
        self.wfg, _ = _pack(W_fuse[:, self.pc:], None)
        self.w3 = fc3_w.detach().to(_f32).reshape(-1).contiguous()
        self.b3 = float(fc3_b.detach().reshape(-1)[0].item())

    def supported(self, B: int, n: int) -> bool:
        return bool(L.load().jm_attention_fusion_supported(B, n, self.ic, self.pc, self.rc, self.oc))

    @torch.no_grad()
    def __call__(self, point_feats: torch.Tensor, img_feats: torch.Tensor) -> torch.Tensor:
        """point_feats (B, pc, n), img_feats (B, ic, n) -> (B, oc, n)"""
        import ctypes
        P, I = point_feats.to(_f32).contiguous(), img_feats.to(_f32).contiguous()
        B, _, n = P.shape
        out = torch.empty((B, self.oc, n), dtype=_f32, device=P.device)
        L.check(L.load().jm_attention_fusion_forward(
            B, n, self.ic, self.pc, self.rc, self.oc, L.dev(I, _f32, "img_feats"), L.dev(P, _f32, "point_feats"),
            L.dev(self.w1, _f32, "w1"), L.dev(self.w2, _f32, "w2"), L.dev(self.b12, _f32, "b12"), L.dev(self.w3, _f32, "w3"),
            self.b3, L.dev(self.wi, _f32, "wi"), L.dev(self.bi, _f32, "bi"), L.dev(self.wfp, _f32, "wfp"),
            L.dev(self.wfg, _f32, "wfg"), L.dev(self.bf, _f32, "bf"), ctypes.c_void_p(out.data_ptr()), L.stream_ptr()),
            "attention_fusion")
        return out


class PackedImageFusion:
    """the final LI-Fusion image feature at the points (backbone.py:187-195) without the full-resolution map:
    csrc/image_fusion.hip.  Built from the per-level composed weights wc_i (C_i, q, k_i, k_i) = the level's
    ConvTranspose2d weight contracted with its slice of the BatchNorm-folded 1x1 fusion convolution, and the folded
    bias (deconvolution biases included) — `DetectAffinityEngine._composed_image_fusion` produces both."""

    def __init__(self, composed_weights, bias, strides):
        import ctypes
        lib = L.load()
        self.strides = [int(k) for k in strides]
        self.channels = [int(w.shape[0]) for w in composed_weights]
        self.q = int(composed_weights[0].shape[1])
        self.ok = self.q <= 32 and len(self.strides) <= 4 and all(c >= 16 and c % 16 == 0 for c in self.channels) \
            and all(k in (1, 2, 4, 8, 16) for k in self.strides)
        if not self.ok:
            return
        dev = composed_weights[0].device
        self.packed = []
        for wc, k, c in zip(composed_weights, self.strides, self.channels):
            wc = wc.detach().to(_f32).contiguous()
            assert wc.shape == (c, self.q, k, k)
            wp = torch.empty((lib.jm_image_fusion_packed_elems(c, k),), dtype=_f32, device=dev)
            L.check(lib.jm_image_fusion_pack(c, self.q, k, L.dev(wc, _f32, "wc"), ctypes.c_void_p(wp.data_ptr()), L.stream_ptr()),
                    "image_fusion_pack")
            self.packed.append(wp)
        self.bias = torch.zeros(32, dtype=_f32, device=dev)
        self.bias[:self.q] = bias.detach().to(_f32)

    def supported(self, maps, h: int, w: int) -> bool:
        return self.ok and all(m.is_cuda and m.dtype == _f32 and m.is_contiguous(memory_format=torch.channels_last)
                               and m.shape[2] * k == h and m.shape[3] * k == w for m, k in zip(maps, self.strides))

    @torch.no_grad()
    def __call__(self, maps, xy: torch.Tensor, h: int, w: int) -> torch.Tensor:
        """maps[i] (B, C_i, h / k_i, w / k_i) channels-last, xy (B, N, 2) in [-1, 1] -> (B, q, N)"""
        import ctypes
        lib = L.load()
        xy = xy.to(_f32).contiguous()
        B, N = xy.shape[:2]
        nl = len(maps)
        out = torch.empty((B, self.q, N), dtype=_f32, device=xy.device)
        ws_bytes = lib.jm_image_fusion_gather_workspace_bytes(B, N)
        ws = torch.empty((ws_bytes + 256,), dtype=torch.uint8, device=xy.device)
        base = (ws.data_ptr() + 255) // 256 * 256
        ch = (ctypes.c_int * nl)(*self.channels)
        st = (ctypes.c_int * nl)(*self.strides)
        mp = (ctypes.c_void_p * nl)(*[m.data_ptr() for m in maps])
        wp = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.packed])
        L.check(lib.jm_image_fusion_gather(B, N, h, w, self.q, nl, ch, st, mp, wp, L.dev(self.bias, _f32, "bias"),
                                           L.dev(xy, _f32, "xy"), ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(base),
                                           ws_bytes, L.stream_ptr()), "image_fusion_gather")
        return out


@torch.no_grad()
def bias_relu_(x: torch.Tensor, bias: torch.Tensor) -> torch.Tensor:
    """x (B, C, H, W) channels-last, in place: relu(x + bias[c]) in ONE pass (csrc/elementwise.hip)"""
    import ctypes
    if not (x.is_cuda and x.dtype == _f32 and x.is_contiguous(memory_format=torch.channels_last) and x.shape[1] % 4 == 0):
        return torch.relu_(x.add_(bias.view(1, -1, 1, 1)))
    b = bias.detach().to(_f32).contiguous()
    L.check(L.load().jm_bias_relu_channels_last(x.numel(), x.shape[1], ctypes.c_void_p(x.data_ptr()), L.dev(b, _f32, "bias"),
                                                L.stream_ptr()), "bias_relu")
    return x


@torch.no_grad()
def conv3x3_rgb_bias_relu(image: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, packed=None) -> torch.Tensor:
    """image (B, 3, H, W) NCHW, weight (cout, 3, 3, 3), bias (cout) -> relu(conv3x3(image, padding 1) + bias) as a
    channels-last (B, cout, H, W) tensor, one pass (csrc/conv_rgb.hip).  `packed` = the (27, cout) tap-major weight when
    the caller caches it (`pack_rgb_weight`)."""
    import ctypes
    x = image.to(_f32).contiguous()
    B, C, H, W = x.shape
    assert C == 3 and tuple(weight.shape[1:]) == (3, 3, 3)
    cout = weight.shape[0]
    wt = packed if packed is not None else pack_rgb_weight(weight)
    b = bias.detach().to(_f32).contiguous()
    out = torch.empty((B, cout, H, W), dtype=_f32, device=x.device, memory_format=torch.channels_last)
    L.check(L.load().jm_conv3x3_rgb_bias_relu(B, H, W, cout, L.dev(x, _f32, "image"), L.dev(wt, _f32, "weight"), L.dev(b, _f32, "bias"),
                                              ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "conv3x3_rgb")
    return out


def wino_supported(cin: int, cout: int) -> bool:
    return bool(L.load().jm_conv3x3_wino_supported(int(cin), int(cout)))


@torch.no_grad()
def pack_wino_weight(weight: torch.Tensor) -> torch.Tensor:
    """(cout, cin, 3, 3) -> the 16 * cin * cout floats jm_conv3x3_wino_bias_relu streams: U = G g G^T per (cin, cout) pair in
    MFMA operand order (csrc/conv_wino.hip); made once per weight"""
    import ctypes
    w = weight.detach().to(_f32).contiguous()
    cout, cin = w.shape[0], w.shape[1]
    assert tuple(w.shape[2:]) == (3, 3)
    lib = L.load()
    packed = torch.empty(lib.jm_conv3x3_wino_packed_elems(cin, cout), dtype=_f32, device=w.device)
    L.check(lib.jm_conv3x3_wino_pack(cin, cout, L.dev(w, _f32, "weight"), ctypes.c_void_p(packed.data_ptr()), L.stream_ptr()),
            "conv3x3_wino_pack")
    return packed


@torch.no_grad()
def conv3x3_wino_bias_relu(x: torch.Tensor, packed: torch.Tensor, bias: Optional[torch.Tensor], cout: int, relu: bool = True) -> torch.Tensor:
    """x (B, cin, H, W) channels-last, packed = pack_wino_weight(weight (cout, cin, 3, 3)) -> relu(conv3x3(x, padding 1) + bias)
    as a channels-last (B, cout, H, W) tensor: fused Winograd F(2x2, 3x3), fp32 (csrc/conv_wino.hip)"""
    import ctypes
    assert x.is_cuda and x.dtype == _f32 and x.is_contiguous(memory_format=torch.channels_last)
    B, cin, H, W = x.shape
    assert packed.numel() == 16 * cin * cout
    b = bias.detach().to(_f32).contiguous() if bias is not None else None
    out = torch.empty((B, cout, H, W), dtype=_f32, device=x.device, memory_format=torch.channels_last)
    L.check(L.load().jm_conv3x3_wino_bias_relu(B, H, W, cin, cout, ctypes.c_void_p(x.data_ptr()), L.dev(packed, _f32, "packed"),
                                               L.dev(b, _f32, "bias") if b is not None else None, int(relu),
                                               ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "conv3x3_wino")
    return out


def pack_rgb_weight(weight: torch.Tensor) -> torch.Tensor:
    """(cout, 3, 3, 3) -> (27, cout) tap-major, the layout jm_conv3x3_rgb_bias_relu reads with scalar loads"""
    return weight.detach().to(_f32).permute(1, 2, 3, 0).reshape(27, weight.shape[0]).contiguous()
