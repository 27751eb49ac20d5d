"""Per-point Conv1d stacks on (B, C, n) tensors as one launch (csrc/conv1d_stack.hip).

Replaces, on the composed inference path, the kernel-size-1 `Conv1d (+ BatchNorm1d) (+ ReLU)` chains of the RPN heads
(jmodt/detection/modeling/rpn.py:34-58), the SharedMLP of the feature-propagation modules on cat[interpolated, skip]
(jmodt/ops/pointnet2/pointnet2_modules.py:139-153) and the hoisted first set-abstraction layer of
jmodt_amd/ops/pointnet2/fused.py — each of which is otherwise a batched GEMM + bias broadcast + ReLU pass per layer.
"""
import ctypes
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib as L
from .fusion import _pack

_f32 = torch.float32
# the 64-point-tile kernel (csrc/conv1d_stack64.hip) serves SINGLE-layer stacks — the hoisted first set-abstraction layers, where it
# measured faster than the 32-point kernel (13 against 20 us, 76 against 80 us); on multi-layer stacks it measured slower (RPN heads
# 189 against 187 us, FP1 181 against 160 us: one workgroup per CU, nothing hides its phases) and is not offered there
TILE64 = 1        # (a module constant, no environment switch.  0: never, 2: wherever the kernel's tiles fit — what
                  # tests/test_gpu_detector.py sets to check the kernel itself on multi-layer stacks against float64)


class PackedConv1dStack:
    """layers: [(W (cout, cin), b (cout) or None, relu)] with BatchNorm already folded; the FIRST layer's input is the
    channel concatenation [x0 (c0) | x1 (c1)] (c1 = 0: single operand); xyz1: x1 is point-major xyz (B, n, 3)."""

    def __init__(self, layers: Sequence[Tuple[torch.Tensor, Optional[torch.Tensor], bool]], c0: int, c1: int = 0,
                 xyz1: bool = False):
        assert 1 <= len(layers) <= 3 and layers[0][0].shape[1] == c0 + c1
        self.c0, self.c1, self.xyz1 = int(c0), int(c1), bool(xyz1)
        self.widths = [int(W.shape[0]) for W, _, _ in layers]
        self.relu = [int(bool(r)) for _, _, r in layers]
        W0, b0, _ = layers[0]
        dev = W0.device
        zeros = lambda n: torch.zeros(n, dtype=_f32, device=dev)      # noqa: E731
        self.w0a, self.b0 = _pack(W0[:, :c0], b0 if b0 is not None else zeros(W0.shape[0]))
        self.w0b = _pack(W0[:, c0:], None)[0] if c1 else None
        self.w, self.b = [self.w0a], [self.b0]
        for W, b, _ in layers[1:]:
            wp, bp = _pack(W, b if b is not None else zeros(W.shape[0]))
            self.w.append(wp)
            self.b.append(bp)
        nl = len(layers)
        self._widths_c = (ctypes.c_int * nl)(*self.widths)
        self._relu_c = (ctypes.c_int * nl)(*self.relu)
        self._w_c = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.w])
        self._b_c = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.b])
        # the 64-point-tile kernel's layout (csrc/conv1d_stack64.hip): layer 0 on the concatenated input, B-operand order
        lib = L.load()
        self.w64, self.b64 = [], []
        for W, b, _ in layers:
            Wc = W.detach().to(_f32).contiguous()
            n_out, k = Wc.shape
            wp = torch.empty((int(lib.jm_conv1d_stack64_packed_elems(n_out, k)),), dtype=_f32, device=dev)
            bp = torch.empty(((n_out + 31) // 32 * 32,), dtype=_f32, device=dev)
            bb = b.detach().to(_f32).contiguous() if b is not None else None
            L.check(lib.jm_conv1d_stack64_pack(n_out, k, L.dev(Wc, _f32, "W"), k, L.dev(bb, _f32, "b") if bb is not None else None,
                                               ctypes.c_void_p(wp.data_ptr()), ctypes.c_void_p(bp.data_ptr()), L.stream_ptr()), "conv1d_stack64_pack")
            self.w64.append(wp)
            self.b64.append(bp)
        self._w64_c = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.w64])
        self._b64_c = (ctypes.c_void_p * nl)(*[t.data_ptr() for t in self.b64])

    def supported(self, B: int, n: int) -> bool:
        return self.supported64(B, n) or bool(L.load().jm_conv1d_stack_supported(B, n, self.c0, self.c1, int(self.xyz1), len(self.widths),
                                                                                 self._widths_c))

    def supported64(self, B: int, n: int) -> bool:
        return (TILE64 == 2 or (TILE64 == 1 and len(self.widths) == 1)) and bool(L.load().jm_conv1d_stack64_supported(B, n, self.c0, self.c1, int(self.xyz1), len(self.widths), self._widths_c))

    @torch.no_grad()
    def __call__(self, x0: torch.Tensor, x1: Optional[torch.Tensor] = None, point_major: bool = False) -> torch.Tensor:
        """x0 (B, c0, n), x1 (B, c1, n) — or (B, n, 3) with xyz1 — -> (B, widths[-1], n), or (B, n, widths[-1]) with
        point_major (the set-abstraction kernels gather whole rows of that layout)"""
        x0 = x0.to(_f32).contiguous()
        B, c0, n = x0.shape
        assert c0 == self.c0 and (x1 is None) == (self.c1 == 0)
        if x1 is not None:
            x1 = x1.to(_f32).contiguous()
            assert tuple(x1.shape) == ((B, n, 3) if self.xyz1 else (B, self.c1, n))
        out = torch.empty((B, n, self.widths[-1]) if point_major else (B, self.widths[-1], n), dtype=_f32, device=x0.device)
        if self.supported64(B, n):
            L.check(L.load().jm_conv1d_stack64_forward(
                B, n, self.c0, L.dev(x0, _f32, "x0"), self.c1, L.dev(x1, _f32, "x1") if x1 is not None else None, int(self.xyz1),
                len(self.widths), self._widths_c, self._w64_c, self._b64_c, self._relu_c, int(point_major), ctypes.c_void_p(out.data_ptr()),
                L.stream_ptr()), "conv1d_stack64")
            return out
        L.check(L.load().jm_conv1d_stack_forward(
            B, n, self.c0, L.dev(x0, _f32, "x0"), self.c1, L.dev(x1, _f32, "x1") if x1 is not None else None, int(self.xyz1),
            len(self.widths), self._widths_c, L.dev(self.w0a, _f32, "w0a"),
            L.dev(self.w0b, _f32, "w0b") if self.w0b is not None else None, self._w_c, self._b_c, self._relu_c,
            int(point_major), ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "conv1d_stack")
        return out


def points_linear_supported(B: int, n: int, k1: int, k2: int, n_out: int) -> bool:
    return bool(L.load().jm_points_linear_supported(int(B), int(n), int(k1), int(k2), int(n_out)))


@torch.no_grad()
def points_linear(x1: torch.Tensor, W: torch.Tensor, bias: Optional[torch.Tensor], act: int = 0, x2: Optional[torch.Tensor] = None,
                  rowscale: Optional[torch.Tensor] = None, rowscale_stride: int = 1, out_rows: Optional[int] = None) -> torch.Tensor:
    """act(W [x1 ; x2] + bias) (* rowscale per point) on (B, C, n) tensors with few points — one launch of independent waves
    (csrc/points_gemm.hip: the coarse end of the backbone, where the 32-point-tile kernels cannot fill the machine and the
    library GEMMs are slow).  W (n_out, k1 + k2) contiguous; act 0 none / 1 ReLU / 2 tanh / 3 sigmoid; out_rows = ld: the output as
    point-major rows (B n, ld) instead of (B, n_out, n)."""
    B, k1, n = x1.shape
    k2 = 0 if x2 is None else x2.shape[1]
    n_out = W.shape[0]
    if W.shape[1] != k1 + k2:
        raise ValueError(f"points_linear: weight of {W.shape[1]} input channels for operands of {k1} + {k2}")
    if out_rows:
        out = torch.empty((B * n, int(out_rows)), dtype=_f32, device=x1.device)
    else:
        out = torch.empty((B, n_out, n), dtype=_f32, device=x1.device)
    L.check(L.load().jm_points_linear(
        B, n, k1, L.dev(x1, _f32, "x1"), k2, L.dev(x2, _f32, "x2") if x2 is not None else None, n_out, L.dev(W, _f32, "W"), W.shape[1],
        L.dev(bias, _f32, "bias") if bias is not None else None, int(act), L.dev(rowscale, _f32, "rowscale") if rowscale is not None else None,
        int(rowscale_stride), 1 if out_rows else 0, int(out_rows or 0), ctypes.c_void_p(out.data_ptr()), L.stream_ptr()), "points_linear")
    return out
