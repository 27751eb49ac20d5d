"""Pairwise link / start-end affinity head on the fp32 matrix cores.

Mirrors the two pieces of the reference that the north star names:
  * the heads themselves — `link_layer`, `se_layer` of jmodt/detection/modeling/rcnn.py:91-111:
    Conv1d(512,512)+ReLU, Dropout(DP_RATIO=0), Conv1d(512,512)+ReLU, Conv1d(512,1), bias, no BN
    (config.py:166-169); child indices 0/2/3 and `.conv.{weight,bias}` names are kept so a
    reference checkpoint's `rcnn_net.link_layer.*` / `rcnn_net.se_layer.*` entries load as is;
  * the inference-time affinity of jmodt/tracking/tracker.py:81-112:
        cor = |pred_i - det_j|;  S = link(cor);  A = (softmax(S,1) + softmax(S,0)) / 2
        start = se(cor.mean(0)),  end = se(cor.mean(1))
    evaluated by ONE C-ABI call that never materialises the (P*D, 512) pair tensor.
"""
import ctypes
from typing import Optional, Tuple

import torch
import torch.nn as nn

from .. import _lib as L
from .pointnet2 import pytorch_utils as pt_utils
from .pointnet2.pyramid import side_stream

_f32 = torch.float32


def make_affinity_mlp(channel_in: int = 512, fc=(512, 512), dp_ratio: float = 0.0, use_bn: bool = False) -> nn.Sequential:
    """builds link_layer / se_layer exactly as rcnn.py:91-111 does (xavier-normal weights, zero
    bias: rcnn.py:116-134)"""
    layers = []
    pre = channel_in
    for width in fc:
        layers.append(pt_utils.Conv1d(pre, width, bn=use_bn))
        pre = width
    layers.append(pt_utils.Conv1d(pre, 1, activation=None))
    if dp_ratio >= 0:
        layers.insert(1, nn.Dropout(dp_ratio))
    head = nn.Sequential(*layers)
    for m in head.modules():
        if isinstance(m, nn.Conv1d):
            nn.init.xavier_normal_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
    return head


def _pack(head: nn.Sequential):
    """(Mlp3 struct, tensors kept alive) from a 3-conv head; supports exactly the reference layout"""
    convs = [m for m in head.modules() if isinstance(m, nn.Conv1d)]
    if len(convs) != 3 or any(isinstance(m, (nn.BatchNorm1d,)) for m in head.modules()):
        raise NotImplementedError("affinity kernel supports the reference head: 3 Conv1d(k=1), no BN")
    for m in head.modules():
        if isinstance(m, nn.Dropout) and m.p != 0 and head.training:
            raise NotImplementedError("affinity kernel: dropout p>0 in training mode")
    c1, c2, c3 = convs
    if c3.out_channels != 1 or any(c.kernel_size != (1,) for c in convs):
        raise NotImplementedError("affinity kernel: kernel_size must be 1 and the last layer 1-wide")
    keep = []

    def dp(t):
        t = t.detach().to(_f32).contiguous()
        if not t.is_cuda:
            raise RuntimeError("affinity head weights must live on the GPU (jmodt_amd has no CPU path)")
        keep.append(t)
        return ctypes.c_void_p(t.data_ptr())

    def bias(c):
        return c.bias if c.bias is not None else torch.zeros(c.out_channels, device=c.weight.device)

    s = L.Mlp3(c1.in_channels, c1.out_channels, c2.out_channels, dp(c1.weight.view(c1.out_channels, -1)), dp(bias(c1)),
               dp(c2.weight.view(c2.out_channels, -1)), dp(bias(c2)), dp(c3.weight.view(-1)), dp(bias(c3)))
    return s, keep


@torch.no_grad()
def pairwise_affinity(pred_features: torch.Tensor, det_features: torch.Tensor, link_model: nn.Sequential,
                      se_model: Optional[nn.Sequential] = None, return_raw: bool = False,
                      overlap_start_end: bool = False) -> Tuple[torch.Tensor, ...]:
    """pred_features (P, C), det_features (D, C) ->
        link_scores (P, D)  dual-softmax affinity             (tracker.py:86-89)
        start_logits (D), end_logits (P)  raw se outputs      (tracker.py:105-110 applies
                                                               w_se * sigmoid on top)
    (+ raw link scores (P, D) when return_raw)."""
    lib = L.load()
    pf = pred_features.to(_f32).contiguous()
    df = det_features.to(_f32).contiguous()
    P, D = pf.shape[0], df.shape[0]
    dev = pf.device
    link, keep1 = _pack(link_model)
    se, keep2 = _pack(se_model) if se_model is not None else (None, [])
    A = torch.empty((P, D), dtype=_f32, device=dev)
    raw = torch.empty((P, D), dtype=_f32, device=dev) if return_raw else None
    se_out = torch.empty((D + P,), dtype=_f32, device=dev)   # contiguous: the kernel writes logits in place
    start, end = se_out[:D], se_out[D:]
    link_p = ctypes.byref(link)
    main = torch.cuda.current_stream(dev)
    if se is not None:
        # the start/end head is a short latency-bound chain ((P+D) rows).  overlap_start_end runs it on a side stream owned
        # HERE, under the link head's GEMMs; OFF by default: inside the composed engine that fork / join costs far more than the
        # 0.1 ms it hides once every stream really has a hardware queue of its own (GPU_MAX_HW_QUEUES = 8: 513 frames/s with the
        # fork, 625 without; at the default of 4 queues the side stream happened to share the main stream's queue: 609 / 631).
        # (fork / join with stream waits; the library itself keeps no streams)
        se_p = ctypes.byref(se)
        side = side_stream(dev, 2) if overlap_start_end else main
        se_bytes = lib.jm_affinity_start_end_workspace_bytes(P, D, se_p)
        se_ws = torch.empty((max(se_bytes, 16),), dtype=torch.uint8, device=dev)
        if side is not main:
            side.wait_stream(main)
            for t in (pf, df, se_out, se_ws, *keep2):
                t.record_stream(side)
        with torch.cuda.stream(side):
            L.check(lib.jm_affinity_start_end(P, D, L.dev(pf, _f32, "pred_features"), L.dev(df, _f32, "det_features"), se_p,
                                              ctypes.c_void_p(start.data_ptr()), ctypes.c_void_p(end.data_ptr()),
                                              ctypes.c_void_p(se_ws.data_ptr()), se_bytes, L.stream_ptr()),
                    "pairwise_affinity(start/end)")
    ws_bytes = lib.jm_affinity_workspace_bytes(P, D, link_p, None)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    L.check(lib.jm_affinity_forward(P, D, L.dev(pf, _f32, "pred_features"), L.dev(df, _f32, "det_features"), link_p,
                                    None, ctypes.c_void_p(raw.data_ptr()) if raw is not None else None,
                                    ctypes.c_void_p(A.data_ptr()), None, None, ctypes.c_void_p(ws.data_ptr()), ws_bytes,
                                    L.stream_ptr()), "pairwise_affinity")
    if se is not None and side is not main:
        main.wait_stream(side)
    out = (A, start, end) if se_model is not None else (A,)
    return out + ((raw,) if return_raw else ())


@torch.no_grad()
def pairwise_affinity_batched(pred_features: torch.Tensor, det_features: torch.Tensor, link_model: nn.Sequential,
                              se_model: Optional[nn.Sequential] = None, return_raw: bool = False,
                              overlap_start_end: bool = False, split_bf16: bool = False) -> Tuple[torch.Tensor, ...]:
    """nb independent problems at once: pred_features (nb, P, C), det_features (nb, D, C) ->
    link_scores (nb, P, D), start_logits (nb, D), end_logits (nb, P) (+ raw (nb, P, D)): `pairwise_affinity` for every
    frame pair of a batch as ONE GEMM chain over nb*P*D pair rows (jm_affinity_forward_batched)"""
    lib = L.load()
    pf = pred_features.to(_f32).contiguous()
    df = det_features.to(_f32).contiguous()
    nb, P, _ = pf.shape
    D = df.shape[1]
    dev = pf.device
    link, keep1 = _pack(link_model)
    se, keep2 = _pack(se_model) if se_model is not None else (None, [])
    A = torch.empty((nb, P, D), dtype=_f32, device=dev)
    raw = torch.empty((nb, P, D), dtype=_f32, device=dev) if return_raw else None
    se_out = torch.empty((nb, D + P), dtype=_f32, device=dev)
    main = torch.cuda.current_stream(dev)
    side = main
    if se is not None:
        se_p = ctypes.byref(se)
        side = side_stream(dev, 2) if overlap_start_end else main
        se_bytes = lib.jm_affinity_start_end_batched_workspace_bytes(nb, P, D, se_p)
        se_ws = torch.empty((max(se_bytes, 16),), dtype=torch.uint8, device=dev)
        if side is not main:
            side.wait_stream(main)
            for t in (pf, df, se_out, se_ws, *keep2):
                t.record_stream(side)
        with torch.cuda.stream(side):
            L.check(lib.jm_affinity_start_end_batched(nb, P, D, L.dev(pf, _f32, "pred_features"), L.dev(df, _f32, "det_features"),
                                                      se_p, ctypes.c_void_p(se_out.data_ptr()), ctypes.c_void_p(se_ws.data_ptr()),
                                                      se_bytes, L.stream_ptr()), "pairwise_affinity_batched(start/end)")
    link_p = ctypes.byref(link)
    if split_bf16:
        # EXPERIMENTAL (csrc/affinity_x3.hip): the link head's products on the bf16 matrix pipe as 3-term splits
        raw = torch.empty((nb, P, D), dtype=_f32, device=dev)
        xb = lib.jm_affinity_x3_workspace_bytes(nb, P, D, link_p)
        xws = torch.empty((max(xb, 16),), dtype=torch.uint8, device=dev)
        L.check(lib.jm_affinity_link_scores_x3(nb, P, D, L.dev(pf, _f32, "pred_features"), L.dev(df, _f32, "det_features"), link_p,
                                               ctypes.c_void_p(raw.data_ptr()), ctypes.c_void_p(xws.data_ptr()), xb, L.stream_ptr()),
                "pairwise_affinity_batched(x3)")
        stats = torch.empty((2 * nb * (P + D),), dtype=_f32, device=dev)
        L.check(lib.jm_affinity_dual_softmax_batched(nb, P, D, L.dev(raw, _f32, "raw"), ctypes.c_void_p(A.data_ptr()),
                                                     ctypes.c_void_p(stats.data_ptr()), L.stream_ptr()), "pairwise_affinity_batched(softmax)")
        if side is not main:
            main.wait_stream(side)
        out = (A, se_out[:, :D], se_out[:, D:]) if se_model is not None else (A,)
        return out + ((raw,) if return_raw else ())
    ws_bytes = lib.jm_affinity_batched_workspace_bytes(nb, P, D, link_p)
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=dev)
    L.check(lib.jm_affinity_forward_batched(nb, P, D, L.dev(pf, _f32, "pred_features"), L.dev(df, _f32, "det_features"), link_p,
                                            ctypes.c_void_p(raw.data_ptr()) if raw is not None else None,
                                            ctypes.c_void_p(A.data_ptr()), ctypes.c_void_p(ws.data_ptr()), ws_bytes,
                                            L.stream_ptr()), "pairwise_affinity_batched")
    if side is not main:
        main.wait_stream(side)
    out = (A, se_out[:, :D], se_out[:, D:]) if se_model is not None else (A,)
    return out + ((raw,) if return_raw else ())


@torch.no_grad()
def mlp3_forward(x: torch.Tensor, head: nn.Sequential) -> torch.Tensor:
    """the same head on materialised rows x (M, C) -> (M)   (e.g. rcnn.py:272-285 start/end features)"""
    lib = L.load()
    x = x.to(_f32).contiguous()
    mlp, keep = _pack(head)
    y = torch.empty((x.shape[0],), dtype=_f32, device=x.device)
    ws_bytes = lib.jm_mlp3_workspace_bytes(x.shape[0], ctypes.byref(mlp))
    ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=x.device)
    L.check(lib.jm_mlp3_forward(x.shape[0], L.dev(x, _f32, "x"), ctypes.byref(mlp), ctypes.c_void_p(y.data_ptr()),
                                ctypes.c_void_p(ws.data_ptr()), ws_bytes, L.stream_ptr()), "mlp3_forward")
    return y


@torch.no_grad()
def linear_rows(x: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, relu: bool) -> torch.Tensor:
    """x (M, K) rows, weight (N, K), bias (N) -> act(x W^T + b) (M, N) as one launch (jm_linear_rows): the small-M dense
    layers of the RCNN heads"""
    x = x.to(_f32).contiguous()
    W = weight.detach().to(_f32).contiguous()
    b = bias.detach().to(_f32).contiguous()
    M, K = x.shape
    N = W.shape[0]
    y = torch.empty((M, N), dtype=_f32, device=x.device)
    L.check(L.load().jm_linear_rows(M, K, N, L.dev(x, _f32, "x"), L.dev(W, _f32, "weight"), L.dev(b, _f32, "bias"),
                                    ctypes.c_void_p(y.data_ptr()), int(bool(relu)), L.stream_ptr()), "linear_rows")
    return y
