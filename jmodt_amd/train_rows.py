"""The joint-mode training forward on ROW-major activations: the differentiable composition of the detector whose forward AND
backward are the hand-written kernels of csrc/rows_gemm.hip / csrc/rows_ops.hip (ops/rows.py), not torch autograd over the
reference's (B, C, npoint, nsample) tensors.

Same network and parameter containers as jmodt_amd/train_joint.py's operator route (point_rcnn.py:24-70 in TRAIN mode:
backbone.py:159-196, rpn.py:71-87, rcnn.py:158-202), same outputs, with
  * every per-point / per-(centre, neighbour) activation a (rows, channels) tensor — set abstraction on the DISTINCT rows of the
    ball-query groups (device-side plan, nothing of size npoint x nsample x C exists), feature propagation, the LI-Fusion
    gather + attention block, the RPN / RCNN heads, the RCNN input MLPs;
  * BatchNorm in EVAL mode (frozen running statistics: cfg.RPN.FIXED-style, point_rcnn.py:29-30), folded into the neighbouring
    weight with plain differentiable torch arithmetic on the parameters — d(loss)/d(gamma, beta) follow by autograd from the folded
    weight's gradient (`BnFold`: all scales of the network from four concatenated vectors, not four launches per layer);
  * the FPS pyramid, the ball queries and the 3-NN search (coordinates only, no gradient) on the engine's side stream, started
    a step ahead when the caller announces the next batch (FpsPyramid);
  * the image branch's 3x3 convolutions and the deconvolution pyramid on MIOpen's autograd in channels-last memory (not a
    SURVEY.md §8 row), the final fusion map composed per level as detector._image_fusion_map does.
"""
import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from .ops import rows as R
from .ops.pointnet2 import pointnet2_utils
from .ops.pointnet2.pyramid import FpsPyramid, side_stream
from .ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
from .profile import prof


_pairs_cache = {}      # id(root) -> (registration epoch, weakref to root, pairs)


def bn_pairs(root: nn.Module):
    """[(convolution / Linear module, BatchNorm module)] of every BatchNorm of the detector, found by structure: pytorch_utils units
    (`unit.conv`, `unit.bn.bn`), IA_Layer.conv1 = Sequential(Conv1d, BatchNorm1d, ReLU), AttentionFusion (conv1, bn1), the image
    blocks (conv1, bn1), (image_fusion_conv, image_fusion_bn) — backbone.py:16-32,35-81,150-157, pytorch_utils.py:36-102.
    The walk (2 ms of host time for the detector) is repeated only after a module / parameter registration (_registry.EPOCH)."""
    import weakref
    from ._registry import EPOCH
    hit = _pairs_cache.get(id(root))
    if hit is not None and hit[0] == EPOCH[0] and hit[1]() is root:
        return hit[2]
    pairs = _bn_pairs_walk(root)
    nbn = sum(1 for m in root.modules() if isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)))
    assert len(pairs) == len({id(b) for _, b in pairs}) == nbn, "a BatchNorm of the network is not paired with its convolution (train_rows.bn_pairs)"
    _pairs_cache[id(root)] = (EPOCH[0], weakref.ref(root), pairs)
    return pairs


def _bn_pairs_walk(root: nn.Module):
    pairs = []
    for m in root.modules():
        if hasattr(m, "conv") and hasattr(m, "bn") and hasattr(m.bn, "bn"):
            pairs.append((m.conv, m.bn.bn))
        elif isinstance(m, nn.Sequential) and len(m) >= 2 and isinstance(m[0], (nn.Conv1d, nn.Conv2d)) and isinstance(m[1], (nn.BatchNorm1d, nn.BatchNorm2d)):
            pairs.append((m[0], m[1]))
        elif hasattr(m, "conv1") and hasattr(m, "bn1") and isinstance(m.bn1, (nn.BatchNorm1d, nn.BatchNorm2d)):
            pairs.append((m.conv1, m.bn1))
        elif hasattr(m, "image_fusion_conv") and hasattr(m, "image_fusion_bn"):
            pairs.append((m.image_fusion_conv, m.image_fusion_bn))
    return pairs


class BnFold:
    """eval-mode BatchNorm folded into the preceding convolution for the WHOLE network with a handful of launches:
    s = gamma / sqrt(running_var + eps), t = beta - running_mean * s from four concatenated vectors, every folded weight
    W * s[:, None] from ONE multi-tensor kernel (ops/rows.fold_all) whose backward — d(W) = d(Wf) * s, d(s) = rowsum(d(Wf) * W) —
    is one more.  Differentiable w.r.t. W, gamma, beta (the running statistics are buffers)."""

    def __init__(self, root, pairs=None):
        """root: the module whose BatchNorms are folded — or `pairs`, an explicit [(convolution, BatchNorm)] list (a section of
        the network: tools/quarantine/train_graphs.py)"""
        pairs = bn_pairs(root) if pairs is None else list(pairs)
        self._slots = {}
        if not pairs:
            return
        bns = [bn for _, bn in pairs]
        eps = {m.eps for m in bns}
        assert len(eps) == 1, "mixed BatchNorm eps"
        gamma = torch.cat([m.weight for m in bns])
        beta = torch.cat([m.bias for m in bns])
        with torch.no_grad():
            mean = torch.cat([m.running_mean for m in bns])
            inv = torch.rsqrt(torch.cat([m.running_var for m in bns]) + eps.pop())
        s = gamma * inv
        t = beta - mean * s
        sizes = [m.num_features for m in bns]
        soffs = [sum(sizes[:i]) for i in range(len(sizes))]
        if any(conv.bias is not None for conv, _ in pairs):
            # folded biases b * s + t of ALL pairs as one vector expression (a cached zero block stands in where a convolution has no
            # bias): two launches and one split whose outputs are all used — per pair it was a mul + an add forward, two muls
            # backward, and SplitWithSizesBackward materialised a zero gradient for every scale slice that no bias consumed (~110
            # launches per joint-mode step, tools/joint_aten_sources.py)
            bvec = torch.cat([conv.bias if conv.bias is not None else _zeros_const((bn.num_features,), s.device) for conv, bn in pairs])
            t = bvec * s + t
        ts = torch.split(t, sizes)
        wfs = R.fold_all(s, soffs, [conv.weight for conv, _ in pairs])
        for (conv, bn), wf, tv in zip(pairs, wfs, ts):
            self._slots[id(conv)] = (wf, tv)

    def unit(self, unit: nn.Module) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """folded (W (out, in), b (out)) of a pytorch_utils Conv1d / Conv2d unit (conv [+ bn.bn]), differentiable"""
        return self.conv(unit.conv, unit.bn.bn if hasattr(unit, "bn") else None)

    def conv(self, conv: nn.Module, bn: Optional[nn.Module]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """(W (out, in * kernel), b) of a 1x1 convolution / Linear [+ BatchNorm]"""
        if bn is None:
            return conv.weight.reshape(conv.weight.shape[0], -1), conv.bias
        wf, b = self._slots[id(conv)]
        return wf.reshape(wf.shape[0], -1), b

    def conv4d(self, conv: nn.Module) -> Tuple[torch.Tensor, torch.Tensor]:
        """(folded weight in the convolution's own 4-D shape and memory format, bias) of a BatchNorm-followed Conv2d"""
        return self._slots[id(conv)]


_zero_blocks = {}      # (shape, device) -> a constant block of zeros (never written, never requires grad)


def _zeros_const(shape, device) -> torch.Tensor:
    key = (tuple(int(v) for v in shape), str(device))
    z = _zero_blocks.get(key)
    if z is None:
        z = _zero_blocks[key] = torch.zeros(key[0], dtype=torch.float32, device=device)
    return z


def _pad_dim(x: torch.Tensor, dim: int, pad: int) -> torch.Tensor:
    """x with `pad` zeros appended along `dim` — as a concatenation with a cached zero block: ONE launch forward and NONE backward
    (cat's backward hands out views), where F.pad is fill + copy forward and a clone backward (35 launches per joint-mode step,
    tools/joint_aten_sources.py)"""
    if pad == 0:
        return x
    shape = list(x.shape)
    shape[dim] = pad
    return torch.cat([x, _zeros_const(shape, x.device)], dim=dim)


def _pad_rows(W: torch.Tensor, b: Optional[torch.Tensor], mult: int = 4):
    """zero rows appended so that the layer's output width is a multiple of `mult` (the 1-wide objectness and the 46-wide RCNN
    regression layers: the row kernels work on multiples of four channels)"""
    n = W.shape[0]
    pad = (-n) % mult
    if pad == 0:
        return W, b
    return _pad_dim(W, 0, pad), (_pad_dim(b, 0, pad) if b is not None else None)


def _head_rows(fold: BnFold, head: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """a Conv1d head (rpn.py:34-58, rcnn.py:43-89) on rows: (M, C) -> (M, pad4(out)); Dropout between the units when active"""
    units, cur_layers, cur_acts = list(head), [], []
    out = x
    for m in units:
        if isinstance(m, nn.Dropout):
            if m.training and m.p > 0:
                if cur_layers:
                    out = R.rows_mlp(out, cur_layers, cur_acts)
                    cur_layers, cur_acts = [], []
                out = F.dropout(out, m.p, True)
            continue
        W, b = fold.unit(m)
        cur_layers.append(_pad_rows(W, b))
        cur_acts.append(1 if getattr(m, "activation", None) is not None else 0)
    if cur_layers:
        out = R.rows_mlp(out, cur_layers, cur_acts)
    return out


def _attention_rows(fold: BnFold, mod, point: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
    """AttentionFusion.forward (backbone.py:35-81) on rows: point (M, pc), img (M, ic) -> (M, oc)"""
    ia = mod.IA_Layer
    rc = ia.fc1.weight.shape[0]
    # gate = sigmoid(fc3(tanh(fc1(img) + fc2(point)))): fc1 / fc2 as ONE two-operand layer
    w12 = torch.cat([ia.fc1.weight, ia.fc2.weight], dim=1)
    t = R.rows_mlp(img, [_pad_rows(w12, ia.fc1.bias + ia.fc2.bias)], [2], x2=point)            # (M, pad4(rc)); padded columns tanh(0) = 0
    w3 = ia.fc3.weight
    if t.shape[1] != rc:
        w3 = _pad_dim(w3, 1, t.shape[1] - rc)
    z = R.rows_mlp(t, [_pad_rows(w3, ia.fc3.bias)], [0])                                        # (M, 4): column 0 is the logit
    gate = torch.sigmoid(z[:, :1])
    Wi, bi = fold.conv(ia.conv1[0], ia.conv1[1])
    j = R.rows_mlp(img, [(Wi, bi)], [1]) * gate
    Wf, bf = fold.conv(mod.conv1, mod.bn1)
    return R.rows_mlp(point, [(Wf, bf)], [1], x2=j)


def _sa_level_rows(fold: BnFold, sa, xyz: torch.Tensor, feats: Optional[torch.Tensor], new_xyz: torch.Tensor, grid=None,
                   canon: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one PointnetSAModuleMSG level on rows: xyz (S, n, 3), feats (S n, C) or None, new_xyz (S, m, 3) -> (S m, sum C_k)"""
    S, n, _ = xyz.shape
    flat_xyz, flat_ctr = xyz.reshape(-1, 3), new_xyz.reshape(-1, 3).contiguous()
    groupers = list(sa.groupers)
    if len(groupers) == 2:
        g0, g1 = groupers
        neigh = pointnet2_utils.ball_query_dual(g0.radius, g0.nsample, g1.radius, g1.nsample, xyz, new_xyz, grid=grid)
    else:
        neigh = [pointnet2_utils.ball_query(g.radius, g.nsample, xyz, new_xyz) for g in groupers]
    plans = [R.RowsPlan(nb, n, canon) for nb in neigh]
    return R.sa_level_rows(feats, flat_xyz, flat_ctr, plans, [[fold.unit(u) for u in mlp] for mlp in sa.mlps])


def _channel_sums(g: torch.Tensor) -> torch.Tensor:
    """(B, C, H, W) -> (C) sums over batch and pixels.  On channels-last maps = the column sums of the (B H W, C) row matrix on the
    two-pass kernels of csrc/jm_rows.h: torch's reduction of many inputs to few outputs zeroes a semaphore buffer with a memset,
    which is not reliably ordered inside a replayed HIP graph on this stack (tools/quarantine/graphed.py)"""
    if (g.is_cuda and torch.cuda.is_current_stream_capturing() and g.dtype == torch.float32 and g.shape[1] % 4 == 0
            and g.is_contiguous(memory_format=torch.channels_last)):
        return R.colsum(g.permute(0, 2, 3, 1).reshape(-1, g.shape[1]))
    return g.sum(dim=(0, 2, 3))


class _Conv3x3BiasRelu(torch.autograd.Function):
    """relu(conv3x3(x, w, padding 1) + b) of an image block's first layer (backbone.py:16-32 conv1 + bn1 + relu, BatchNorm folded
    into w / b by the caller).  Forward = the inference engine's one-pass kernels (csrc/conv_rgb.hip for the 3-channel first layer,
    the fused Winograd F(2x2, 3x3) of csrc/conv_wino.hip for the others: no bias / ReLU passes over the 0.5 GB activations);
    backward: d(x) of the Winograd layers = the same kernel on the flipped, transposed weight (a stride-1 3x3 convolution's data
    gradient IS one), d(w) on MIOpen (aten.convolution_backward), d(b) = a channel sum."""

    @staticmethod
    def forward(ctx, x, w, b):
        from .ops.fusion import conv3x3_rgb_bias_relu, conv3x3_wino_bias_relu, pack_rgb_weight, pack_wino_weight, wino_supported
        cout, cin = w.shape[0], w.shape[1]
        if cin == 3:
            y = conv3x3_rgb_bias_relu(x, w, b, pack_rgb_weight(w))
            ctx.kind = "rgb"
        elif wino_supported(cin, cout) and x.is_contiguous(memory_format=torch.channels_last):
            y = conv3x3_wino_bias_relu(x, pack_wino_weight(w), b, cout)
            ctx.kind = "wino"
        else:
            y = torch.relu_(F.conv2d(x, w, b, stride=1, padding=1))
            ctx.kind = "miopen"
        ctx.save_for_backward(x, w, y)
        return y

    @staticmethod
    def backward(ctx, dy):
        from .ops.fusion import conv3x3_wino_bias_relu, pack_wino_weight, wino_supported
        x, w, y = ctx.saved_tensors
        cout, cin = w.shape[0], w.shape[1]
        dpre = torch.ops.aten.threshold_backward(dy, y, 0)
        need_x = ctx.needs_input_grad[0]
        dx = None
        wino_dx = need_x and ctx.kind == "wino" and wino_supported(cout, cin) and dpre.is_contiguous(memory_format=torch.channels_last)
        if wino_dx:
            dx = conv3x3_wino_bias_relu(dpre, pack_wino_weight(w.flip(2, 3).transpose(0, 1)), None, cin, relu=False)
        # d(b) = the channel sums of d(pre), as a plain reduction: the library's own bias gradient comes out as ZEROS when the
        # call is replayed from a HIP graph (tools/quarantine/train_graphs.py; tools/_dbg notes in tools/quarantine/graphed.py) — and costs a launch either way
        gx, dw, _ = torch.ops.aten.convolution_backward(dpre, x, w, [cout], [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                       [need_x and not wino_dx, True, False])
        return (dx if wino_dx else gx), dw, _channel_sums(dpre)


class _BiasReluInplace(torch.autograd.Function):
    """relu(x + b[c]) written over x (a convolution's bias-free output, which nothing else reads), d(b) = channel sums of the
    masked gradient: one elementwise pass forward (csrc/elementwise.hip on channels-last maps), no bias inside the convolution —
    see _Conv3x3BiasRelu.backward for why the convolution library's bias gradient is avoided"""

    @staticmethod
    def forward(ctx, x, b):
        from .ops.fusion import bias_relu_
        y = bias_relu_(x, b)
        ctx.mark_dirty(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dpre = torch.ops.aten.threshold_backward(dy, y, 0)
        return dpre, _channel_sums(dpre)


def _image_pyramid(fold: BnFold, net, image: torch.Tensor) -> List[torch.Tensor]:
    """the four BasicBlocks (backbone.py:16-32; conv3x3 + BN + ReLU + conv3x3 / 2), BatchNorm folded, channels-last"""
    x = image
    maps = []
    for blk in net.Img_Block:
        w1, t = fold.conv4d(blk.conv1)         # (channels-last when the parameter is: train_joint.prepare_rows converts them once)
        y = _Conv3x3BiasRelu.apply(x, w1, t)
        x = F.conv2d(y, blk.conv2.weight, blk.conv2.bias, stride=blk.conv2.stride, padding=blk.conv2.padding)
        maps.append(x)
    return maps


def _image_fusion_map(fold: BnFold, net, maps: List[torch.Tensor]) -> torch.Tensor:
    """relu(bn(conv1x1(cat_i deconv_i(img_i)))) (backbone.py:187-193), BatchNorm folded.  The training path keeps the reference's
    un-composed form: four kernel == stride transposed convolutions, their 64-channel concatenation, one 1x1 convolution —
    204 GFLOP per 4 frames forward + backward against 363 for the composed form the inference engine gathers from
    (tools/joint_image_probe.py: 5.1 against 6.1 ms)"""
    Wf, bf = fold.conv4d(net.image_fusion_conv)
    # the deconvolutions' biases ride through the (linear) 1x1 convolution: Wf (de + b_d) + bf = Wf de + (Wf b_d + bf) — a 64-vector
    # product instead of a pass over the concatenated map, and no bias inside any library convolution
    biases = [dc.bias if dc.bias is not None else m.new_zeros(dc.out_channels) for dc, m in zip(net.DeConv, maps)]
    Wf2 = Wf.reshape(Wf.shape[0], -1)
    b_eff = bf + Wf2 @ torch.cat(biases)
    de = torch.cat([F.conv_transpose2d(m, dc.weight, None, stride=dc.stride, padding=dc.padding, output_padding=dc.output_padding)
                    for dc, m in zip(net.DeConv, maps)], dim=1)
    if de.is_cuda and de.is_contiguous(memory_format=torch.channels_last) and de.shape[1] % 4 == 0 and Wf2.shape[0] % 4 == 0:
        # the 1 x 1 fusion convolution + bias + ReLU IS a dense layer on the (B H W, C) rows of the channels-last map: one bounds-checked
        # GEMM launch per direction (csrc/rows_gemm.hip) instead of a library convolution, a bias / ReLU pass and a channel sum
        B, ctot, H, W = de.shape
        y = R.rows_mlp(de.permute(0, 2, 3, 1).reshape(B * H * W, ctot), [(Wf2, b_eff)], [1])  # (B H W, q)
        return y.view(B, H, W, Wf2.shape[0]).permute(0, 3, 1, 2)                              # = channels-last (B, q, H, W)
    return _BiasReluInplace.apply(F.conv2d(de, Wf, None), b_eff)


def backbone_forward_rows(engine, xyz: torch.Tensor, image: torch.Tensor, pts_xy: torch.Tensor, fold: BnFold,
                          pyr: Optional[FpsPyramid] = None) -> torch.Tensor:
    """PointNet2MSG.forward (backbone.py:159-196) on rows -> point features (B N, C)"""
    net, cfg = engine.rpn.backbone_net, engine.cfg
    B, N, _ = xyz.shape
    own = pyr is None
    if own:
        pyr = FpsPyramid(xyz, list(cfg.sa_npoints), overlap=engine.overlap, with_interp=True, grid_radii=engine._grid_radii())
    # stream I: the image pyramid and the fused image map (MIOpen + the one-pass first layers), forward here and — autograd runs a
    # node's backward on the stream of its forward — BACKWARD there too: the convolutions' large kernels run next to the point
    # branch's many small launches.  ORDER of issue matters twice.  Forward: block i is issued one level ahead of its consumer (the
    # LI-Fusion gather of level i), so the convolutions run under set abstraction i.  Backward: the autograd engine hands out READY
    # nodes latest-created first, so a branch created at the very start of the forward (round-5 first form: the whole pyramid and
    # the fusion map up front) is launched only after every point-branch node — 10 ms of convolution backward alone at the end of
    # the step (tools/joint_timeline.sh).  Created where they are consumed, block i's backward is handed out right after level i's
    # set abstraction backward and runs under levels i-1 .. 1; the fusion map's right after the final attention block's.
    main = torch.cuda.current_stream(xyz.device)
    img_stream = side_stream(xyz.device, 1) if engine.overlap else main
    maps, map_events = [], []

    def image_block(i):
        img_stream.wait_stream(main) if i == 0 else None
        with torch.cuda.stream(img_stream):
            blk = net.Img_Block[i]
            w1, t = fold.conv4d(blk.conv1)         # (channels-last when the parameter is: train_joint.prepare_rows converts them once)
            if img_stream is not main:             # made on the main stream, read here — and by this stream's backward nodes after
                w1.record_stream(img_stream)       # the fold object is gone: the allocator must not recycle them under those reads
                t.record_stream(img_stream)        # (w1 = a view of the fold's ONE slab: one block, one mark)
            y = _Conv3x3BiasRelu.apply(image if i == 0 else maps[i - 1], w1, t)
            m = F.conv2d(y, blk.conv2.weight, blk.conv2.bias, stride=blk.conv2.stride, padding=blk.conv2.padding)
            ev = torch.cuda.Event()
            ev.record(img_stream)
        if img_stream is not main:
            m.record_stream(main)
        maps.append(m)
        map_events.append(ev)

    if img_stream is not main:
        image.record_stream(img_stream)
    l_xyz, l_feats, l_xy = [xyz], [None], [pts_xy]
    for i, sa in enumerate(net.SA_modules):
        prof.region(f"image_block_{i + 1}(MIOpen)", lambda k=i: image_block(k))
        idx, new_xyz = pyr.level(i)
        with prof.scope(f"rpn_sa{i + 1}"):
            feats = _sa_level_rows(fold, sa, l_xyz[i], l_feats[i], new_xyz, grid=pyr.grid(i))
        xy_i = (pointnet2_utils.gather_point_rows(l_xy[i], idx) if idx.is_cuda else
                    torch.gather(l_xy[i], 1, idx.long().unsqueeze(-1).expand(-1, -1, 2)))
        main.wait_event(map_events[i])
        with prof.scope(f"li_fusion{i + 1}"):
            feats = _attention_rows(fold, net.Fusion_Conv[i], feats, R.feature_gather_rows(maps[i], xy_i))
        l_xyz.append(new_xyz); l_feats.append(feats); l_xy.append(xy_i)
    # the fused image map: issued here (under the feature-propagation modules), consumed by the final attention block
    if img_stream is not main:
        for t_ in fold.conv4d(net.image_fusion_conv):
            t_.record_stream(img_stream)
    with torch.cuda.stream(img_stream):
        fused_img = prof.region("image_deconv+fusion_conv(MIOpen)", lambda: _image_fusion_map(fold, net, maps))
        fused_ev = torch.cuda.Event()
        fused_ev.record(img_stream)
    if img_stream is not main:
        fused_img.record_stream(main)
    nfp = len(net.FP_modules)
    for i in range(-1, -(nfp + 1), -1):
        with prof.scope(f"fp{nfp + 1 + i}"):
            nn3, w = pyr.interp(nfp + i)
            carried = R.three_interpolate_rows(l_feats[i], nn3, w)
            mlp = net.FP_modules[i].mlp
            layers = [fold.unit(u) for u in mlp]
            l_feats[i - 1] = R.rows_mlp(carried, layers, [1] * len(layers), x2=l_feats[i - 1])
    main.wait_event(fused_ev)
    with prof.scope("li_fusion_final"):
        out = _attention_rows(fold, net.final_fusion_img_point, l_feats[0], R.feature_gather_rows(fused_img, pts_xy))
    if own:
        pyr.release()
    return out


# RoI sets are full of exact copies (cyclic roipool padding): centres picked from copies of one point are copies of one another
# (same coordinates, same neighbour list, hence bit-identical features), so the NEXT level's rows are planned on the first centre of
# every such class — the training-path form of the inference engine's representative centres (csrc/sa_dedupe.hip)
CANON_CENTRES = True      # (a module constant, not an environment switch: tests/test_gpu_rows.py compares both settings)


def _centre_canon(canon: torch.Tensor, pick: torch.Tensor) -> torch.Tensor:
    """canon (R, n) int32 canonical point of every point of a set, pick (R, m) the points chosen as centres -> (R, m) int32: the
    first centre whose point has the same canonical point.  The gradient of a copy's feature row then lands on its representative's
    row — the same parameters and the same pooled points receive it (max-pool hands the gradient to ONE of the tied copies in the
    reference as well, pointnet2_modules.py:58-61)"""
    cp = torch.gather(canon, 1, pick.long())
    m = cp.shape[1]
    slot = torch.arange(m, dtype=torch.int32, device=cp.device)
    same = cp.unsqueeze(2) == cp.unsqueeze(1)                                # (R, m, m): [r, i, j] centre j is a copy of centre i
    return torch.where(same, slot.view(1, 1, m), slot.new_full((), m)).amin(dim=2)


def rcnn_forward_rows(engine, pts_input: torch.Tensor, fold: BnFold, count: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
    """RCNN.forward (rcnn.py:176-202) on pooled RoI points (R, S, 5 + C): rcnn_cls (R, 1), rcnn_reg (R, 46), rcnn_feat (R, 512).
    count (R,) int32: distinct points per RoI (rows count .. S - 1 are cyclic copies, roipool3d_kernel.cu:123-160) — every level's
    rows are then planned on canonical entries only: the first level's on the canonical points, the later ones on the first of
    the centres that were picked from copies of one point (_centre_canon)"""
    net = engine.rcnn_net
    Rn, S, Cin = pts_input.shape
    k = net.rcnn_input_channel
    rows = pts_input.reshape(Rn * S, Cin)
    xyz = pts_input[:, :, 0:3].contiguous()
    x5 = _pad_dim(rows[:, :k], 1, (-k) % 4)                                     # (R S, 8): K = 5 padded to a multiple of 4
    rpn_feat = rows[:, k:].contiguous()
    up = [fold.unit(u) for u in net.xyz_up_layer]
    W0, b0 = up[0]
    up[0] = (_pad_dim(W0, 1, x5.shape[1] - W0.shape[1]), b0)
    xyz_feat = R.rows_mlp(x5, up, [1] * len(up))
    Wm, bm = fold.unit(net.merge_down_layer[0])
    feats = R.rows_mlp(xyz_feat, [(Wm, bm)], [1], x2=rpn_feat)              # (R S, C): merge_down on [xyz_feature | rpn_feature]
    l_xyz = xyz
    canon = None
    if count is not None:
        from .ops.pointnet2 import fused
        canon = fused.canon_from_count(count, S)
    for li, sa in enumerate(net.SA_modules):
        with prof.scope(f"rcnn_sa{li + 1}"):
            n = l_xyz.shape[1]
            if sa.npoint is not None:
                pick, new_xyz = pointnet2_utils.farthest_point_sample_xyz(l_xyz, sa.npoint)
                feats = _sa_level_rows(fold, sa, l_xyz, feats, new_xyz, canon=canon)
                l_xyz = new_xyz
                if canon is not None and CANON_CENTRES:
                    canon = _centre_canon(canon, pick)
                else:
                    canon = None
            else:       # GroupAll: one group of all n points per RoI, coordinates not re-centred (pointnet2_utils.py:273-290)
                idx = torch.arange(n, dtype=torch.int32, device=xyz.device).expand(Rn, 1, n).contiguous()
                plan = R.RowsPlan(idx, n, canon)
                feats = R.sa_scale_rows(feats, l_xyz.reshape(-1, 3), None, plan, [fold.unit(u) for u in sa.mlps[0]])
                l_xyz = None
    ncls = net.cls_layer[-1].conv.out_channels
    nreg = net.reg_layer[-1].conv.out_channels
    return dict(rcnn_cls=_head_rows(fold, net.cls_layer, feats)[:, :ncls], rcnn_reg=_head_rows(fold, net.reg_layer, feats)[:, :nreg],
                rcnn_feat=feats)


def rpn_forward_rows(engine, xyz, image, pts_xy, fold: BnFold, pyr: Optional[FpsPyramid] = None) -> Dict[str, torch.Tensor]:
    """backbone + RPN heads on rows (rpn.py:71-87): rpn_cls (B, N, 1), rpn_reg (B, N, C), backbone_features (B, C, N) view, and the
    (B N, C) feature rows themselves"""
    rpn = engine.rpn
    B, N, _ = xyz.shape
    feats = backbone_forward_rows(engine, xyz, image, pts_xy, fold, pyr)                 # (B N, C)
    ncls = rpn.rpn_cls_layer[-1].conv.out_channels
    nreg = rpn.rpn_reg_layer[-1].conv.out_channels
    rpn_cls = _head_rows(fold, rpn.rpn_cls_layer, feats)[:, :ncls].reshape(B, N, ncls)
    rpn_reg = _head_rows(fold, rpn.rpn_reg_layer, feats)[:, :nreg].reshape(B, N, nreg)
    C = feats.shape[1]
    return dict(rpn_cls=rpn_cls, rpn_reg=rpn_reg, backbone_features=feats.view(B, N, C).transpose(1, 2), feature_rows=feats)


@torch.no_grad()
def pooled_rois(engine, xyz, rpn_out: Dict[str, torch.Tensor], rois_per_frame: int):
    """ProposalLayer + roipool3d WITHOUT gradient (the reference's are not differentiable either): rois (B, K, 7), pooled points
    (B K, S, 5 + C), distinct points per RoI (B K) — the first `rois_per_frame` proposals of every frame stand in for
    ProposalTargetLayer's sampled RoIs (config.py:153)"""
    cfg = engine.cfg
    B, N, _ = xyz.shape
    feats = rpn_out["feature_rows"].detach()
    C = feats.shape[1]
    det = dict(rpn_cls=rpn_out["rpn_cls"].detach().contiguous(), rpn_reg=rpn_out["rpn_reg"].detach().contiguous(), backbone_xyz=xyz)
    rois, _ = engine.proposals(det)
    rois = rois[:, :rois_per_frame].contiguous()
    pf = torch.empty((B, N, 2 + C), dtype=torch.float32, device=xyz.device)           # point_rcnn.py:42-44, proposal_target_layer.py:26
    pf[:, :, 0] = (torch.sigmoid(det["rpn_cls"][:, :, 0]) > cfg.rpn_score_thresh).float()
    pf[:, :, 1] = torch.norm(xyz, p=2, dim=2) / 70.0 - 0.5
    pf[:, :, 2:] = feats.view(B, N, C)                                                # rows ARE the (B, N, C) layout roipool reads
    pooled, _, count = roipool3d_canonical_gpu(xyz, pf, rois, cfg.pool_extra_width, cfg.rcnn_num_points, return_count=True)
    return rois, pooled.view(-1, cfg.rcnn_num_points, pooled.shape[-1]), count.view(-1)


def rcnn_branch_rows(engine, pts_input, count, fold: Optional[BnFold] = None, ready: Optional[torch.cuda.Event] = None) -> Dict[str, torch.Tensor]:
    """stream R: the RCNN (rcnn.py:158-202).  Its BACKWARD then runs there too (autograd: a node's backward on its forward's
    stream), next to the backbone's backward on the main stream — neither feeds the other: roipool3d has no gradient.  The
    caller's stream waits for the outputs (main.wait_stream) before it reads them.
    fold: None = the branch folds the RCNN's own BatchNorms (config.py:107 ships none) on ITS stream, into a graph of its own —
    the RPN's and the RCNN's halves of the loss are back-propagated separately (train_joint._rows_forward_backward), and a fold node
    serves one backward."""
    dev = pts_input.device
    main = torch.cuda.current_stream(dev)
    side = side_stream(dev, 3) if engine.overlap else main
    # `ready`: an event recorded on the main stream behind the RoI pooling — the branch must not wait for what the caller has
    # queued on the main stream SINCE (the backbone's whole backward)
    if ready is not None and side is not main:
        side.wait_event(ready)
    else:
        side.wait_stream(main)
    with torch.cuda.stream(side):
        out = rcnn_forward_rows(engine, pts_input, BnFold(engine.rcnn_net) if fold is None else fold, count)
    if side is not main:
        pts_input.record_stream(side)
        count.record_stream(side)
    out["_stream"] = side
    return out


def joint_forward_rows(engine, xyz, image, pts_xy, rois_per_frame: int = 64, pyr: Optional[FpsPyramid] = None) -> Dict[str, torch.Tensor]:
    """the detector in TRAIN composition (point_rcnn.py:24-70) on rows; same outputs as train_joint.joint_forward.
    Its backward runs on three streams (main, image, RCNN), each node on the stream of its forward; gradients cross streams only
    through the autograd engine's own hand-over.  Safe next to other live autograd graphs over the same parameters (an earlier
    step's outputs, DDP's stashed AccumulateGrad nodes: tests/test_gpu_rows.py holds two of them across ten asynchronous steps)."""
    fold = BnFold(engine.rpn)
    out = rpn_forward_rows(engine, xyz, image, pts_xy, fold, pyr)
    rois, pts_input, count = pooled_rois(engine, xyz, out, rois_per_frame)
    rc = rcnn_branch_rows(engine, pts_input, count)
    side = rc.pop("_stream")
    main = torch.cuda.current_stream(xyz.device)
    if side is not main:
        main.wait_stream(side)
        for t in rc.values():
            t.record_stream(main)
    out.update(rc, rois=rois)
    out.pop("feature_rows")
    return out
