"""The joint-mode training step as HIP-graph SECTIONS (route 'graphs'): the rows route of train_rows.py — same kernels, same
network (point_rcnn.py:24-70 in TRAIN mode: backbone.py:159-196, rpn.py:71-87, rcnn.py:158-202), same gradients — with every
single-stream piece of it captured once and replayed with one launch per direction (graphed.GraphedSection).

Sections and their streams (F: geometry, I: image, M: main, R: RCNN):

    F  geometry      coordinates only, no gradient: the FPS chain, the four levels' neighbour-search grids and ball queries, the
                     distinct-row plans of all eight scales, the feature-propagation 3-NN + weights, the levels' pixel coordinates —
                     everything of pointnet2_modules.py:35-47 / :147-150 that never sees a feature.  Started a step AHEAD for the
                     announced cloud (next_xyz); `geometry_take` copies its outputs into the step's own buffers.
       (every section below folds the eval-mode BatchNorms of ITS modules into their convolutions inside the graph —
        train_rows.BnFold on the section's pairs — and its backward graph hands d(loss)/d(W, gamma, beta) to the parameters)
    I  image_1..4    BasicBlock i (backbone.py:16-32), fused_map: the deconvolution pyramid + fusion convolution (:187-193)
    M  level_1..4    set abstraction (rows) + LI-Fusion gather + attention block of level i
    M  fp            the four feature-propagation modules
    M  final         final attention fusion + both RPN heads
    M  pooled        proposal layer + roipool3d (no gradient, as in the reference)
    R  rcnn          rcnn.py:176-202 on the pooled points

The host enqueues ~30 graph launches, a dozen event waits, the thin loss, ONE backward, the all-reduce and Adam instead of ~1050
kernel launches through ~60 autograd Functions.  Requires frozen BatchNorm statistics (train_joint.freeze_bn), like the rows route.
"""
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import train_rows as TR
from .graphed import GraphedSection
from .ops import rows as R
from .ops.pointnet2 import pointnet2_utils
from .ops.pointnet2.pyramid import side_stream


def _split_params(all_pairs, mods: Sequence[nn.Module]):
    """(BatchNorm pairs, trainable parameters) of a section's modules: a pair belongs to the section that holds its convolution.
    The section folds its own pairs (train_rows.BnFold on the pair list, inside the captured function) and receives EVERY
    parameter as an argument: the backward graph then produces the gradients of W, gamma, beta themselves"""
    inside = {id(x) for m in mods for x in m.modules()}
    pairs = [(conv, bn) for conv, bn in all_pairs if id(conv) in inside]
    seen, params = set(), []
    for m in mods:
        for p in m.parameters():
            if id(p) not in seen and p.requires_grad:
                seen.add(id(p))
                params.append(p)
    return pairs, params


class JointGraphs:
    """the sections of one engine at one batch shape; built lazily by `joint_graphs(engine)`"""

    def __init__(self, engine, graphed: Optional[str] = None):
        """graphed: comma-separated section groups to CAPTURE — geom, image, level, fp, final, pooled, rcnn — the others run
        eagerly under plain autograd in the same composition (default: the JM_JOINT_GRAPHED environment switch, else all)"""
        import os
        if graphed is None:
            graphed = os.environ.get("JM_JOINT_GRAPHED", "geom,image,level,fp,final,pooled,rcnn")
        on = {g.strip() for g in graphed.split(",") if g.strip()}
        self.graphed = on
        self.engine = engine
        net, rcnn = engine.rpn.backbone_net, engine.rcnn_net
        self.all_pairs = TR.bn_pairs(engine)
        self.nlev = len(net.SA_modules)
        self.sec_geom = GraphedSection(self._geom_fn, "geometry", enabled="geom" in on)
        self.sec_take = GraphedSection(lambda *ts: tuple(t.clone() for t in ts), "geometry_take", enabled="geom" in on)
        self.sec_img = [GraphedSection(lambda *ts, k=i: self._img_fn(k, *ts), f"image_{i + 1}", enabled="image" in on) for i in range(self.nlev)]
        self.sec_fmap = GraphedSection(self._fmap_fn, "fused_map", enabled="image" in on)
        self.sec_level = [GraphedSection(lambda *ts, k=i: self._level_fn(k, *ts), f"level_{i + 1}", enabled="level" in on) for i in range(self.nlev)]
        self.sec_fp = GraphedSection(self._fp_fn, "fp", enabled="fp" in on)
        self.sec_final = GraphedSection(self._final_fn, "final", enabled="final" in on)
        self.sec_pooled = GraphedSection(self._pooled_fn, "pooled", enabled="pooled" in on)
        self.sec_rcnn = GraphedSection(self._rcnn_fn, "rcnn", enabled="rcnn" in on)
        # which modules every section touches -> (pairs, leaf parameters)
        self.p_img = [_split_params(self.all_pairs, [net.Img_Block[i]]) for i in range(self.nlev)]
        self.p_fmap = _split_params(self.all_pairs, list(net.DeConv) + [net.image_fusion_conv, net.image_fusion_bn])
        self.p_level = [_split_params(self.all_pairs, [net.SA_modules[i], net.Fusion_Conv[i]]) for i in range(self.nlev)]
        self.p_fp = _split_params(self.all_pairs, list(net.FP_modules))
        self.p_final = _split_params(self.all_pairs, [net.final_fusion_img_point, engine.rpn.rpn_cls_layer, engine.rpn.rpn_reg_layer])
        self.p_rcnn = _split_params(self.all_pairs, [rcnn.xyz_up_layer, rcnn.merge_down_layer, rcnn.SA_modules, rcnn.cls_layer, rcnn.reg_layer])
        self.rois_per_frame = 64
        self._announced = None           # (cloud tensor, pts_xy tensor, event on F) of the geometry in flight
        self._geom_out = None
        self._meta = {}                  # cloud shape -> [[(groups, nsample) per scale] per level]
        self._shape = None

    # ------------------------------------------------------------------------------------------------------------------ geometry
    def _geom_fn(self, xyz, pts_xy):
        eng = self.engine
        net, cfg = eng.rpn.backbone_net, eng.cfg
        radii = eng._grid_radii()
        out, meta = [], []
        cur, cur_xy = xyz, pts_xy
        levels = []
        for i, sa in enumerate(net.SA_modules):
            grid = None
            if radii[i]:
                g = pointnet2_utils.BallQueryGrid(cur, radii[i])
                grid = g if g.ws is not None else None
            idx, new_xyz = pointnet2_utils.farthest_point_sample_xyz(cur, cfg.sa_npoints[i])
            xy_i = pointnet2_utils.gather_point_rows(cur_xy, idx)
            groupers = list(sa.groupers)
            if len(groupers) == 2:
                g0, g1 = groupers
                neigh = pointnet2_utils.ball_query_dual(g0.radius, g0.nsample, g1.radius, g1.nsample, cur, new_xyz, grid=grid)
            else:
                neigh = [pointnet2_utils.ball_query(g.radius, g.nsample, cur, new_xyz) for g in groupers]
            n = cur.shape[1]
            plans = [R.RowsPlan(nb, n) for nb in neigh]
            out += [new_xyz, xy_i]
            pm = []
            for p in plans:
                out += [p.d, p.offsets, p.row_point, p.row_group]
                pm.append((p.groups, p.ns))
            meta.append(pm)
            levels.append(new_xyz)
            cur, cur_xy = new_xyz, xy_i
        for k in range(len(levels)):
            unknown = xyz if k == 0 else levels[k - 1]
            nn3, w = pointnet2_utils.three_nn_weights(unknown, levels[k])
            out += [nn3, w]
        self._meta[tuple(xyz.shape)] = meta
        return tuple(out)

    @property
    def geom_meta(self):
        return self._meta[self._shape]

    def _geom_unpack(self, ts: Sequence[torch.Tensor]):
        """[(new_xyz, xy, [RowsPlan per scale])] per level, [(nn3, w)] per level"""
        k, levels = 0, []
        for pm in self.geom_meta:
            new_xyz, xy = ts[k], ts[k + 1]
            k += 2
            plans = []
            for groups, ns in pm:
                plans.append(R.RowsPlan.from_tensors(ts[k], ts[k + 1], ts[k + 2], ts[k + 3], groups, ns))
                k += 4
            levels.append((new_xyz, xy, plans))
        interp = []
        for _ in self.geom_meta:
            interp.append((ts[k], ts[k + 1]))
            k += 2
        return levels, interp

    # --------------------------------------------------------------------------------------------------------------------- image
    def _img_fn(self, i, x, *ws):
        net = self.engine.rpn.backbone_net
        fold = TR.BnFold(None, self.p_img[i][0])
        blk = net.Img_Block[i]
        w1, t = fold.conv4d(blk.conv1)
        y = TR._Conv3x3BiasRelu.apply(x, w1, t)
        return F.conv2d(y, blk.conv2.weight, blk.conv2.bias, stride=blk.conv2.stride, padding=blk.conv2.padding)

    def _fmap_fn(self, *ts):
        net = self.engine.rpn.backbone_net
        maps = list(ts[:self.nlev])
        fold = TR.BnFold(None, self.p_fmap[0])
        return TR._image_fusion_map(fold, net, maps)

    # --------------------------------------------------------------------------------------------------------------- point branch
    def _level_fn(self, i, *ts):
        """ts = xyz_i, [feats_{i-1}], new_xyz, xy_i, 4 tensors per scale plan, map_i, folded..., leafs..."""
        net = self.engine.rpn.backbone_net
        sa = net.SA_modules[i]
        nsc = len(sa.groupers)
        k = 0
        xyz = ts[k]; k += 1
        feats = None
        if i > 0:
            feats = ts[k]; k += 1
        new_xyz, xy_i = ts[k], ts[k + 1]
        k += 2
        plans = []
        for sc in range(nsc):
            groups, ns = self.geom_meta[i][sc]
            plans.append(R.RowsPlan.from_tensors(ts[k], ts[k + 1], ts[k + 2], ts[k + 3], groups, ns))
            k += 4
        fmap = ts[k]; k += 1
        fold = TR.BnFold(None, self.p_level[i][0])
        f = R.sa_level_rows(feats, xyz.reshape(-1, 3), new_xyz.reshape(-1, 3).contiguous(), plans,
                            [[fold.unit(u) for u in mlp] for mlp in sa.mlps])
        return TR._attention_rows(fold, net.Fusion_Conv[i], f, R.feature_gather_rows(fmap, xy_i))

    def _fp_fn(self, *ts):
        net = self.engine.rpn.backbone_net
        n = self.nlev
        l_feats = [None] + list(ts[:n])
        interp = [(ts[n + 2 * k], ts[n + 2 * k + 1]) for k in range(n)]
        fold = TR.BnFold(None, self.p_fp[0])
        nfp = len(net.FP_modules)
        for i in range(-1, -(nfp + 1), -1):
            nn3, w = interp[nfp + i]
            carried = R.three_interpolate_rows(l_feats[i], nn3, w)
            layers = [fold.unit(u) for u in net.FP_modules[i].mlp]
            l_feats[i - 1] = R.rows_mlp(carried, layers, [1] * len(layers), x2=l_feats[i - 1])
        return l_feats[0]

    def _final_fn(self, feats0, fused_img, pts_xy, *ws):
        eng = self.engine
        rpn, net = eng.rpn, eng.rpn.backbone_net
        fold = TR.BnFold(None, self.p_final[0])
        feats = TR._attention_rows(fold, net.final_fusion_img_point, feats0, R.feature_gather_rows(fused_img, pts_xy))
        B, N = pts_xy.shape[0], pts_xy.shape[1]
        ncls = rpn.rpn_cls_layer[-1].conv.out_channels
        nreg = rpn.rpn_reg_layer[-1].conv.out_channels
        rpn_cls = TR._head_rows(fold, rpn.rpn_cls_layer, feats)[:, :ncls].reshape(B, N, ncls).contiguous()
        rpn_reg = TR._head_rows(fold, rpn.rpn_reg_layer, feats)[:, :nreg].reshape(B, N, nreg).contiguous()
        return feats, rpn_cls, rpn_reg

    def _pooled_fn(self, xyz, rpn_cls, rpn_reg, feats):
        out = dict(rpn_cls=rpn_cls, rpn_reg=rpn_reg, feature_rows=feats)
        rois, pts_input, count = TR.pooled_rois(self.engine, xyz, out, self.rois_per_frame)
        return rois, pts_input.contiguous(), count.contiguous()

    def _rcnn_fn(self, pts_input, count, *ws):
        fold = TR.BnFold(None, self.p_rcnn[0])
        out = TR.rcnn_forward_rows(self.engine, pts_input, fold, count)
        return out["rcnn_cls"].contiguous(), out["rcnn_reg"].contiguous(), out["rcnn_feat"]

    # ------------------------------------------------------------------------------------------------------------------ the step
    def geometry(self, xyz, pts_xy):
        """this batch's geometry in the step's own buffers; `announce` must have been called for (xyz, pts_xy), else it runs now"""
        dev = xyz.device
        self._shape = tuple(xyz.shape)
        main = torch.cuda.current_stream(dev)
        fs = side_stream(dev, 0) if self.engine.overlap else main
        hit = self._announced
        if hit is None or hit[0] is not xyz or hit[1] is not pts_xy:
            self._launch_geometry(xyz, pts_xy)
            hit = self._announced
        self._announced = None
        main.wait_event(hit[2])
        with torch.no_grad():
            taken = self.sec_take(*self._geom_out) if self.sec_take.enabled else self._geom_out
        if fs is not main:
            fs.wait_stream(main)          # the next announcement overwrites what `geometry_take` has just read
        return taken

    def _launch_geometry(self, xyz, pts_xy):
        dev = xyz.device
        main = torch.cuda.current_stream(dev)
        fs = side_stream(dev, 0) if self.engine.overlap else main
        if fs is not main:
            fs.wait_stream(main)
        with torch.cuda.stream(fs), torch.no_grad():
            self._geom_out = self.sec_geom(xyz, pts_xy)
            ev = torch.cuda.Event()
            ev.record(fs)
        self._announced = (xyz, pts_xy, ev)

    def announce(self, xyz, pts_xy):
        """the NEXT batch's cloud: its geometry starts now on stream F, under this step"""
        if self.engine.overlap and xyz is not None:
            self._launch_geometry(xyz, pts_xy)

    def forward_backbone(self, xyz, image, pts_xy, geom) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        """backbone + RPN heads: (feature rows (B N, C), rpn_cls (B, N, 1), rpn_reg (B, N, C))"""
        eng = self.engine
        dev = xyz.device
        main = torch.cuda.current_stream(dev)
        img_stream = side_stream(dev, 1) if eng.overlap else main
        levels, interp = self._geom_unpack(geom)
        maps, events = [], []

        def image_block(i):
            if i == 0 and img_stream is not main:
                img_stream.wait_stream(main)
            with torch.cuda.stream(img_stream):
                m = self.sec_img[i](image if i == 0 else maps[i - 1], *self.p_img[i][1])
                ev = torch.cuda.Event()
                ev.record(img_stream)
            maps.append(m)
            events.append(ev)

        l_xyz, l_feats = [xyz], [None]
        for i in range(self.nlev):
            image_block(i)
            new_xyz, xy_i, plans = levels[i]
            main.wait_event(events[i])
            args = [l_xyz[i]] + ([l_feats[i]] if i > 0 else []) + [new_xyz, xy_i]
            for p in plans:
                args += [p.d, p.offsets, p.row_point, p.row_group]
            args += [maps[i]] + self.p_level[i][1]
            l_feats.append(self.sec_level[i](*args))
            l_xyz.append(new_xyz)
        with torch.cuda.stream(img_stream):
            fused = self.sec_fmap(*maps, *self.p_fmap[1])
            fused_ev = torch.cuda.Event()
            fused_ev.record(img_stream)
        flat_interp = [t for pair in interp for t in pair]
        feats0 = self.sec_fp(*l_feats[1:], *flat_interp, *self.p_fp[1])
        main.wait_event(fused_ev)
        return self.sec_final(feats0, fused, pts_xy, *self.p_final[1])

    def captures(self) -> Dict[str, int]:
        secs = [self.sec_geom, self.sec_take, *self.sec_img, self.sec_fmap, *self.sec_level, self.sec_fp, self.sec_final,
                self.sec_pooled, self.sec_rcnn]
        return {s.name: s.captures for s in secs}


_graphs = {}


def joint_graphs(engine) -> JointGraphs:
    import weakref
    from ._registry import EPOCH
    hit = _graphs.get(id(engine))
    if hit is None or hit[0] != EPOCH[0] or hit[1]() is not engine:
        hit = _graphs[id(engine)] = (EPOCH[0], weakref.ref(engine), JointGraphs(engine))
    return hit[2]


def forward_backward(engine, xyz, image, pts_xy, gt_tids, world, local, rois_per_frame, next_xyz, next_xy=None):
    """forward, thin loss and backward of the joint-mode step on sections; returns (local loss (device scalar), outputs).
    Issued in the order that keeps the streams busy: backbone + RPN heads forward (M / I) -> proposals + RoI pooling (no
    gradient) -> the BACKWARD of the RPN part of the loss (M / I: the loss is a sum, its parts back-propagate independently and
    share no section) -> RCNN forward, re-id loss and their backward on stream R, under the backbone's backward."""
    import os
    import time
    import torch.distributed as tdist
    from . import dist as jdist
    from .ops.affinity_train import AffinityTrainState, affinity_train_loss
    trace = bool(os.environ.get("JM_STEP_TRACE"))
    marks = [("start", time.perf_counter())]
    if trace:
        from . import graphed
        torch.cuda.synchronize()
        graphed.TIMELINE = []
        t_ref = torch.cuda.Event(enable_timing=True)
        t_ref.record()

    def mark(name):
        if trace:
            marks.append((name, time.perf_counter()))
    jg = joint_graphs(engine)
    jg.rois_per_frame = rois_per_frame
    dev = xyz.device
    main = torch.cuda.current_stream(dev)
    geom = jg.geometry(xyz, pts_xy)
    mark("geometry_take")
    if next_xyz is not None:
        jg.announce(next_xyz, pts_xy if next_xy is None else next_xy)
    mark("announce")
    feats, rpn_cls, rpn_reg = jg.forward_backbone(xyz, image, pts_xy, geom)
    mark("backbone_forward")
    with torch.no_grad():
        rois, pts_input, count = jg.sec_pooled(xyz, rpn_cls, rpn_reg, feats)
    pooled_ev = torch.cuda.Event()
    pooled_ev.record()
    mark("pooled")
    n = float(rpn_cls.shape[1])
    rpn_loss = (rpn_cls.sum() + rpn_reg.sum()) / n
    rcnn_first = bool(int(os.environ.get("JM_JOINT_RCNN_FIRST", "0")))
    if not rcnn_first:
        rpn_loss.backward()
        mark("rpn_backward")
    side = side_stream(dev, 3) if engine.overlap else main
    if side is not main:
        side.wait_event(pooled_ev)            # not main's position NOW: that would put the RCNN behind the backbone's whole backward
    B = gt_tids.shape[0]
    with torch.cuda.stream(side):
        rcnn_cls, rcnn_reg, rcnn_feat = jg.sec_rcnn(pts_input, count, *jg.p_rcnn[1])
        st = AffinityTrainState(rcnn_feat.view(B, -1, rcnn_feat.shape[-1]), gt_tids)
        counts = None
        if jdist.collective_path(world, local):      # the re-id means run over the GLOBAL element counts (as in the finetune step)
            counts = st.counts.clone()
            tdist.all_reduce(counts, op=tdist.ReduceOp.SUM)
        reid = affinity_train_loss(st, engine.rcnn_net.link_layer, engine.rcnn_net.se_layer, counts=counts)
        rcnn_loss = rcnn_cls.sum() + rcnn_reg.sum() + reid
        rcnn_loss.backward()
        total = rcnn_loss.detach() + rpn_loss.detach()
    mark("rcnn")
    if rcnn_first:
        rpn_loss.backward()
        mark("rpn_backward")
    # every stream that ran a piece of the step is joined before the all-reduce / optimizer on the main stream
    if side is not main:
        gt_tids.record_stream(side)
        total.record_stream(main)
    for slot in (1, 3):
        if engine.overlap:
            main.wait_stream(side_stream(dev, slot))
    if trace:
        t0 = marks[0][1]
        print("[step] host: " + "  ".join(f"{nm} {1e3 * (t - t0):.2f}" for nm, t in marks[1:]), flush=True)
        torch.cuda.synchronize()
        tl, graphed.TIMELINE = graphed.TIMELINE, None
        spans = {}
        for name, what, ev in tl:
            spans.setdefault((name, what[:3]), []).append(t_ref.elapsed_time(ev))
        print("[step] device (ms after the step's start; section: begin-end): " +
              "  ".join(f"{k[0]}.{k[1]} {v[0]:.2f}-{v[-1]:.2f}" for k, v in sorted(spans.items(), key=lambda kv: kv[1][0])), flush=True)
    return total, dict(rois=rois, rpn_cls=rpn_cls, rpn_reg=rpn_reg, backbone_features=feats, rcnn_cls=rcnn_cls, rcnn_reg=rcnn_reg,
                       rcnn_feat=rcnn_feat)
