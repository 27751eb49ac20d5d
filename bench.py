"""bench.py — throughput of JMODT's detection + association hot path on MI355X.

Default workload = the thing BASELINE.json's metric names: the composed detect + affinity forward
(BASELINE configs[2] "Full RPN + LI-Fusion + roipool3d + iou3d_nms forward, KITTI-shape batch=8" plus the
pairwise link / start-end affinity of configs[0]) on synthetic KITTI-shaped frames resident in HBM:
16384 points, 384x1280 image canvas (native 375x1242 zero padded), 128 proposals per frame.  One "step" =
one batch of 8 frames through jmodt_amd.detector.DetectAffinityEngine.  value = frames/s, whole job.

    python bench.py [--gpus N --steps K --warmup W] [--workload detect|sa|ops|dense|train]

  detect  configs[2] + affinity (default)
  sa      configs[1]: FPS + dual ball_query + group_points over the four RPN SA levels
  ops     every hot-path op once at its SURVEY.md §8(d) shape
  dense   configs[4] shapes: 65536-pt clouds, 256 RoIs, 256^2 affinity
  train   configs[3]: frozen detector forward + data-parallel finetune step of the link / start-end heads
          (one bucketed fp32 gradient all-reduce over RCCL)

N > 1 runs one rank per GPU under torch.distributed.run — either launched that way by the caller (RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_* in the environment) or, from a plain shell (`python bench.py --gpus 8`), by bench.py re-executing
itself under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (the process-per-GPU
equivalent of the reference's nn.DataParallel, tools/train.py:86-88).  Frames are independent, so every rank processes
its own batch with no data-path collective (weak scaling); the timed region is bracketed by
barrier + synchronize and the MAX over ranks is used.

One COMPACT JSON line (<= 4 KB) on rank 0's stdout: the contract's keys, `roofline`, `roofline_by_time`, `cpu_baseline`, the
per-cloud values; the FULL record goes to bench_out/<workload>.json (--full-out).  There, `kernels` lists every C-ABI entry point of the step with its algorithmic bytes /
flops (jmodt_amd/profile.py: SURVEY.md §8(d) formulas evaluated on the call's own arguments) and its time from
HIP events recorded on the launching stream inside the timed region, plus caller-side spans (`name(MIOpen)` /
`name(rocBLAS)` = library calls; `name(span)` = a stage that wraps jm entries listed on their own) and the exposed
waits on the FPS / image side streams; `roofline` is the dominant jm
entry; `cpu_baseline` is the chained CPU oracle (oracle/pipeline.py: the C restatement for the jmodt ops + the
same PyTorch-CPU operators the reference calls for everything else) on a bounded sample.
"""
import argparse
import gc
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from jmodt_amd import synth  # noqa: E402
from jmodt_amd.profile import prof  # noqa: E402

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)
HBM_COPY_CEILING_GBS = 6290.0
MFMA_F32_PEAK_TF = 157.3
METRIC = "frames/sec detect+affinity on 16384-pt KITTI frames; per-kernel HBM-BW fraction"

# jmodt/config.py:75-77 (RPN.SA_CONFIG) + channel widths entering each level (config.py:78-82)
SA_LEVELS = [
    dict(n=16384, m=4096, radii=(0.1, 0.5), ns=(16, 32), c=0),
    dict(n=4096, m=1024, radii=(0.5, 1.0), ns=(16, 32), c=96),
    dict(n=1024, m=256, radii=(1.0, 2.0), ns=(16, 32), c=256),
    dict(n=256, m=64, radii=(2.0, 4.0), ns=(16, 32), c=512),
]


# ---------------------------------------------------------------------------------------------- detect
def detect_inputs(B, seed, dev, tiny=False, kind="uniform", points=16384):
    if tiny:
        xyz, img, xy = synth.frames(B, 2048, seed, H=96, W=320, native=(94, 310), kind=kind)
    else:
        xyz, img, xy = synth.frames(B, points, seed, kind=kind)
    return dict(xyz=torch.from_numpy(xyz).to(dev), image=torch.from_numpy(img).to(dev), pts_xy=torch.from_numpy(xy).to(dev))


def make_detect_state(B, seed, dev, tiny=False, kind="uniform", points=16384, rois=None):
    import dataclasses
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    torch.manual_seed(seed)
    cfg = DetectorConfig.tiny() if tiny else DetectorConfig.survey()
    if rois and not tiny:
        cfg = dataclasses.replace(cfg, rpn_post_nms_top_n=rois)
    eng = DetectAffinityEngine(cfg).to(dev)
    return dict(engine=eng, **detect_inputs(B, seed, dev, tiny, kind, points))


def _with_headline(clouds, value):
    if clouds and "uniform" in clouds:
        clouds["uniform"]["value"] = value
    return clouds


def rcnn_rows():
    """(dense rows, executed rows) of the duplicate-compacted RCNN scales of the LAST forward (device counters: call after a
    synchronize)"""
    from jmodt_amd.ops.pointnet2 import fused
    return {e[0]: {"rows_dense": int(e[1]), "rows_executed": fused.DedupeStats.rows_executed(e)} for e in fused.DedupeStats.last}


def listed_rows():
    """{bench row: rows dense / executed} of the listed RPN scales of the LAST forward (class counts in device memory: call after a
    synchronize)"""
    from jmodt_amd.ops.pointnet2 import fused
    out, seen = {}, set()
    for name, dense, ns, plan in reversed(fused.ListedStats.last):       # newest first: one entry per (row, scale) = the LAST step's
        if (name, ns) in seen:
            continue
        seen.add((name, ns))
        pl = plan[:8].tolist()
        r = out.setdefault(name, {"rows_dense": 0, "rows_executed": 0})
        r["rows_dense"] += int(dense)
        r["rows_executed"] += sum(int(pl[c]) << c for c in range(8))
    return out


def detect_step(st):
    """one batch; the NEXT batch's cloud is announced so that its FPS pyramid runs under this batch's work (a
    streaming detector always knows its next batch; here it is the same resident synthetic batch).  Every step
    still launches exactly one FPS pyramid and one of everything else inside the timed region."""
    pf = st.get("prefetch", True)
    return st["engine"](st["xyz"], st["image"], st["pts_xy"], next_xyz=_upcoming(st) if pf else None, next_image=st["image"] if pf else None)


def streaming_rate(st, n_steps, world, batch):
    """the same composed step fed the way tools/eval.py feeds it: a NEW batch every step, from pinned host memory, its H2D copy
    (47 MB of image + 1.6 MB of points for 8 frames) on a copy stream under the previous step.  Two device buffer sets alternate
    (the engine's next-batch announcement names the very tensors the next call receives).  Returns frames/s over `n_steps`."""
    eng = st["engine"]
    dev = st["xyz"].device
    keys = ("xyz", "image", "pts_xy")
    host = [{k: st[k].cpu().pin_memory() for k in keys} for _ in range(2)]
    host[1]["xyz"] = host[1]["xyz"].flip(0).contiguous().pin_memory()          # (a different batch: the frames in another order)
    host[1]["image"] = host[1]["image"].flip(0).contiguous().pin_memory()
    host[1]["pts_xy"] = host[1]["pts_xy"].flip(0).contiguous().pin_memory()
    bufs = [{k: torch.empty_like(st[k]) for k in keys} for _ in range(2)]
    copy = torch.cuda.Stream(device=dev)
    done = [torch.cuda.Event(), torch.cuda.Event()]      # H2D of buffer set i complete
    free = [torch.cuda.Event(), torch.cuda.Event()]      # the step that read buffer set i has been issued (recorded on the main stream)
    main = torch.cuda.current_stream(dev)

    def upload(i, step_no):
        copy.wait_event(free[i])                         # (recorded: set i's last reader is behind us on the main stream)
        with torch.cuda.stream(copy):
            for k in keys:
                bufs[i][k].copy_(host[step_no % 2][k], non_blocking=True)
            done[i].record(copy)

    for i in range(2):
        free[i].record(main)
    upload(0, 0)
    pf = st.get("prefetch", True)

    def run(step_no):
        i = step_no % 2
        upload(1 - i, step_no + 1)                       # the NEXT batch travels while this one is computed
        main.wait_event(done[i])
        b, nb = bufs[i], bufs[1 - i]
        out = eng(b["xyz"], b["image"], b["pts_xy"], next_xyz=[nb["xyz"]] if pf else None, next_image=None)
        free[i].record(main)
        return out
    for w in range(2):
        run(w)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(2, 2 + n_steps):
        run(k)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    eng._drop_kept()
    return world * batch * n_steps / dt


def _upcoming(st):
    """the clouds of the next `prefetch_depth` batches (the same resident synthetic batch each): one FPS pyramid is started per
    step whatever the depth, it is only started earlier"""
    return [st["xyz"]] * max(1, int(st["engine"].prefetch_depth))


def _liven(eng, seed):
    """non-trivial BatchNorm statistics / biases and head weights that produce a spread of scores and visible box
    regression (the default initialisation gives every point the same objectness to ~1e-3: proposal selection would be
    decided by rounding noise and a per-stage comparison with the CPU chain would stop at the first discrete decision)"""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for m in eng.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.running_mean.copy_(torch.randn(m.running_mean.shape, generator=g) * 0.1)
                m.running_var.copy_(torch.rand(m.running_var.shape, generator=g) + 0.5)
                m.weight.copy_(torch.rand(m.weight.shape, generator=g) + 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g) * 0.1)
        eng.rpn.rpn_cls_layer[2].conv.bias.zero_()
        eng.rpn.rpn_cls_layer[2].conv.weight.mul_(8.0)
        eng.rpn.rpn_reg_layer[2].conv.weight.copy_(torch.randn(eng.rpn.rpn_reg_layer[2].conv.weight.shape, generator=g) * 0.3)
        eng.rcnn_net.reg_layer[-1].conv.weight.copy_(torch.randn(eng.rcnn_net.reg_layer[-1].conv.weight.shape, generator=g) * 0.3)
        eng.rcnn_net.cls_layer[-1].conv.weight.mul_(6.0)


def cpu_baseline_detect(frames=2, dev=None):
    """the chained CPU oracle (float32, the reference's arithmetic) on `frames` full-size frames; with `dev`, the engine
    then runs the SAME frames with the SAME weights on the GPU and every stage's output is compared with the chain's
    (free running on both sides: stages behind a discrete decision are compared only when the decision was identical)"""
    from jmodt_amd.detector import DetectAffinityEngine, DetectorConfig
    from oracle.pipeline import Chain
    torch.manual_seed(99)
    cfg = DetectorConfig.survey()
    eng = DetectAffinityEngine(cfg)
    _liven(eng, 100)
    sd = eng.state_dict()
    xyz, img, xy = synth.frames(frames, 16384, 4321)
    chain = Chain(sd, cfg, torch.float32)
    t0 = time.perf_counter()
    with torch.no_grad():
        want = chain.forward(xyz, img, xy)
    dt = time.perf_counter() - t0
    # BASELINE configs[0] on its own: the link / start-end head on 64 cached proposal features, PyTorch-CPU
    f64 = torch.relu(torch.randn(2, 64, cfg.rcnn_sa_mlps[-1][-1]))
    with torch.no_grad():                                   # 3 warm-up passes, median of 10
        for _ in range(3):
            chain.affinity(f64[0], f64[1])
        ts = []
        for _ in range(10):
            t1 = time.perf_counter()
            chain.affinity(f64[0], f64[1])
            ts.append(time.perf_counter() - t1)
        aff64 = sorted(ts)[len(ts) // 2]
    parity = None
    if dev is not None:
        eng = eng.to(dev)
        with torch.no_grad():
            cache, aff, inter = eng(torch.from_numpy(xyz).to(dev), torch.from_numpy(img).to(dev), torch.from_numpy(xy).to(dev))
        torch.cuda.synchronize()

        def err(got, ref):
            got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else np.asarray(got)
            ref = ref.detach().float().cpu().numpy() if isinstance(ref, torch.Tensor) else np.asarray(ref)
            return {"max_abs_err": float(np.abs(got.astype(np.float64) - ref.astype(np.float64)).max()), "max_abs_ref": float(np.abs(ref).max())}
        parity = {"note": "GPU engine vs the float32 CPU chain, same weights and frames, both free running; float64 teacher-forced "
                          "bars are in tests/test_gpu_detector.py::test_full_width_*",
                  "fps_indices_identical": all(np.array_equal(a.cpu().numpy(), b) for a, b in zip(eng.last_fps_idx, chain.last["fps_idx"])),
                  "backbone_features": err(inter["backbone_features"], want["backbone_features"]),
                  "rpn_cls": err(inter["rpn_cls"], want["rpn_cls"]), "rpn_reg": err(inter["rpn_reg"], want["rpn_reg"])}
        # the proposal layer is a discrete decision (top-k by score, NMS): the two sides took the SAME decisions when every
        # RoI slot holds the same box up to decode rounding (float32 sin / cos / atan2 differ in the last bits CPU vs GPU)
        roi_err = np.abs(inter["rois"].cpu().numpy().astype(np.float64) - want["rois"].astype(np.float64)).max(-1)
        same_rois = bool((roi_err < 1e-3).all())
        parity["rois_same_selection"] = same_rois
        parity["rois_matching_slots"] = float((roi_err < 1e-3).mean())
        parity["rois_max_abs_err_on_matching_slots"] = float(roi_err[roi_err < 1e-3].max()) if (roi_err < 1e-3).any() else None
        if same_rois:
            B, M = want["rois"].shape[:2]
            parity["roipool_pts_input"] = err(inter["pts_input"], want["pts_input"])
            for k in ("rcnn_feat", "rcnn_cls", "rcnn_reg"):
                parity[k] = err(inter[k], want[k])
            parity["pred_boxes3d"] = err(inter["pred_boxes3d"], want["pred_boxes3d"])
            counts = cache.counts_host()
            parity["detection_keep_identical"] = all(
                np.array_equal(cache.roi_index[b, :counts[b]].cpu().numpy(), want["keep"][b]) for b in range(B))
            parity["affinity"] = {"max_abs_err": max(err(aff[b][0], want["affinity"][b][0])["max_abs_err"] for b in range(B)),
                                  "start_end_max_abs_err": max(max(err(aff[b][1], want["affinity"][b][1])["max_abs_err"],
                                                                   err(aff[b][2], want["affinity"][b][2])["max_abs_err"]) for b in range(B))}
    return frames / dt, dt, chain.stage_seconds, aff64, parity


# ---------------------------------------------------------------------------------------------- sa
def make_sa_inputs(B, seed, dev):
    xyz = torch.from_numpy(synth.cloud(B, 16384, seed=seed)).to(dev)
    g = torch.Generator(device="cpu").manual_seed(seed)
    feats = [None] + [torch.randn(B, lv["c"], lv["n"], generator=g).to(dev) for lv in SA_LEVELS[1:]]
    return xyz, feats


def sa_step(xyz, feats, overlap=True, ahead=None):
    """FPS + dual ball_query + group_points (xyz and features, both scales) over the 4 levels; the FPS chain
    (coordinates only) runs ahead on a side stream (ops/pointnet2/pyramid.py).
    ahead = {"depth": D, "fifo": [], "n": 0} kept between steps: the FPS pyramids of the next D batches (the same resident cloud)
    are in flight, each chain on a side stream of its own — every step still starts exactly one pyramid, D steps early"""
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.pointnet2.pyramid import FpsPyramid
    npts = [lv["m"] for lv in SA_LEVELS]
    if ahead is not None and overlap and ahead["depth"] > 0:
        fifo, slots = ahead["fifo"], (0, 4, 5, 6)
        while len(fifo) < ahead["depth"] + 1:
            fifo.append(FpsPyramid(xyz, npts, overlap=True, slot=slots[ahead["n"] % min(len(slots), ahead["depth"] + 1)]))
            ahead["n"] += 1
        pyr = fifo.pop(0)
    else:
        pyr = FpsPyramid(xyz, npts, overlap=overlap)
    outs, cur = [], xyz
    for li, lv in enumerate(SA_LEVELS):
        (r0, r1), (ns0, ns1), c = lv["radii"], lv["ns"], lv["c"]
        _, new_xyz = pyr.level(li)
        cur_t = cur.transpose(1, 2).contiguous()      # (B, 3, n) layout for group_points
        with prof.scope(f"L{li + 1}"):
            i0, i1 = pu.ball_query_dual(r0, ns0, r1, ns1, cur, new_xyz)
            for nb in (i0, i1):
                with prof.scope("xyz"):
                    outs.append(pu.grouping_operation(cur_t, nb))
                if c:
                    with prof.scope("feat"):
                        outs.append(pu.grouping_operation(feats[li], nb))
        cur = new_xyz
    pyr.release()
    return outs


def cpu_baseline_sa(B):
    """the oracle (CPU restatement, OpenMP) on ONE batch of the sa workload"""
    from oracle import oracle as orc
    xyz = synth.cloud(B, 16384, seed=4321)
    rng = np.random.default_rng(0)
    feats = [None] + [rng.normal(size=(B, lv["c"], lv["n"])).astype(np.float32) for lv in SA_LEVELS[1:]]
    orc.lib()
    t0 = time.perf_counter()
    cur = xyz
    for li, lv in enumerate(SA_LEVELS):
        idx = orc.furthest_point_sample(cur, lv["m"])
        cur_t = np.ascontiguousarray(cur.transpose(0, 2, 1))
        new_xyz = np.ascontiguousarray(orc.gather_operation(cur_t, idx).transpose(0, 2, 1))
        for r, ns in zip(lv["radii"], lv["ns"]):
            nb = orc.ball_query(r, ns, cur, new_xyz)
            orc.grouping_operation(cur_t, nb)
            if lv["c"]:
                orc.grouping_operation(feats[li], nb)
        cur = new_xyz
    dt = time.perf_counter() - t0
    return B / dt, dt


# ---------------------------------------------------------------------------------------------- ops
def make_ops_inputs(B, seed, dev, small=False):
    """inputs for the hot-path ops at SURVEY.md §8d shapes (config 3 sizes, M_roi = 128)"""
    from jmodt_amd.ops.affinity import make_affinity_mlp
    from jmodt_amd.ops.pointnet2.pointnet2_modules import PointnetSAModule
    N = 2048 if small else 16384
    sc = 8 if small else 1
    g = torch.Generator(device="cpu").manual_seed(seed)
    xyz_np = synth.cloud(B, N, seed=seed)
    d = dict(xyz=torch.from_numpy(xyz_np).to(dev), N=N)
    d["xy"] = torch.from_numpy(synth.pts_xy(xyz_np)).to(dev)
    d["M"] = 16 if small else 128
    d["boxes"] = torch.from_numpy(synth.proposals(xyz_np, d["M"], seed + 1)).to(dev)
    d["feat130"] = torch.randn(B, N, 130, generator=g).to(dev)
    d["fp"] = []
    for n, m, c in ((256, 64, 1024), (1024, 256, 512), (4096, 1024, 512), (16384, 4096, 256)):
        n, m = n // sc, max(m // sc, 3)
        d["fp"].append((d["xyz"][:, :n].contiguous(), d["xyz"][:, :m].contiguous(), torch.randn(B, c // sc, m, generator=g).to(dev)))
    d["maps"] = [(torch.randn(B, c // sc, h // sc, w // sc, generator=g).to(dev), d["xy"][:, :n // sc].contiguous())
                 for c, h, w, n in ((64, 192, 640, 4096), (128, 96, 320, 1024), (256, 48, 160, 256),
                                    (512, 24, 80, 64), (32, 384, 1280, 16384))]
    d["maps_cl"] = [(fm.contiguous(memory_format=torch.channels_last), xy) for fm, xy in d["maps"]]
    nb = 6300 // sc
    d["bev"], d["scores"] = [], []
    for b in range(B):
        bb, ss = synth.bev_boxes(nb, seed + 10 + b)
        d["bev"].append(torch.from_numpy(bb).to(dev)); d["scores"].append(torch.from_numpy(ss).to(dev))
    rs, rp = synth.rpn_output(B, N, seed + 20)
    d["rpn_scores"], d["rpn_props"] = torch.from_numpy(rs).to(dev), torch.from_numpy(rp).to(dev)
    torch.manual_seed(seed)
    R = B * d["M"]
    d["roi_xyz"] = (torch.rand(R, 512, 3, generator=g) - 0.5).mul_(torch.tensor([4.0, 2.0, 2.0])).to(dev)
    d["roi_feat"] = torch.randn(R, 128, 512, generator=g).to(dev)
    d["rcnn_sa1"] = PointnetSAModule(mlp=[128, 128, 128, 128], npoint=128, radius=0.2, nsample=64, bn=False).to(dev).eval()
    d["link"], d["se"] = make_affinity_mlp().to(dev).eval(), make_affinity_mlp().to(dev).eval()
    d["pf256"] = torch.from_numpy(synth.roi_features(256, 512, seed + 2)).to(dev)
    d["df256"] = torch.from_numpy(synth.roi_features(256, 512, seed + 3)).to(dev)
    return d


def ops_step(d):
    from jmodt_amd.ops.affinity import pairwise_affinity, pairwise_affinity_batched
    from jmodt_amd.ops.fusion import feature_gather
    from jmodt_amd.ops.iou3d.iou3d_utils import nms_normal_gpu
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.proposal import distance_based_proposal
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu, roipool3d_gpu
    B = d["xyz"].shape[0]
    for li, (unknown, known, feats) in enumerate(d["fp"]):
        with prof.scope(f"FP{li + 1}"):
            dist, idx = pu.three_nn(unknown, known)
            w = 1.0 / (dist + 1e-8)
            pu.three_interpolate(feats, idx, w / w.sum(2, keepdim=True))
    for mi, (fm, xy) in enumerate(d["maps"]):
        with prof.scope(f"map{mi + 1}_nchw"):
            feature_gather(fm, xy)
    for mi, (fm, xy) in enumerate(d["maps_cl"]):   # same maps in channels_last memory format (no copy inside the op)
        with prof.scope(f"map{mi + 1}_channels_last"):
            feature_gather(fm, xy)
    roipool3d_gpu(d["xyz"], d["feat130"], d["boxes"], 0.2, 512)
    roipool3d_canonical_gpu(d["xyz"], d["feat130"], d["boxes"], 0.2, 512)
    with prof.scope("single"):
        for b in range(B):
            nms_normal_gpu(d["bev"][b], d["scores"][b], 0.8)
    distance_based_proposal(d["rpn_scores"], d["rpn_props"], 9000, 100, 0.8, "normal")
    from jmodt_amd.ops.pointnet2 import fused as _fused
    keep_listed, _fused.LISTED = _fused.LISTED, False
    try:
        with torch.no_grad(), prof.scope("rcnn_sa1"):            # the DENSE kernel at the SURVEY.md §8d shape (831 GFLOP per 1024 RoIs)
            d["rcnn_sa1"](d["roi_xyz"], d["roi_feat"])
    finally:
        _fused.LISTED = keep_listed
    with torch.no_grad(), prof.scope("rcnn_sa1_listed"):         # the same call in the duplicate-aware form (what a caller gets)
        d["rcnn_sa1"](d["roi_xyz"], d["roi_feat"])
    for P in (64, 128, 256):        # the affinity head at BASELINE configs[0] / [2] / [4] sizes, one problem each
        with prof.scope(f"affinity_{P}x{P}"):
            pairwise_affinity(d["pf256"][:P], d["df256"][:P], d["link"], d["se"])
    with prof.scope(f"affinity_batched_{B}x128x128"):
        pairwise_affinity_batched(d["pf256"][:128].unsqueeze(0).expand(B, -1, -1), d["df256"][:128].unsqueeze(0).expand(B, -1, -1),
                                  d["link"], d["se"])


# ---------------------------------------------------------------------------------------------- dense
def make_dense_inputs(B, seed, dev, small=False):
    """BASELINE configs[4] shapes: 65536 points per frame, 256 proposals, 256 x 256 affinity"""
    from jmodt_amd.ops.affinity import make_affinity_mlp
    N = 32768 if small else 65536
    M = 32 if small else 256
    xyz_np = synth.cloud(B, N, seed=seed)
    g = torch.Generator(device="cpu").manual_seed(seed)
    d = dict(xyz=torch.from_numpy(xyz_np).to(dev), N=N, M=M, m=512 if small else 4096)
    d["boxes"] = torch.from_numpy(synth.proposals(xyz_np, M, seed + 1)).to(dev)
    d["feat130"] = torch.randn(B, N, 130, generator=g).to(dev)
    torch.manual_seed(seed)
    d["link"], d["se"] = make_affinity_mlp().to(dev).eval(), make_affinity_mlp().to(dev).eval()
    d["pf"] = torch.from_numpy(synth.roi_features(M, 512, seed + 2)).to(dev)
    d["df"] = torch.from_numpy(synth.roi_features(M, 512, seed + 3)).to(dev)
    return d


def dense_step(d):
    from jmodt_amd.ops.affinity import pairwise_affinity_batched
    from jmodt_amd.ops.pointnet2 import pointnet2_utils as pu
    from jmodt_amd.ops.roipool3d.roipool3d_utils import roipool3d_canonical_gpu
    xyz, m = d["xyz"], d["m"]
    B = xyz.shape[0]
    idx, new_xyz = pu.farthest_point_sample_xyz(xyz, m)
    xyz_t = xyz.transpose(1, 2).contiguous()
    i0, i1 = pu.ball_query_dual(0.1, 16, 0.5, 32, xyz, new_xyz)
    for nb in (i0, i1):
        pu.grouping_operation(xyz_t, nb)
    pu.three_nn(xyz, new_xyz)
    roipool3d_canonical_gpu(xyz, d["feat130"], d["boxes"], 0.2, 512)
    pairwise_affinity_batched(d["pf"].unsqueeze(0).expand(B, -1, -1), d["df"].unsqueeze(0).expand(B, -1, -1), d["link"], d["se"])


# ---------------------------------------------------------------------------------------------- train
def joint_route():
    """the route joint_step takes for this process (JM_JOINT_ROUTE: rows (default) | operators | auto)"""
    r = os.environ.get("JM_JOINT_ROUTE", "rows")
    return "rows" if r == "auto" else r


def make_rcnn_state(frames, seed, dev, tiny=False, kind="uniform"):
    """BASELINE configs[3] in the reference's DEFAULT training mode (tools/train.py:86-107 with the shipped config.py:57
    RPN.FIXED = True, FINETUNE off): the RPN frozen and evaluated without gradient (point_rcnn.py:28-31), the RCNN and the re-id
    heads train; the step is jmodt_amd/train_joint.rcnn_step"""
    from jmodt_amd import train_joint
    st = make_detect_state(frames, seed, dev, tiny=tiny, kind=kind)
    eng = st["engine"]
    train_joint.prepare_rcnn(eng)
    g = torch.Generator(device="cpu").manual_seed(seed)
    rois_per_frame = min(64, eng.cfg.rpn_post_nms_top_n)
    st["tids"] = torch.randint(0, 13, (frames, rois_per_frame), generator=g).float().to(dev)
    st["opt"] = torch.optim.Adam(train_joint.rcnn_parameters(eng), lr=2e-4, weight_decay=1e-2, fused=True)
    st["rois_per_frame"] = rois_per_frame
    st["rcnn"] = True
    return st


def make_joint_state(frames, seed, dev, tiny=False, kind="uniform"):
    """BASELINE configs[3] in JOINT mode (tools/train.py:96-107 without cfg.TRAIN.FINETUNE, RPN.FIXED off): every parameter of the
    detector and of the affinity heads trains; the step is jmodt_amd/train_joint.joint_step"""
    from jmodt_amd import train_joint
    st = make_detect_state(frames, seed, dev, tiny=tiny, kind=kind)
    eng = st["engine"]
    if joint_route() == "rows":
        if "JM_JOINT_CONV_FIND" not in os.environ and not tiny:
            train_joint.CONV_FIND = True     # MIOpen's find mode for the image convolutions, forward and backward (+2 %; seconds at first use)
        train_joint.prepare_rows(eng)    # train mode (RPN-head dropout active), every BatchNorm FROZEN on its running statistics
    else:
        eng.train()                      # the un-fused operator route: BatchNorm on batch statistics
    g = torch.Generator(device="cpu").manual_seed(seed)
    rois_per_frame = min(64, eng.cfg.rpn_post_nms_top_n)
    st["tids"] = torch.randint(0, 13, (frames, rois_per_frame), generator=g).float().to(dev)
    for p in eng.parameters():
        p.requires_grad_(True)
    st["opt"] = torch.optim.Adam(list(eng.parameters()), lr=2e-4, weight_decay=1e-2, fused=True)
    st["rois_per_frame"] = rois_per_frame
    st["joint"] = True
    return st


def make_train_state(frames, seed, dev, tiny=False, kind="uniform"):
    """BASELINE configs[3] per-GPU share: `frames` frames (= frames/2 (prev, next) pairs) through the FROZEN
    detector (cfg.RPN.FIXED + finetune: tools/train.py:96-107 trains only the link / start-end heads), then the
    pairwise affinity losses on the 64 sampled RoIs per frame (config.py:153) and Adam"""
    st = make_detect_state(frames, seed, dev, tiny=tiny, kind=kind)
    eng = st["engine"]
    g = torch.Generator(device="cpu").manual_seed(seed)
    rois_per_frame = min(64, eng.cfg.rpn_post_nms_top_n)
    st["tids"] = torch.randint(0, 13, (frames, rois_per_frame), generator=g).float().to(dev)     # 0 = background, 12 tracks
    link, se = eng.rcnn_net.link_layer.train(), eng.rcnn_net.se_layer.train()
    for p in eng.parameters():
        p.requires_grad_(False)
    for p in list(link.parameters()) + list(se.parameters()):
        p.requires_grad_(True)
    st["opt"] = torch.optim.Adam(list(link.parameters()) + list(se.parameters()), lr=2e-4, weight_decay=1e-2, fused=True)   # one launch
    st["rois_per_frame"] = rois_per_frame
    return st


def _grad_collectives(joint=False):
    """gradient collectives the last step issued (joint / rcnn steps: train_joint; finetune: affinity_train)"""
    if joint:
        from jmodt_amd import train_joint
        return train_joint.LAST_GRAD_COLLECTIVES
    from jmodt_amd.ops import affinity_train
    return affinity_train.LAST_GRAD_COLLECTIVES


def train_step(st, world):
    """one data-parallel finetune step: frozen composed detector forward (no grad) -> 512-d RoI features ->
    local forward/backward of the pairwise affinity losses -> ONE bucketed gradient all-reduce over RCCL -> Adam"""
    from jmodt_amd.ops.affinity_train import finetune_step_static
    eng = st["engine"]
    if st.get("rcnn"):
        from jmodt_amd.train_joint import rcnn_step
        pf = st.get("prefetch", True)
        # the NEXT step's frozen half (RPN forward, proposals, RoI pooling: nothing of it depends on this step's update) is issued
        # under this step's RCNN forward / backward / Adam; every timed step still executes one frozen half and one trainable half
        ahead = pf and st.get("ahead", True)
        return rcnn_step(eng, st["xyz"], st["image"], st["pts_xy"], st["tids"], st["opt"], world=world, rois_per_frame=st["rois_per_frame"],
                         next_xyz=_upcoming(st) if pf else None, next_image=st["image"] if pf else None,
                         next_batch=(st["xyz"], st["image"], st["pts_xy"]) if ahead else None)
    if st.get("joint"):
        from jmodt_amd.train_joint import joint_step
        return joint_step(eng, st["xyz"], st["image"], st["pts_xy"], st["tids"], st["opt"], world=world,
                          rois_per_frame=st["rois_per_frame"], route=joint_route(),
                          next_xyz=st["xyz"] if st.get("prefetch", True) else None)
    with torch.no_grad():
        pf = st.get("prefetch", True)
        _, inter = eng.detect(st["xyz"], st["image"], st["pts_xy"], next_xyz=_upcoming(st) if pf else None, next_image=st["image"] if pf else None)
    B = st["xyz"].shape[0]
    feats = inter["rcnn_feat"].view(B, -1, inter["rcnn_feat"].shape[1])[:, :st["rois_per_frame"]].contiguous()
    # static-shape, sync-free: the host never waits for the device inside a step
    return prof.region("finetune(fwd+bwd+allreduce+adam)", lambda: finetune_step_static(
        feats, st["tids"], eng.rcnn_net.link_layer, eng.rcnn_net.se_layer, st["opt"], world=world))


# ---------------------------------------------------------------------------------------------- main
def self_launch(n_gpus: int) -> int:
    """`python bench.py --gpus N` from a plain shell: the same command line under torch.distributed.run, one rank per GPU on
    this node, rendezvous on 127.0.0.1 (the container hostname may not resolve) at a free port.  The child ranks print the
    ONE JSON line (rank 0) straight to this process's stdout; returns the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    argv = [a for a in sys.argv[1:] if a != "--launch"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if n_gpus == 1:
        env["JM_BENCH_FORCE_DIST"] = "1"         # --launch on one GPU: still create the RCCL communicator, barrier, all-reduce
    return subprocess.call(cmd, env=env)


WORKLOAD_TEXT = {
    "detect": "BASELINE configs[2] + affinity: composed detect+affinity forward (LI-Fusion backbone 4xSA-MSG + 4xFP, RPN "
              "heads, proposal layer, roipool3d+canonical, RCNN 3xSA + heads, box decode, detection NMS, pairwise "
              "affinity of consecutive frames), 16384-pt frames, 384x1280 image canvas, 128 proposals/frame",
    "sa": "BASELINE configs[1]: pointnet2 FPS + ball_query(2 radii) + group_points over the 4 RPN SA levels "
          "(16384->4096->1024->256->64), 16384-pt synthetic clouds; the FPS pyramids of upcoming batches run ahead on side streams",
    "ops": "supplementary: three_nn+interpolate (4 FP levels), LI-Fusion gather (5 maps), roipool3d (128 RoIs x 512 "
           "pts x 133), RPN nms_normal (6300 boxes), proposal selection, fused RCNN SA1, 128x128 affinity, per frame",
    "train": "BASELINE configs[3]: frozen composed detector forward + data-parallel finetune step of the link / "
             "start-end heads (64 RoIs x 512-d per frame, pairwise affinity losses, bucketed fp32 gradient all-reduce, Adam)",
    "train_joint": "BASELINE configs[3], joint mode (every parameter trains): differentiable forward AND backward of the whole detector "
                   "on the hand-written row kernels (csrc/rows_*.hip: set abstraction on the distinct (centre, neighbour) rows, feature "
                   "propagation, LI-Fusion gather + attention, heads; image 3x3 / stride-2 / transposed convolutions on MIOpen), RPN + RCNN "
                   "head sums + re-id loss, bucketed fp32 all-reduce of all 16.7 M parameters (66.9 MB), Adam",
    "train_joint_operators": "BASELINE configs[3], joint mode (every parameter trains), UN-FUSED operator route: torch autograd over the "
                             "(B, C, npoint, nsample) tensors (grouping / interpolation / LI-Fusion gather backward on the jm_*_grad kernels, "
                             "convolutions and BatchNorm on MIOpen's autograd), RPN + RCNN head sums + re-id loss, bucketed fp32 all-reduce "
                             "of all 16.7 M parameters (66.9 MB), Adam",
    "train_rcnn": "BASELINE configs[3], the reference's default training mode (config.py:57 RPN.FIXED, point_rcnn.py:28-31): frozen fused "
                  "RPN forward without gradient -> proposals -> roipool3d -> RCNN forward / backward on the row kernels (csrc/rows_*.hip) + "
                  "re-id heads (csrc/affinity_train.hip), bucketed fp32 all-reduce of the RCNN's + heads' gradients only, Adam",
    "dense_detect": "BASELINE configs[4], composed: the SAME detect+affinity forward as `detect` on 65536-pt frames (co-operative "
                    "FPS, hash-grid ball query / 3-NN at the first level), 256 proposals/frame, 256x256 affinity per frame pair",
    "dense": "supplementary, BASELINE configs[4] shapes: 65536-pt clouds (co-operative FPS -> 4096, dual ball query, "
             "grouping, 3-NN), roipool3d+canonical for 256 RoIs, 256x256 affinity per frame",
}


def workload_key(args):
    if args.workload == "train" and args.joint:
        return "train_joint" if joint_route() == "rows" else "train_joint_operators"
    if args.workload == "train" and getattr(args, "rcnn", False):
        return "train_rcnn"
    return args.workload


def train_mode_keys(args):
    """what a training line times, spelled out in `config`: which route, which loss, where the RoIs
    come from, what the BatchNorms do — so that nobody reads the proxy step as the reference's full training iteration"""
    if args.workload != "train":
        return {}
    proxy = {"loss": "proxy: sums of the head outputs + the re-id loss of rcnn.py:204-287 / train_functions.py:282-329 (the reference's "
                     "classification / regression losses are the caller's: SURVEY.md section 2 out of scope)",
             "proposals": "test-mode first-K of the ProposalLayer stand in for ProposalTargetLayer's GT-sampled RoIs (config.py:153: 64 per frame)"}
    if args.joint:
        rows = joint_route() == "rows"
        return {"mode": "joint (RPN.FIXED off: every parameter trains)", "route": joint_route(),
                "batchnorm": ("frozen: eval-mode running statistics folded into the weights, gamma / beta trainable (the reference's joint "
                              "mode trains BatchNorm on batch statistics: JM_JOINT_ROUTE=operators is that form)" if rows else
                              "train mode: batch statistics, running statistics updated (as the reference's joint mode)"), **proxy}
    if getattr(args, "rcnn", False):
        return {"mode": "rcnn (config.py:57 RPN.FIXED = True, FINETUNE off: the reference's default)",
                "route": "frozen fused engine + rows" + ("" if getattr(args, "no_ahead", False) else
                                                         "; the NEXT step's frozen half is issued under this step's RCNN (each timed step runs one of each)"),
                "batchnorm": "RPN: eval mode, folded (point_rcnn.py:29-30); RCNN: none (config.py RCNN.USE_BN = False)", **proxy}
    return {"mode": "finetune (tools/train.py:96-107: link / start-end heads only)", "route": "frozen fused engine + affinity_train kernels",
            "batchnorm": "eval mode, folded", "loss": "the re-id loss of train_functions.py:282-329 (L1 forms)",
            "proposals": "test-mode first-K of the ProposalLayer (config.py:153: 64 per frame)"}


# the image branch's own kernels: not a SURVEY.md §8 row (the reference calls nn.Conv2d there); priced in `image_branch_kernel`
IMAGE_BRANCH_ENTRIES = ("conv3x3_rgb_bias_relu", "conv3x3_wino_bias_relu", "bias_relu_channels_last")


def image_branch_kernel(kernels):
    """the fused Winograd convolution (csrc/conv_wino.hip), the largest kernel of the image branch, on both bases"""
    k = next((k for k in kernels if k["kernel"].split("/")[-1] == "conv3x3_wino_bias_relu"), None)
    if k is None or not k.get("ms_per_step"):
        return None
    return {"kernel": k["kernel"], "bound": "mfma", "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
            "ms_per_step": k["ms_per_step"], "launches_per_step": k["launches_per_step"],
            "achieved": round(k["executed_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12, 2), "frac": k.get("executed_mfma_frac"),
            "basis": "flops the kernel EXECUTES on the matrix cores: 16 products per 2x2 output tile and (cin, cout) pair "
                     "(Winograd F(2x2, 3x3)); HIP events around the entry on the image stream while the main chain's kernels share "
                     "the machine (isolated: tools/conv_wino_bench.py, DESIGN.md §4)",
            "direct_form": {"tflops": k.get("achieved_tflops"), "frac_of_peak": k.get("mfma_frac"),
                            "basis": "2 * 9 * cin * cout flops per output pixel, what a direct / implicit-GEMM convolution executes: "
                                     "can exceed the peak because 20 of 36 products are never formed"},
            "note": "not a SURVEY.md §8 row (the reference calls nn.Conv2d): listed because it is the largest hand-written kernel of the "
                    "step; `roofline` stays on the §8 path"}


def pick_roofline(kernels, traffic_json, full_table=True):
    """the dominant jm_* entry (caller-side torch spans and stream waits are listed but are not ours to price)"""
    # the FPS chain on its side stream is not on the critical path when the consumer hardly ever waits for it (its exposed
    # share is reported under `overlap`): the roofline kernel is then the largest entry of the main chain
    own = own_rows(kernels, full_table and fps_hidden(kernels))
    if not own:
        return None
    # dominant = the entry that would take longest AT THE ROOFLINE (executed flops / MFMA peak, algorithmic bytes / HBM peak):
    # HIP-event times of the small kernels of the overlapped pipeline include waiting behind the image branch's convolutions
    # on the other stream, so "largest measured time" would pick whichever kernel queued longest, not the most work
    def sol_ms(k):
        fl = k.get("executed_flops_per_step", k.get("algo_flops_per_step", 0))
        by = 0 if k["kernel"].startswith("fps_pyramid/") or "evals_per_s" in k else k.get("algo_bytes_per_step", 0)
        return max(fl / (MFMA_F32_PEAK_TF * 1e12), by / (HBM_PEAK_GBS * 1e9)) * 1e3
    own.sort(key=lambda k: -sol_ms(k))
    dom = own[0]
    traffic = traffic_json.get(dom["kernel"], {}).get("bytes") if traffic_json else None
    if "mfma_frac" in dom:
        if "executed_flops_per_step" in dom:
            # the first layer is hoisted in front of the gather (per-point GEMM in another kernel): price the kernel on the
            # flops it EXECUTES; the reference formulation's figure is listed next to it
            ex_tf = round(dom["executed_flops_per_step"] / (dom["ms_per_step"] * 1e-3) / 1e12, 2)
            return {"bound": "mfma", "kernel": dom["kernel"], "achieved": ex_tf, "peak": MFMA_F32_PEAK_TF, "unit": "TFLOP/s",
                    "frac": dom["executed_mfma_frac"], "traffic": traffic,
                    "basis": "flops the kernel executes (layers 2..L of the set-abstraction MLP on every (centre, sample) row) / "
                             "HIP-event time around the entry point on its launch stream inside the timed region (includes "
                             "launch gaps)",
                    "algorithmic": {"tflops": dom["achieved_tflops"], "frac_of_peak": dom["mfma_frac"],
                                    "basis": "SURVEY.md §8(d): 2*rows*sum(c_in*c_out) of the WHOLE MLP (831 GFLOP for RCNN SA1 per "
                                             "1024 RoIs) / the same time; exceeds the executed figure because the hoisted first "
                                             "layer's per-row work is never done (it can exceed the peak for that reason)"}}
        return {"bound": "mfma", "kernel": dom["kernel"], "achieved": dom["achieved_tflops"], "peak": MFMA_F32_PEAK_TF,
                "unit": "TFLOP/s", "frac": dom["mfma_frac"], "traffic": traffic,
                "basis": "algorithmic flops per SURVEY.md §8(d); time = HIP events around the entry point on its launch stream "
                         "inside the timed region (includes launch gaps)"}
    r = {"bound": "hbm", "kernel": dom["kernel"], "achieved": dom["achieved_gbs"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
         "frac": dom["hbm_frac"], "traffic": traffic, "measured_copy_ceiling_gbs": HBM_COPY_CEILING_GBS,
         "basis": "algorithmic bytes per SURVEY.md §8(d) (FPS: streaming-equivalent B*m*20n — latency/VALU-bound, "
                  "compulsory bytes listed alongside); time = HIP events on the launch stream inside the timed region"}
    if "us_per_fps_iteration" in dom:
        r["us_per_fps_iteration"] = dom["us_per_fps_iteration"]
    return r


def own_rows(kernels, hide_fps):
    """the jm entries of the SURVEY.md §8 path (no stalls, no caller-side spans, no image-branch kernels); `hide_fps`: the FPS
    chain runs on its side stream and the consumer hardly ever waits for it, so it is not part of the main chain"""
    own = [k for k in kernels if not k.get("stall") and ("algo_bytes_per_step" in k or "algo_flops_per_step" in k)
           and "(" not in k["kernel"] and k["kernel"].split("/")[-1] not in IMAGE_BRANCH_ENTRIES]
    if hide_fps:
        own = [k for k in own if not k["kernel"].startswith("fps_pyramid/")]
    return own


def fps_hidden(kernels):
    chain = sum(k["ms_per_step"] for k in kernels if k["kernel"].startswith("fps_pyramid/"))
    exposed = sum(k["ms_per_step"] for k in kernels if k.get("stall") and k["kernel"].startswith("fps_exposed"))
    return chain > 0 and exposed < 0.1 * chain


def roofline_by_time(kernels, ms_step):
    """the OTHER reading of "dominant kernel": the jm entry with the largest measured time per step on the chain the step waits
    for (HIP events, fully instrumented steps), with its own fraction of the roofline that bounds it.  Next to `roofline` (largest
    speed-of-light time) it shows where the time is as opposed to where the work is."""
    own = own_rows(kernels, fps_hidden(kernels))
    if not own:
        return None
    k = max(own, key=lambda r: r["ms_per_step"])
    r = {"kernel": k["kernel"], "ms_per_step": k["ms_per_step"], "launches_per_step": k["launches_per_step"],
         "share_of_step": round(k["ms_per_step"] / ms_step, 4) if ms_step else None}
    if "us_per_fps_iteration" in k:
        r.update(bound="latency (sequential arg-max chain; cloud register-resident)", us_per_fps_iteration=k["us_per_fps_iteration"],
                 evals_per_s=k.get("evals_per_s"), valu_frac=k.get("valu_frac"),
                 hbm_frac_on_compulsory_bytes=k.get("hbm_frac"))
    elif "mfma_frac" in k:
        ex = "executed_mfma_frac" in k
        tf = (k["executed_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12) if ex and k["ms_per_step"] > 0 else k.get("achieved_tflops")
        r.update(bound="mfma", achieved=round(tf, 2), peak=MFMA_F32_PEAK_TF, unit="TFLOP/s",
                 frac=k["executed_mfma_frac"] if ex else k["mfma_frac"])
    elif "evals_per_s" in k:
        r.update(bound="valu (pairwise evaluations)", evals_per_s=k["evals_per_s"], valu_frac=k.get("valu_frac"),
                 hbm_frac_on_compulsory_bytes=k.get("hbm_frac"))
    else:
        r.update(bound="hbm", achieved=k["achieved_gbs"], peak=HBM_PEAK_GBS, unit="GB/s", frac=k["hbm_frac"])
    if k.get("traffic_bytes_per_launch"):
        r["traffic"] = k["traffic_bytes_per_launch"]
    return r


def normalise_fractions(result):
    """every `*_frac` of the record is a fraction of a PEAK the kernel could reach: rows whose algorithmic figure (the dense /
    un-hoisted / direct formulation of SURVEY.md §8d) differs from what the kernel executes carry the executed fraction as
    `mfma_frac` and the formulation's equivalent rate as `dense_equivalent_tflops` — a rate, not a fraction (it exceeds the peak
    wherever work is skipped: Winograd, hoisted first layers, listed / compacted rows)"""
    for k in result.get("kernels") or []:
        if "executed_mfma_frac" in k:
            k["dense_equivalent_tflops"] = k.get("achieved_tflops")
            k["mfma_frac"] = k.pop("executed_mfma_frac")
            if k.get("ms_per_step"):
                k["achieved_tflops"] = round(k["executed_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12, 2)
    for key in ("roofline", "roofline_by_time", "image_branch_kernel"):
        r = result.get(key)
        if isinstance(r, dict):
            for sub in ("algorithmic", "direct_form"):
                if isinstance(r.get(sub), dict):
                    r[sub].pop("frac_of_peak", None)


def isolated_affinity(st, roofline, n=10):
    """the roofline kernel ALONE on the machine, same operands as in the step (HIP events on the launching stream, after the
    timed region): in the step it shares the CUs with the next batch's image pyramid and the detections' side stream, so the
    in-step fraction says how the step is packed as much as how good the kernel is"""
    from jmodt_amd.ops.affinity import pairwise_affinity_batched
    eng = st["engine"]
    with torch.no_grad():
        _, _, inter = eng(st["xyz"], st["image"], st["pts_xy"])
        B = st["xyz"].shape[0]
        feats = inter["rcnn_feat"].view(B, -1, inter["rcnn_feat"].shape[1])
        prev = torch.roll(feats, 1, 0)
        link = eng.rcnn_net.link_layer
        keep = prof.enabled
        prof.enabled = False
        try:
            for _ in range(2):
                pairwise_affinity_batched(prev, feats, link, None)
            torch.cuda.synchronize()
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
            ev[0].record()
            for i in range(n):
                pairwise_affinity_batched(prev, feats, link, None)        # the link head = the entry the roofline row times
                ev[i + 1].record()
            torch.cuda.synchronize()
        finally:
            prof.enabled = keep
    ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))
    avg = sum(ms) / n
    flops = roofline["achieved"] * 1e12 * roofline.get("avg_launch_ms", 0) * 1e-3 if roofline.get("avg_launch_ms") else None
    if not flops:
        return None
    return {"avg_launch_ms": round(avg, 4), "min_launch_ms": round(ms[0], 4), "achieved": round(flops / (avg * 1e-3) / 1e12, 2),
            "frac": round(flops / (avg * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4),
            "note": f"{n} back-to-back calls of the same entry (jm_affinity_forward_batched: both GEMMs, projection sum, dual softmax) with "
                    "nothing else on the GPU"}


def fps_summary(kernels, ms_step, in_flight=1):
    """the FPS chain as a top-level figure: its largest level's time per iteration and the chain's share of the step
    (in_flight chains of consecutive batches run side by side: a share above 1 then means `in_flight` overlapping chains, each
    longer than a step)"""
    rows = [k for k in kernels if "us_per_fps_iteration" in k and k["kernel"].startswith("fps_pyramid/")]
    if not rows:
        return None
    big = max(rows, key=lambda r: r["ms_per_step"])
    chain = sum(k["ms_per_step"] for k in kernels if k["kernel"].startswith("fps_pyramid/"))
    exposed = sum(k["ms_per_step"] for k in kernels if k.get("stall") and k["kernel"].startswith("fps_exposed"))
    return {"kernel": big["kernel"], "us_per_fps_iteration": big["us_per_fps_iteration"], "ms_per_step": big["ms_per_step"],
            "chain_ms_per_step": round(chain, 4), "chain_share_of_step": round(chain / ms_step, 4) if ms_step else None,
            "exposed_ms_per_step": round(exposed, 4), "hidden_by_prefetch": fps_hidden(kernels), "chains_in_flight": in_flight}


COMPACT_LIMIT = 4096     # bytes: the driver records the TAIL of stdout; round 3's 24 KB line could not be parsed from it


def _short(v, n):
    return v if not isinstance(v, str) or len(v) <= n else v[:n - 1] + "\u2026"


def compact_line(full, full_path):
    """the ONE stdout line: the contract's keys + roofline / roofline_by_time / cpu_baseline / the per-cloud values, at most
    COMPACT_LIMIT bytes.  Everything else (kernel table, variants' notes, per-stage parity against the CPU chain) is in the
    full record at `full_record`."""
    def sub(d, keys, n=160):
        return None if d is None else {k: _short(d[k], n) for k in keys if k in d and d[k] is not None}
    c = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "host_enqueue_ms_per_step",
                              "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    c["config"] = sub(full["config"], ("workload", "frames_per_gpu_per_step", "points", "parallelism", "process_groups", "rank_binding"), 400)
    rf = full.get("roofline")
    if rf is not None:
        r = sub(rf, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "max_launch_ms",
                     "us_per_fps_iteration", "measured_copy_ceiling_gbs"))
        r.setdefault("traffic", None)
        if "avg_launch_ms" in r:
            r["avg_launch_ms"], r["max_launch_ms"] = round(r["avg_launch_ms"], 5), round(r["max_launch_ms"], 5)
        if rf.get("rocprof"):
            r["rocprof"] = sub(rf["rocprof"], ("avg_us", "frac", "source"), 80)
        if rf.get("isolated"):
            r["isolated"] = sub(rf["isolated"], ("avg_launch_ms", "frac"))
        c["roofline"] = r
    else:
        c["roofline"] = None
    c["roofline_by_time"] = full.get("roofline_by_time")
    if full.get("fps") is not None:
        c["fps"] = full["fps"]
    cb = full.get("cpu_baseline")
    if cb is not None:
        c["cpu_baseline"] = sub(cb, ("value", "unit", "cores", "kind", "sample", "configs0_affinity_64x64_pytorch_cpu_ms"), 330)
    cl = full.get("clouds")
    if isinstance(cl, dict) and "uniform" in cl:
        c["clouds"] = {k: v.get("value") for k, v in cl.items() if isinstance(v, dict)}
        if cl["uniform"].get("value_dense_rcnn_kernels") is not None:
            c["clouds"]["uniform_dense_rcnn_kernels"] = cl["uniform"]["value_dense_rcnn_kernels"]
    for k in ("value_streaming", "no_prefetch_value", "prefetch_depth_values", "no_image_prefetch_value", "no_overlap_value", "step_mfma_frac",
              "dropped_fractions"):
        if full.get(k) is not None:
            c[k] = full[k]
    if full.get("overlap"):
        c["overlap"] = sub(full["overlap"], ("side_streams", "next_batch_fps_prefetch", "fps_pyramids_ahead", "next_batch_image_prefetch", "fps_chain_ms", "fps_exposed_ms",
                                             "image_branch_exposed_ms"))
    if full.get("grad_allreduce"):
        c["grad_allreduce"] = sub(full["grad_allreduce"], ("world", "issued", "bytes_per_step", "ms_per_step", "mode"))
    if full.get("image_branch_kernel"):
        c["image_branch_kernel"] = sub(full["image_branch_kernel"], ("kernel", "achieved", "frac", "ms_per_step"))
    c["full_record"] = full_path
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    for drop in ("image_branch_kernel", "overlap", "fps", "step_mfma_frac", "no_overlap_value"):   # never reached at today's sizes
        if len(line.encode()) <= COMPACT_LIMIT:
            break
        c.pop(drop, None)
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    assert len(line.encode()) <= COMPACT_LIMIT, len(line)
    return line


# bench row -> kernel-name needles in the committed rocprofv3 per-shape table (profiles/<round>_detect_kernel_stats.txt): the
# kernels one C-ABI entry launches
ROCPROF_NEEDLE = {"rcnn_sa1/sa_mlp_pm_forward": ["sa_mlp_pm_kernel"],
                  # (side-stream rows whose HIP-event time is mostly queueing behind the image branch's persistent workgroups: the
                  # kernels' own durations from the committed trace go next to it, `rocprof_kernel_us`)
                  "detections/nms_batched": ["nms_mask_lane_kernel<false>", "nms_reduce_kernel"],
                  "proposal_layer/argsort_desc_stable": ["argsort_desc_kernel"],
                  "roipool3d_canonical_cnt": ["roipool3d_kernel<true, true>"],
                  "affinity_8x128x128/affinity_forward_batched": ["affinity_fused_kernel", "af_pack_kernel",
                                                                  "softmax_stats_kernel", "dual_softmax_kernel"]}


PROFILE_ROUNDS = ("r06", "r05", "r04")      # this round's committed summaries, else the previous round's OF THE SAME WORKLOAD


def rocprof_average(kernel_row, workload="detect", live_us=None, launches=None):
    """sum over the entry's kernels of the average duration of each kernel's LARGEST shape in the committed rocprofv3
    --kernel-trace --stats summary of this same command (bench.py itself runs un-profiled): the cross-check of the live
    HIP-event time.  Keyed on (round, WORKLOAD, kernel, grid x workgroup): only the summary of the workload being run is read
    (round 4's `ops` line matched a 445 us compacted dispatch of the detect profile to its 5.1 ms dense entry and printed a
    fraction of 7.85), the matched shape is returned, and a match whose duration is not within 3x of the live HIP-event time of
    the same entry is not the same dispatch and is dropped."""
    needles = ROCPROF_NEEDLE.get(kernel_row)
    if not needles:
        return None
    wl = {"detect": "detect", "sa": "sa", "ops": "ops"}.get(workload)
    if wl is None:
        return None
    for rnd in PROFILE_ROUNDS:
        path = os.path.join(ROOT, "profiles", f"{rnd}_{wl}_kernel_stats.txt")
        try:
            lines = open(path).read().splitlines()
        except OSError:
            continue
        start = next((i for i, ln in enumerate(lines) if ln.startswith("per dispatch shape")), None)
        if start is None:
            continue                      # (round 2's summaries have one row per kernel NAME: several shapes mixed)
        total, found, shapes = 0.0, [], []
        for needle in needles:
            best = None
            for ln in lines[start + 2:]:
                if needle in ln:
                    f = ln.split()
                    try:
                        avg = float(f[-3])
                    except (ValueError, IndexError):
                        continue
                    if best is None or avg > best[0]:
                        best = (avg, f"{f[-7]} x {f[-6]}")
            if best is not None:
                total += best[0]
                found.append(needle)
                shapes.append(best[1])
        if not found:
            continue
        if live_us is not None and live_us > 0 and not (live_us / 3.0 <= total <= live_us * 3.0):
            return None                   # another dispatch of the same kernel name: no cross-check rather than a wrong one
        return {"avg_us": round(total, 2), "kernels": found, "shapes": shapes, "round": rnd, "workload": wl,
                "source": os.path.relpath(path, ROOT) + " (per-shape table)"}
    return None


def sanitise_fractions(node, path="", dropped=None):
    """every `frac` / `*_frac` of the record must be a fraction: 0 < f <= 1.  Offenders are REMOVED and listed (a fraction above 1 is
    a bookkeeping error — bytes or flops of one dispatch over the time of another — never evidence), so that no line the driver
    records can carry one; returns the list of (path, value) removed"""
    dropped = [] if dropped is None else dropped
    if isinstance(node, dict):
        for k in list(node.keys()):
            v = node[k]
            if (k == "frac" or k.endswith("_frac")) and isinstance(v, (int, float)) and not isinstance(v, bool):
                if not (0.0 < float(v) <= 1.0):
                    dropped.append((f"{path}/{k}", v))
                    del node[k]
            else:
                sanitise_fractions(v, f"{path}/{k}", dropped)
    elif isinstance(node, list):
        for i, v in enumerate(node):
            sanitise_fractions(v, f"{path}[{i}]", dropped)
    return dropped


def pin_rank(local_rank: int, ranks_on_node: int, sysfs: str = "/sys", bus_ids=None):
    """one process per GPU: bind this rank to its own block of host cores and bound its intra-op thread pool (the reference's
    nn.DataParallel runs ONE process, tools/train.py:86-88; eight processes that each enqueue 4 - 20 ms of launches per step on
    cores they share with the others' 128-thread pools do not).  The block = whole physical cores of the NUMA node this rank's GPU
    hangs off (jmodt_amd/hostbind.py: the GPU's PCI address from the HIP runtime, its numa_node and the node's cpulist from sysfs),
    shared evenly with the other local ranks on that node; where the platform does not tell (numa_node = -1, no sysfs nodes in the
    container) the allowed cores are split evenly in rank order.  JM_BENCH_NO_PIN=1 switches it off.  Returns what was done (goes
    into the line)."""
    if os.environ.get("JM_BENCH_NO_PIN") == "1" or not hasattr(os, "sched_setaffinity"):
        return {"pinned": False}
    try:
        from jmodt_amd import hostbind
        got = hostbind.rank_cores(local_rank, ranks_on_node, sorted(os.sched_getaffinity(0)), sysfs, bus_ids)
        mine = sorted(got["cores"])
        os.sched_setaffinity(0, mine)
        threads = max(1, min(len(mine), 16))
        torch.set_num_threads(threads)
        return {"pinned": True, "cores": hostbind_ranges(mine), "n_cores": len(mine), "torch_threads": threads,
                "source": got["source"], "numa_node": got["numa_node"]}
    except OSError as e:                   # (a container that forbids it: run unpinned, say so)
        return {"pinned": False, "error": str(e)}


def hostbind_ranges(cores):
    """[0, 1, 2, 3, 64, 65] -> "0-3,64-65" """
    out, i = [], 0
    while i < len(cores):
        j = i
        while j + 1 < len(cores) and cores[j + 1] == cores[j] + 1:
            j += 1
        out.append(str(cores[i]) if i == j else f"{cores[i]}-{cores[j]}")
        i = j + 1
    return ",".join(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None, help="frames per GPU per step (default 8; train: 4)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="FPS chain and image branch on the main stream")
    ap.add_argument("--image-prefetch", choices=["early", "late", "off"], default="early",
                    help="next batch's image pyramid: under this batch's backbone (default since round 4: with the lighter main chain "
                         "the image stream is the critical path and this fills the step's tail, +4 %%), after it, or not announced "
                         "(`no_image_prefetch_value` in the line)")
    ap.add_argument("--no-prefetch", action="store_true", help="do not start the next batch's FPS pyramid early")
    ap.add_argument("--no-ahead", action="store_true",
                    help="train --rcnn only: do not issue the next step's frozen half (RPN forward, proposals, RoI pooling) under this step's RCNN")
    ap.add_argument("--prefetch-depth", type=int, default=None,
                    help="FPS pyramids of this many upcoming batches in flight, each serial chain on a side stream of its own (one CU per "
                         "frame).  Default 2 for `sa` (configs[1] IS a sampling chain: 1397 / 2754 / 3920 frames/s at 0 / 1 / 2), 1 for the "
                         "engine workloads (work-bound: a second chain in flight changes nothing at 8 frames per step and costs 2-4 %% at "
                         "4, tools/prefetch_depth_probe.py)")
    ap.add_argument("--tiny", action="store_true", help="smoke-sized shapes (tests only; the JSON says so)")
    ap.add_argument("--workload", default="detect", choices=[w for w in WORKLOAD_TEXT if not w.startswith("train_")])
    ap.add_argument("--cloud", default="uniform", choices=["uniform", "kitti", "packed"],
                    help="detect / train: the synthetic cloud the whole line (value, kernel table) is measured on; the default detect line "
                         "and the --joint / --rcnn train lines always carry all three values under `clouds`")
    ap.add_argument("--joint", action="store_true",
                    help="train only: joint mode — forward / backward through the WHOLE detector (row kernels; JM_JOINT_ROUTE=operators: the "
                         "un-fused autograd route) and the bucketed all-reduce of all 16.7 M parameters (66.9 MB), instead of the finetune "
                         "step of the two heads (4.2 MB)")
    ap.add_argument("--rcnn", action="store_true",
                    help="train only: the reference's default mode (config.py:57 RPN.FIXED) — frozen fused RPN forward, RCNN + re-id heads "
                         "forward / backward on the row kernels, all-reduce of their gradients only")
    ap.add_argument("--no-listed", action="store_true",
                    help="RPN set-abstraction scales on the dense kernels (every back-filled row executed) instead of the listed form")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip everything that runs OTHER shapes after the timed region (the kitti / packed clouds, the dense-RCNN and "
                         "experimental variants): what the rocprofv3 passes use, so that every dispatch of a profile belongs to the "
                         "headline workload (profiles/make_traffic.py attributes counters per (kernel, grid, workgroup))")
    ap.add_argument("--full-out", default=None,
                    help="where the FULL record (kernel table, variants, parity block) is written; default bench_out/<workload>.json. "
                         "stdout carries one compact line of at most 4 KB")
    ap.add_argument("--stream-inputs", action="store_true",
                    help="detect: also report `value_streaming` — every step a fresh batch from pinned host memory, async H2D under the "
                         "previous step (measured after the timed region; on by default for the default workload)")
    ap.add_argument("--launch", action="store_true",
                    help="re-execute under torch.distributed.run even for --gpus 1 (the N > 1 launch path incl. RCCL init / "
                         "barrier / all-reduce on one GPU: what the GPU tier runs)")
    args = ap.parse_args()
    if (args.joint or args.rcnn) and args.workload != "train" or (args.joint and args.rcnn):
        raise SystemExit("bench.py: --joint / --rcnn are modes of --workload train (one of them)")
    if args.batch is None:
        args.batch = 4 if args.workload == "train" else 8

    if args.workload == "train":
        # HIP multiplexes streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  The RCCL communicator of the training
        # step takes some of them, and the engine's four streams then serialise: 381 frames/s per GPU instead of 497-505
        # (measured on one GPU, communicator alive; 6 or more queues restore it).  Read by the HIP runtime at initialisation.
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    if "WORLD_SIZE" not in os.environ and (args.gpus > 1 or args.launch):
        sys.exit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} inside a torch.distributed.run job of WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the jmodt ops have no CPU fallback")
    shared = os.environ.get("JM_BENCH_SHARE_GPU") == "1" and torch.cuda.device_count() >= 1
    if shared:
        # TEST HOOK (tests/test_gpu_detector.py): several ranks on ONE GPU, so that a one-GPU box can execute the N > 1 control plane of
        # the inference workloads (rendezvous, gloo barriers, MAX over ranks, rank-0 output, per-rank MIOpen paths).  Never a
        # benchmark: the line says so, and the RCCL workloads refuse it (two ranks cannot share a device in one communicator)
        if args.workload == "train":
            raise SystemExit("bench.py: JM_BENCH_SHARE_GPU is for the replica workloads only (RCCL needs one GPU per rank)")
        device_index = local_rank % torch.cuda.device_count()
    else:
        if torch.cuda.device_count() <= local_rank:
            raise SystemExit(f"bench.py: rank {rank} wants cuda:{local_rank}, the node has {torch.cuda.device_count()} GPU(s)")
        device_index = local_rank
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    rank_binding = {"pinned": False}
    if world > 1 or "RANK" in os.environ:          # one process per GPU under torch.distributed.run: its own block of host cores
        rank_binding = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    if world > 1 or "RANK" in os.environ or os.environ.get("JM_BENCH_FORCE_DIST") == "1":   # (--launch takes this path on one GPU)
        # one process per GPU: every rank runs MIOpen's find step for the image convolutions at start-up; give each its own user
        # database so that eight ranks do not serialise on (or trip over) the file locks of a shared one.  Read at MIOpen's first use
        for var, sub in (("MIOPEN_USER_DB_PATH", "db"), ("MIOPEN_CUSTOM_CACHE_DIR", "cache")):
            if var not in os.environ:
                path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"jm_miopen_{os.getuid()}", f"rank{local_rank}", sub)
                os.makedirs(path, exist_ok=True)
                os.environ[var] = path
    dist = None
    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ and "MASTER_ADDR" in os.environ      # under torch.distributed.run
    if world > 1 or launched or os.environ.get("JM_BENCH_FORCE_DIST") == "1":   # (a one-rank launch takes the collective path too)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:      # JM_BENCH_FORCE_DIST=1 from a plain shell: a world of one
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(sk.getsockname()[1]), RANK="0", WORLD_SIZE="1")
        # Two process groups.  CONTROL plane (the barriers around the timed region, the MAX of the elapsed times): gloo — it
        # needs no GPU resources.  DATA plane: RCCL, created ONLY for the workload that has a collective on its path (train: the
        # gradient all-reduce); the inference workloads are replicas without any exchange, and a live RCCL
        # communicator costs them throughput for nothing: with it, the same single-GPU step measures 572 instead of 609
        # frames/s (detect) and 380 instead of 505 (train) — per-kernel times unchanged, more gaps between the launches of the
        # engine's five streams (measured with JM_BENCH_FORCE_DIST=1; GPU_MAX_HW_QUEUES and the NCCL_* / RCCL_* channel knobs do
        # not bring it back).  The training step pays that price because it needs the collective.
        # stdout carries exactly one JSON line.  RCCL prints a version banner to fd 1 when the communicator is
        # created (NCCL_DEBUG=VERSION in this image), so fd 1 points at stderr while that happens.
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.workload == "train":
                dist.init_process_group("nccl", device_id=dev)
                ctl = dist.new_group(backend="gloo")
                dist.barrier()                   # creates the communicator (and prints the banner) now
                torch.cuda.synchronize()
            else:
                dist.init_process_group("gloo")
                ctl = dist.group.WORLD
            dist.barrier(group=ctl)
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from jmodt_amd import _lib
    _lib.load()
    if args.no_listed:
        from jmodt_amd.ops.pointnet2 import fused as _fused
        _fused.LISTED = False

    seed = 1234 + rank
    if args.prefetch_depth is None:
        args.prefetch_depth = 2 if args.workload == "sa" else 1
    if args.workload == "detect":
        st = make_detect_state(args.batch, seed + 2, dev, tiny=args.tiny, kind=args.cloud)
        st["engine"].overlap = not args.no_overlap
        st["engine"].prefetch_depth = args.prefetch_depth
        st["prefetch"] = not args.no_prefetch
        st["engine"].prefetch_image = args.image_prefetch != "off"
        st["engine"].prefetch_image_late = args.image_prefetch == "late"
        step = lambda: detect_step(st)  # noqa: E731
    elif args.workload == "sa":
        xyz, feats = make_sa_inputs(args.batch, seed + 1, dev)
        sa_ahead = {"depth": 0 if args.no_prefetch else args.prefetch_depth, "fifo": [], "n": 0}
        step = lambda: sa_step(xyz, feats, overlap=not args.no_overlap, ahead=sa_ahead)  # noqa: E731
    elif args.workload == "ops":
        ops_in = make_ops_inputs(args.batch, seed + 2, dev, small=args.tiny)
        step = lambda: ops_step(ops_in)  # noqa: E731
    elif args.workload == "dense_detect":
        st = make_detect_state(args.batch, seed + 4, dev, tiny=args.tiny, points=65536, rois=256)
        st["engine"].overlap = not args.no_overlap
        st["engine"].prefetch_depth = args.prefetch_depth
        st["prefetch"] = not args.no_prefetch
        st["engine"].prefetch_image = args.image_prefetch != "off"
        st["engine"].prefetch_image_late = args.image_prefetch == "late"
        step = lambda: detect_step(st)  # noqa: E731
    elif args.workload == "dense":
        dense_in = make_dense_inputs(args.batch, seed + 4, dev, small=args.tiny)
        step = lambda: dense_step(dense_in)  # noqa: E731
    else:
        train_st = (make_joint_state if args.joint else make_rcnn_state if args.rcnn else make_train_state)(
            args.batch, seed + 3, dev, tiny=args.tiny, kind=args.cloud)
        train_st["engine"].overlap = not args.no_overlap
        train_st["engine"].prefetch_depth = args.prefetch_depth
        train_st["prefetch"] = not args.no_prefetch
        train_st["ahead"] = not args.no_ahead
        train_st["engine"].prefetch_image = args.image_prefetch != "off"
        train_st["engine"].prefetch_image_late = args.image_prefetch == "late"
        step = lambda: train_step(train_st, world)  # noqa: E731
    # Per-entry HIP events cost GPU time (two records per call: ~0.9 ms of the 21 ms composed step), so the timed region
    # carries events around ONE entry only — the dominant jm kernel, picked from a fully instrumented warm-up step — and the
    # per-entry table comes from fully instrumented steps AFTER the timed region.
    # one untimed priming pass, whatever --warmup says: the first call of every kernel loads its code object, packs / folds the
    # weights, and lets MIOpen pick the image convolutions' kernels (seconds per new shape, once per process) — set-up, not a step
    step()
    torch.cuda.synchronize()
    for i in range(args.warmup):
        last = i == args.warmup - 1
        if last:
            torch.cuda.synchronize()
            prof.reset()
            prof.only = None
            prof.enabled = True
        step()
        if last:
            torch.cuda.synchronize()
            prof.enabled = False
    dom_key = None
    if args.warmup >= 1:
        dom = pick_roofline(prof.summary(1, HBM_PEAK_GBS, MFMA_F32_PEAK_TF), None)
        dom_key = dom["kernel"] if dom else None
    torch.cuda.synchronize()
    gc.collect()
    gc.disable()   # a generation-2 collection in the middle of the timed region is a 30-40 ms host stall
    if dist is not None:
        dist.barrier(group=ctl)
    prof.reset()
    prof.only = {dom_key} if dom_key else None
    prof.enabled = True
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_enqueue = time.perf_counter() - t0      # host time to ENQUEUE the steps (the GPU runs behind it)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof.enabled = False
    if dist is not None:
        dist.barrier(group=ctl)
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=ctl)
        elapsed = float(t.item())
    timed_rows = prof.summary(args.steps, HBM_PEAK_GBS, MFMA_F32_PEAK_TF)
    table_steps = args.steps
    bq_evals = {}
    nms_evals = (0, 0, 0)
    if dom_key:                                   # the per-entry table: fully instrumented steps, outside `value`
        from jmodt_amd.ops.pointnet2 import pointnet2_utils as _pu
        table_steps = max(1, min(5, args.steps))
        prof.reset()
        prof.only = None
        prof.enabled = True
        _pu.BQ_EVALS.clear()
        from jmodt_amd.ops import proposal as _pp
        _pp.NMS_EVALS.clear()
        t1 = time.perf_counter()
        for _ in range(table_steps):
            step()
        torch.cuda.synchronize()
        table_ms = (time.perf_counter() - t1) / table_steps * 1e3
        prof.enabled = False
        bq_evals = _pu.ball_query_evals()          # distance evaluations the grid searches really did (device counters)
        nms_evals = _pp.nms_evals()                # IoU evaluations of the RPN's lazy first-K NMS vs its full pair masks
        if dist is not None:
            dist.barrier(group=ctl)

    # the same workload with one overlap mechanism off at a time (a few steps, after the timed region, outside `value`)
    variants = {}
    if args.workload == "sa" and args.steps >= 2 and not args.headline_only and sa_ahead["depth"] > 0 and not args.no_overlap:
        # the same workload with nothing announced early: every batch's FPS pyramid starts at the head of its own step
        keep_depth, n_var = sa_ahead["depth"], max(2, min(10, args.steps))
        rates = {}
        for d in sorted({0, 1, keep_depth}):
            if d == keep_depth:
                continue
            for pyr in sa_ahead["fifo"]:
                pyr.release()
            sa_ahead.update(depth=d, fifo=[])
            step(); step()
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(n_var):
                step()
            torch.cuda.synchronize()
            rates[d] = round(world * args.batch * n_var / (time.perf_counter() - t2), 2)
        for pyr in sa_ahead["fifo"]:
            pyr.release()
        sa_ahead.update(depth=keep_depth, fifo=[])
        variants["no_prefetch_value"] = rates.get(0)
        variants["prefetch_depth_values"] = {str(d): v for d, v in rates.items()}
        variants["variants_note"] = (f"{n_var} steps each after the timed region: the same step with the FPS pyramids of 0 / 1 upcoming batches "
                                     f"in flight instead of {keep_depth} (0 = every batch's chain starts at the head of its own step)")
    if args.workload == "detect" and args.steps >= 2 and not args.headline_only:
        eng = st["engine"]
        n_var = max(2, min(10, args.steps))        # (5 steps = a 60 ms window: the per-cloud values scattered by 8 % between runs)

        def variant(prefetch, overlap):
            keep = (st["prefetch"], eng.overlap)
            st["prefetch"], eng.overlap = prefetch, overlap
            try:
                step(); step()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(n_var):
                    step()
                torch.cuda.synchronize()
                return world * args.batch * n_var / (time.perf_counter() - t2)
            finally:
                st["prefetch"], eng.overlap = keep
                step()                                  # consume / re-announce under the restored settings
                torch.cuda.synchronize()
        if st["prefetch"] and eng.overlap and eng.prefetch_image:
            eng.prefetch_image = False
            try:
                variants["no_image_prefetch_value"] = round(variant(True, True), 2)
            finally:
                eng.prefetch_image = True
        if st["prefetch"] and eng.overlap:
            variants["no_prefetch_value"] = round(variant(False, True), 2)
        if eng.overlap:
            variants["no_overlap_value"] = round(variant(False, False), 2)
        try:
            variants["value_streaming"] = round(streaming_rate(st, n_var, world, args.batch), 2)
            variants["streaming_note"] = ("a NEW batch every step from pinned host memory, H2D on a copy stream under the previous step, "
                                          "two alternating device buffer sets; the next batch's IMAGE pyramid is not announced (its "
                                          "pixels are still in flight when this step starts)")
        except Exception as ex:        # a report next to the headline, never a reason to lose it
            variants["value_streaming"] = None
            variants["streaming_note"] = f"failed: {ex!r}"
        step(); step()
        torch.cuda.synchronize()
        variants["clouds"] = None
        variants["variants_note"] = (f"{n_var} steps each after the timed region, this rank x world: no_image_prefetch = the next batch's image pyramid "
                                     "is not started under this batch (its FPS pyramid still is); no_prefetch = every batch's FPS pyramid "
                                     "starts at the head of its OWN step (still on the side stream, nothing announced early); no_overlap = "
                                     "FPS chain, image branch and detection glue all on the main stream")
        # the RCNN stage skips (centre, sample) rows that are exact copies (csrc/sa_dedupe.hip); how many there are depends
        # on how many points the RoIs hold: the same network on the three synthetic clouds, rows executed next to rows dense
        torch.cuda.synchronize()
        clouds = {"uniform": {"value": None, "rcnn": rcnn_rows(), "note": "the headline workload (SURVEY.md §8d: the KITTI crop filled uniformly + 10 % duplicates)"}}
        if eng.dedupe_rcnn:
            eng.dedupe_rcnn = False
            try:
                clouds["uniform"]["value_dense_rcnn_kernels"] = round(variant(st["prefetch"], eng.overlap), 2)
            finally:
                eng.dedupe_rcnn = True
        if not args.tiny:
            for kind, note in (("kitti", "density ~ 1/z, ground plane + object clusters (synth.kitti_like_cloud)"),
                               ("packed", "worst case: every point inside one of 16 car-sized boxes, RoIs hold >= 512 distinct points")):
                keep = {k: st[k] for k in ("xyz", "image", "pts_xy")}
                st.update(detect_inputs(args.batch, seed + 2, dev, False, kind))
                try:
                    v = variant(st["prefetch"], eng.overlap)
                    torch.cuda.synchronize()
                    clouds[kind] = {"value": round(v, 2), "rcnn": rcnn_rows(), "note": note}
                finally:
                    st.update(keep)
                    step(); step()
                    torch.cuda.synchronize()
        variants["clouds"] = clouds
        # EXPERIMENTAL, never the headline: the link head's fp32 products as 3-term bf16 splits on the bf16 matrix pipe
        # (csrc/affinity_x3.hip; error vs float64 at the level of the exact-fp32 kernel's: tests/test_gpu_surface.py)
        eng.affinity_split_bf16 = True
        try:
            variants["experimental"] = {"split_bf16_affinity_value": round(variant(st["prefetch"], eng.overlap), 2),
                                        "note": "same workload with ONLY the batched link head on jm_affinity_link_scores_x3 (six bf16 MFMA "
                                                "products per fp32 product, fp32 accumulate); `value` and every other number of this "
                                                "line are from the exact-fp32 kernels"}
        finally:
            eng.affinity_split_bf16 = False
            step(); step()
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier(group=ctl)

    if args.workload == "train" and (args.joint or args.rcnn) and args.steps >= 2 and not args.headline_only and not args.tiny and args.cloud == "uniform":
        # the same step on the other synthetic clouds (the uniform one leaves ~1 distinct row per RPN group; these have rows): a few
        # steps each after the timed region, this rank x world
        n_var = max(2, min(10, args.steps))
        clouds = {"uniform": {"value": None, "note": "the headline workload"}}
        for kind in ("kitti", "packed"):
            keep = {k: train_st[k] for k in ("xyz", "image", "pts_xy")}
            train_st.update(detect_inputs(args.batch, seed + 3, dev, False, kind))
            try:
                step(); step()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                for _ in range(n_var):
                    step()
                torch.cuda.synchronize()
                clouds[kind] = {"value": round(world * args.batch * n_var / (time.perf_counter() - t2), 2), "note": f"{n_var} steps after the timed region"}
            finally:
                train_st.update(keep)
                step(); step()
                torch.cuda.synchronize()
        variants["clouds"] = clouds
        if dist is not None:
            dist.barrier(group=ctl)
    if args.workload == "detect" and args.headline_only:
        torch.cuda.synchronize()
        variants["clouds"] = {"uniform": {"value": None, "rcnn": rcnn_rows(), "note": "--headline-only: the other clouds were not run"}}
    if rank == 0:
        kernels = prof.summary(table_steps, HBM_PEAK_GBS, MFMA_F32_PEAK_TF)
        # HBM bytes per launch from the committed rocprofv3 --pmc passes (collected separately, as the guide
        # prescribes; bench.py itself runs un-profiled)
        tj = None
        for rnd in ("r04",):          # (older files attributed by kernel NAME over mixed workloads: not used any more)
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", f"{rnd}_traffic.json")))
                break
            except (OSError, ValueError):
                continue
        if tj:
            for k in kernels:
                if k["kernel"] in tj and k["ms_per_step"] > 0:
                    per_launch_s = k["ms_per_step"] / max(k["launches_per_step"], 1) * 1e-3
                    if tj[k["kernel"]]["bytes"] / per_launch_s <= HBM_PEAK_GBS * 1e9:     # (a rate above the peak = not this dispatch)
                        k["traffic_bytes_per_launch"] = tj[k["kernel"]]["bytes"]
                        k["traffic_gbs"] = round(tj[k["kernel"]]["bytes"] / per_launch_s / 1e9, 1)
        ms_step = elapsed / args.steps * 1e3
        for k in kernels:
            if k["kernel"] in ROCPROF_NEEDLE and not k["kernel"].startswith("affinity"):
                rp = rocprof_average(k["kernel"], args.workload)
                if rp:
                    k["rocprof_kernel_us"] = rp["avg_us"]
        # the hash-grid ball queries: evaluations actually done (per step) next to the n * m of the scan they replace
        for k in kernels:
            scope = k["kernel"].rsplit("/", 1)[0] + "/ball_query" if "/" in k["kernel"] else "ball_query"
            if k["kernel"].endswith("proposal_select") and nms_evals[2]:
                k["iou_evals_per_step"] = int(nms_evals[0] / table_steps)
                k["pair_mask_evals_per_step"] = int(nms_evals[1] / table_steps)
                k["evals_vs_pair_mask"] = round(nms_evals[0] / max(nms_evals[1], 1), 6)
            if "ball_query" in k["kernel"] and scope in bq_evals and k["ms_per_step"] > 0:
                ev = bq_evals[scope][0] / table_steps
                k["evals_per_step"] = int(ev)
                k["evals_per_s"] = round(ev / (k["ms_per_step"] * 1e-3), 1)
                if k.get("brute_force_evals_per_step"):
                    k["evals_vs_brute_force"] = round(ev / k["brute_force_evals_per_step"], 5)
        # the listed RPN scales: rows executed = sum over classes of groups << q, read back from the plans of the last step
        listed_now = listed_rows()
        for rows in (kernels, timed_rows):
            for k in rows:
                ln = listed_now.get(k["kernel"])
                if k.get("listed") and ln and k["ms_per_step"] > 0:
                    k["rows_executed"], k["rows_dense"] = ln["rows_executed"], ln["rows_dense"]
                    hoisted = k.get("algo_flops_per_step", 0) - k.get("executed_flops_per_step", k.get("algo_flops_per_step", 0))
                    k["executed_flops_per_step"] = int(ln["rows_executed"] * k["flops_per_row"])
                    k["algo_flops_per_step"] = int(ln["rows_dense"] * k["flops_per_row"] + hoisted)     # the dense block, SURVEY.md §8d
                    k["algo_bytes_per_step"] = int(ln["rows_executed"] * k["bytes_per_row"])
                    k["achieved_tflops"] = round(k["algo_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12, 2)
                    k["mfma_frac"] = round(k["achieved_tflops"] / MFMA_F32_PEAK_TF, 4)
                    k["executed_mfma_frac"] = round(k["executed_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4)
                    k["note"] = ("listed form: algo_flops = the dense block (SURVEY.md §8d), executed = 2^ceil(log2 d) rows per group of d "
                                 "distinct neighbours" + (", first layer hoisted" if hoisted else ""))
        # the compacted RCNN scales: executed rows were read back from the device after the timed region
        rows_now = (variants.get("clouds") or {}).get("uniform", {}).get("rcnn", {})
        for rows in (kernels, timed_rows):
            for k in rows:
                scale = k["kernel"].split("/")[0]
                if "flops_per_row" in k and scale in rows_now:
                    ex = rows_now[scale]["rows_executed"]
                    k["rows_executed"], k["rows_dense"] = ex, rows_now[scale]["rows_dense"]
                    k["executed_flops_per_step"] = k["algo_flops_per_step"] = ex * k["flops_per_row"]
                    k["algo_bytes_per_step"] = ex * k["bytes_per_row"]
                    if k["ms_per_step"] > 0:
                        tf = k["executed_flops_per_step"] / (k["ms_per_step"] * 1e-3) / 1e12
                        k["achieved_tflops"], k["mfma_frac"] = round(tf, 2), round(tf / MFMA_F32_PEAK_TF, 4)
                        k["executed_mfma_frac"] = k["mfma_frac"]
        # the roofline kernel's time comes from the TIMED region (its own events only); bytes / flops per call are the same
        roofline = pick_roofline(timed_rows, tj, full_table=False) if dom_key else pick_roofline(kernels, tj)
        if roofline is not None and dom_key:
            roofline["timing"] = (f"HIP events around this entry only, {args.steps} timed steps; the table `kernels` is from "
                                  f"{table_steps} fully instrumented steps run after the timed region ({table_ms:.2f} ms per step "
                                  f"with every entry timed vs {ms_step:.2f} ms)")
        exposed = sum(k["ms_per_step"] for k in kernels if k.get("stall") and k["kernel"].startswith("fps_exposed"))
        fps_total = sum(k["ms_per_step"] for k in kernels if k["kernel"].startswith("fps_pyramid/"))
        img_exposed = sum(k["ms_per_step"] for k in kernels if k.get("stall") and k["kernel"].startswith("image_exposed"))
        frames = world * args.batch * args.steps
        # all fp32 matrix-core work of one step (what each kernel EXECUTES: hoisted first layers excluded, MIOpen's 3x3
        # convolutions included) over the step time
        mfma_flops = sum(k.get("executed_flops_per_step", k.get("algo_flops_per_step", 0)) for k in kernels)
        if roofline is not None and roofline.get("bound") == "mfma":
            dom_row = next((k for k in timed_rows if k["kernel"] == roofline["kernel"]), None)
            if dom_row:
                roofline["avg_launch_ms"] = dom_row["ms_per_step"] / max(dom_row["launches_per_step"], 1)
                roofline["max_launch_ms"] = dom_row["max_launch_ms"]
            rp = rocprof_average(roofline["kernel"], args.workload, live_us=(roofline.get("avg_launch_ms") or 0) * 1e3)
            if rp:
                ex = next((k.get("executed_flops_per_step", k.get("algo_flops_per_step")) for k in kernels if k["kernel"] == roofline["kernel"]), None)
                if ex:
                    rp["frac"] = round(ex / (rp["avg_us"] * 1e-6) / 1e12 / MFMA_F32_PEAK_TF, 4)
                roofline["rocprof"] = rp
            if args.workload in ("detect", "dense_detect") and roofline["kernel"].startswith("affinity_") and not args.tiny:
                iso = isolated_affinity(st, roofline)
                if iso:
                    roofline["isolated"] = iso
        result = {
            "metric": METRIC,
            "value": round(frames / elapsed, 2),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_step, 4),
            "host_enqueue_ms_per_step": round(host_enqueue / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": WORKLOAD_TEXT[workload_key(args)]
                                   + (" [TINY smoke shapes: not a benchmark]" if args.tiny else "")
                                   + (f" [cloud: {args.cloud}]" if args.cloud != "uniform" else ""),
                       **train_mode_keys(args),
                       "frames_per_gpu_per_step": args.batch,
                       "points": (65536 if args.workload in ("dense", "dense_detect") else 16384) if not args.tiny else "tiny",
                       "parallelism": (f"dp{world} (gradient all-reduce)" if args.workload == "train" else f"replicas x{world}")
                                      + (" [TEST: ranks share one GPU, not a benchmark]" if shared else ""),
                       "rank_binding": rank_binding,
                       # what one-process-per-GPU set for THIS rank before HIP / MIOpen initialised (asserted by the launcher tests)
                       "rank_env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "MIOPEN_USER_DB_PATH", "MIOPEN_CUSTOM_CACHE_DIR",
                                                                   "HSA_ENABLE_IPC_MODE_LEGACY") if os.environ.get(k) is not None},
                       "process_groups": (None if dist is None else
                                          ("data plane RCCL (gradient all-reduce), control plane gloo (barriers, max over ranks)"
                                           if args.workload == "train" else
                                           "no data-path collective (replicas): control plane gloo (barriers, max over ranks), no RCCL communicator"))},
            "roofline": roofline,
            "roofline_selection": "the jm entry of the SURVEY.md §8 path with the largest speed-of-light time (executed flops / 157.3 TF, "
                                  "algorithmic bytes / 8 TB/s) of the main chain; measured times of small kernels include waits behind the "
                                  "other stream's convolutions; the image branch's own convolution kernel is priced in `image_branch_kernel`",
            "roofline_by_time": roofline_by_time(kernels, table_ms if dom_key else ms_step),
            "fps": fps_summary(kernels, table_ms if dom_key else ms_step,
                               in_flight=1 if (args.no_prefetch or args.no_overlap or args.workload not in ("sa", "detect", "train", "dense_detect"))
                               else (args.prefetch_depth + 1 if args.workload == "sa" else max(1, args.prefetch_depth))),
            "image_branch_kernel": image_branch_kernel(kernels),
            "step_mfma_frac": round(mfma_flops / (ms_step * 1e-3) / 1e12 / MFMA_F32_PEAK_TF, 4) if ms_step else None,
            "step_mfma_flops": int(mfma_flops),
            **{k: v for k, v in variants.items() if k != "clouds"},
            "clouds": (_with_headline(variants.get("clouds"), round(frames / elapsed, 2)) if args.cloud == "uniform" else
                       {"note": f"this line is measured on the `{args.cloud}` cloud throughout; the three-cloud comparison is part of the default line"}),
            "affinity_operands": ("all RoI slots of every frame (P = D = proposals per frame: fixed work per frame, SURVEY.md §8d), not the "
                                  "detection-NMS survivors: the head therefore does not wait for box decode / score filter / rotated NMS, "
                                  "which run on a side stream under its GEMMs; DetectionCache.associate (tests) is the survivor-only form"
                                  if args.workload in ("detect", "dense_detect") else None),
            "overlap": {"side_streams": not args.no_overlap, "next_batch_fps_prefetch": not (args.no_prefetch or args.no_overlap),
                        "fps_pyramids_ahead": 0 if (args.no_prefetch or args.no_overlap) else args.prefetch_depth,
                        "next_batch_image_prefetch": (args.image_prefetch if not (args.no_prefetch or args.no_overlap) and
                                                      args.workload in ("detect", "train", "dense_detect") else "off"),
                        "fps_chain_ms": round(fps_total, 4), "fps_exposed_ms": round(exposed, 4),
                        "fps_critical_path_share": round(exposed / ms_step, 4) if ms_step else None,
                        "image_branch_exposed_ms": round(img_exposed, 4),
                        "note": "exposed = time the main stream is held at its wait on the side stream (HIP events "
                                "either side of the wait); chain = sum of the FPS entry points on the side stream"},
            "grad_allreduce": ({"world": world, "bytes_per_step": next((k.get("algo_bytes_per_step") for k in kernels
                                                                         if k["kernel"].endswith("grad_allreduce(RCCL)")), None),
                                "ms_per_step": next((k["ms_per_step"] for k in kernels if k["kernel"].endswith("grad_allreduce(RCCL)")), None),
                                "issued": _grad_collectives(args.joint or args.rcnn),
                                "mode": "RCCL all_reduce(SUM) on the flat fp32 gradient bucket, issued whenever a process group exists (a "
                                        "one-rank group included)" if dist is not None else "no process group: nothing issued",
                                "note": ("joint mode: the gradient of every parameter in 64 MiB flat fp32 buckets (66.9 MB = one collective at the "
                                         "reference widths); " if args.joint else
                                         "rcnn mode: the gradient of the RCNN's and the re-id heads' parameters in one flat fp32 bucket; " if args.rcnn else "") +
                                        "one flat fp32 all-reduce of the link / start-end heads' gradients per step (RCCL; HIP events on the "
                                        "launching stream around the collective and its wait) + two 3-float / 1-float all-reduces (global "
                                        "loss-mean counts, loss); `issued` = gradient collectives of the last step"}
                               if args.workload == "train" else None),
            "kernels_from": (f"{table_steps} fully instrumented steps after the timed region" if dom_key else "the timed region"),
            "kernels": kernels,
        }
        if not args.no_cpu_baseline and world == 1 and not args.tiny and args.workload in ("detect", "sa"):
            cores = torch.get_num_threads()
            try:
                extra = {}
                if args.workload == "detect":
                    fps, dt, stages, aff64, parity = cpu_baseline_detect(2, dev)
                    extra = {"stage_seconds": stages, "configs0_affinity_64x64_pytorch_cpu_ms": round(aff64 * 1e3, 2),   # median of 10 after 3 warm-ups
                             "gpu_vs_chain": parity}
                    sample = (f"2 full-size frames (one (prev, next) pair) through the chained CPU oracle in {dt:.1f} s: "
                              "oracle C restatement for the jmodt ops (the reference has no CPU code for them), the "
                              "same PyTorch-CPU operators the reference calls for conv / BN / Linear / grid_sample / "
                              f"softmax, float32, torch threads = {cores}, os.cpu_count() = {os.cpu_count()}")
                else:
                    fps, dt = cpu_baseline_sa(args.batch)
                    sample = (f"one batch of {args.batch} frames of the same workload ({dt:.1f} s), oracle C restatement "
                              "with OpenMP over batch/centres (the reference has no CPU code for these ops)")
                    cores = os.cpu_count()
                result["cpu_baseline"] = {"value": round(fps, 3), "unit": "frames/s", "cores": cores, "kind": "port", "sample": sample,
                                          **extra}
            except Exception as ex:  # the baseline is a report, never a reason to lose the GPU number
                result["cpu_baseline"] = {"value": None, "unit": "frames/s", "cores": cores, "kind": "port", "sample": f"failed: {ex!r}"}
        normalise_fractions(result)
        dropped = sanitise_fractions(result)
        if dropped:
            result["dropped_fractions"] = [{"path": p_, "value": v_} for p_, v_ in dropped]
        # stdout carries ONE compact line (<= COMPACT_LIMIT bytes); the full record (kernel table, variants, per-stage parity
        # against the CPU chain) goes to a file next to it
        full_path = args.full_out or os.path.join("bench_out", args.workload + ("_joint" if args.joint else "_rcnn" if args.rcnn else "") + ("" if args.cloud == "uniform" else "_" + args.cloud)
                                                  + ("_tiny" if args.tiny else "") + ".json")
        abs_path = full_path if os.path.isabs(full_path) else os.path.join(ROOT, full_path)
        try:
            os.makedirs(os.path.dirname(abs_path), exist_ok=True)
            with open(abs_path, "w") as fh:
                json.dump(result, fh, allow_nan=False)
                fh.write("\n")
        except OSError as ex:        # a read-only tree must not cost the line
            full_path = f"not written: {ex!r}"
        sys.stdout.write(compact_line(result, full_path) + "\n")
        sys.stdout.flush()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
