"""Seeded synthetic KITTI-shaped inputs.  Shared by tests, golden generation,
smoke() and bench.py so every consumer sees the same bytes for a given (seed, shape)."""
import numpy as np


def cloud(B, N, seed, dup_frac=0.0, quantize=None):
    """uniform in the KITTI crop x[-40,40] y[-1,3] z[0,70.4] (config.py:34-36).
    dup_frac: fraction of points overwritten by copies of other points (kitti_dataset.py:243-247
    pads short clouds by re-sampling, so exact duplicates are normal input).
    quantize: snap coordinates to multiples of `quantize` (power of two) so squared distances are
    exact in float32 under any FMA contraction."""
    rng = np.random.default_rng(seed)
    lo = np.array([-40.0, -1.0, 0.0], dtype=np.float32)
    hi = np.array([40.0, 3.0, 70.4], dtype=np.float32)
    pts = (rng.random((B, N, 3), dtype=np.float32) * (hi - lo) + lo).astype(np.float32)
    if quantize:
        pts = (np.round(pts / quantize) * quantize).astype(np.float32)
    if dup_frac > 0:
        nd = int(N * dup_frac)
        for b in range(B):
            dst = rng.choice(N, nd, replace=False)
            src = rng.integers(0, N, nd)
            pts[b, dst] = pts[b, src]
    return pts


def kitti_like_cloud(B, N, seed, dup_frac=0.1, n_objects=24):
    """SURVEY.md §8(d)'s "KITTI-like" variant: point density ~ 1/z (log-uniform depth, what a spinning lidar sampled into a
    fixed budget looks like), lateral position inside the camera frustum (|x| <= 0.85 z, the 1242-px image at fu = 721.5),
    70 % of the points on the ground plane y ~ 1.65 m (camera height), the rest on vertical structure / `n_objects`
    car-sized clusters (3.9 x 1.6 x 1.5 m boxes holding a few hundred points each near the sensor), cropped to the KITTI
    range (config.py:34-36); + dup_frac duplicated indices (kitti_dataset.py:243-247)."""
    rng = np.random.default_rng(seed)
    pts = np.empty((B, N, 3), dtype=np.float32)
    for b in range(B):
        z = (2.5 * (70.4 / 2.5) ** rng.random(N)).astype(np.float32)                       # p(z) ~ 1/z on [2.5, 70.4]
        x = (rng.uniform(-1, 1, N) * np.minimum(0.85 * z, 40.0)).astype(np.float32)
        kind = rng.random(N)
        y = np.where(kind < 0.7, 1.65 + rng.normal(0, 0.03, N), rng.uniform(-1.0, 1.65, N)).astype(np.float32)
        # objects: a share of the points snapped into car-sized boxes standing on the ground, nearer objects get more points
        oz = (4.0 * (60.0 / 4.0) ** rng.random(n_objects)).astype(np.float32)
        ox = (rng.uniform(-0.7, 0.7, n_objects) * np.minimum(0.85 * oz, 35.0)).astype(np.float32)
        w = 1.0 / oz
        take = rng.random(N) < 0.18
        which = rng.choice(n_objects, N, p=w / w.sum())
        box = rng.uniform(-0.5, 0.5, (N, 3)).astype(np.float32) * np.array([1.6, 1.5, 3.9], np.float32)
        x = np.where(take, ox[which] + box[:, 0], x)
        y = np.where(take, 0.9 + box[:, 1], y)
        z = np.where(take, oz[which] + box[:, 2], z)
        pts[b] = np.stack([np.clip(x, -40, 40), np.clip(y, -1, 3), np.clip(z, 0, 70.4)], -1)
        if dup_frac > 0:
            nd = int(N * dup_frac)
            dst = rng.choice(N, nd, replace=False)
            pts[b, dst] = pts[b, rng.integers(0, N, nd)]
    return pts


def packed_cloud(B, N, seed, n_objects=16):
    """worst case for the RCNN stage: ALL points inside `n_objects` car-sized boxes (N / n_objects >= 512 points each), so
    every proposal holds at least 512 distinct points and roipool3d pads nothing"""
    rng = np.random.default_rng(seed)
    pts = np.empty((B, N, 3), dtype=np.float32)
    for b in range(B):
        oz = rng.uniform(6.0, 60.0, n_objects).astype(np.float32)
        ox = (rng.uniform(-0.7, 0.7, n_objects) * np.minimum(0.85 * oz, 35.0)).astype(np.float32)
        which = rng.integers(0, n_objects, N)
        box = rng.uniform(-0.5, 0.5, (N, 3)).astype(np.float32) * np.array([1.6, 1.5, 3.9], np.float32)
        pts[b] = np.stack([ox[which] + box[:, 0], 0.9 + box[:, 1], oz[which] + box[:, 2]], -1)
    return pts


def dense_cloud(B, N, seed, extent=4.0):
    """small-extent cloud so balls contain many points (exercises the >nsample truncation)."""
    rng = np.random.default_rng(seed)
    return (rng.random((B, N, 3), dtype=np.float32) * np.float32(extent)).astype(np.float32)


def proposals(pts, M, seed):
    """M boxes per frame centred on random cloud points: (h,w,l)=(1.526,1.629,3.883)*U(.9,1.1)
    (config.py:38), ry~U(-pi,pi), y = box bottom."""
    rng = np.random.default_rng(seed)
    B, N, _ = pts.shape
    boxes = np.zeros((B, M, 7), dtype=np.float32)
    for b in range(B):
        c = pts[b, rng.integers(0, N, M)]
        hwl = np.array([1.526, 1.629, 3.883], dtype=np.float32) * rng.uniform(0.9, 1.1, (M, 3)).astype(np.float32)
        boxes[b, :, 0] = c[:, 0]
        boxes[b, :, 1] = c[:, 1] + hwl[:, 0] / 2
        boxes[b, :, 2] = c[:, 2]
        boxes[b, :, 3:6] = hwl
        boxes[b, :, 6] = rng.uniform(-np.pi, np.pi, M).astype(np.float32)
    return boxes


def bev_boxes(n, seed, extent=40.0, jitter_clusters=True):
    """n BEV boxes [x1,y1,x2,y2,ry] + distinct scores; clustered so NMS has work to do."""
    rng = np.random.default_rng(seed)
    nc = max(1, n // 12)
    centres = rng.uniform(-extent, extent, (nc, 2)).astype(np.float32)
    which = rng.integers(0, nc, n)
    c = centres[which] + (rng.normal(0, 0.6, (n, 2)).astype(np.float32) if jitter_clusters else 0)
    l = rng.uniform(3.4, 4.4, n).astype(np.float32)
    w = rng.uniform(1.4, 1.9, n).astype(np.float32)
    ry = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    boxes = np.stack([c[:, 0] - l / 2, c[:, 1] - w / 2, c[:, 0] + l / 2, c[:, 1] + w / 2, ry], 1).astype(np.float32)
    scores = rng.permutation(n).astype(np.float32) / np.float32(n)  # all distinct
    return boxes, scores


def pts_xy(pts, W=1280, H=384):
    """project with a fixed KITTI-like pinhole and normalise to [-1,1] (kitti_dataset.py:254-255).  The intrinsics are those of
    the 1280-wide canvas, scaled with W: a smoke-sized canvas (96 x 320) sees the same field of view instead of having every
    point fall outside it (round 4: the tiny tests' LI-Fusion gathers used to read nothing but zero padding)"""
    sc = W / 1280.0
    fu = fv = 721.5 * sc
    cu, cv = 609.6 * sc, 172.9 * sc
    z = np.maximum(pts[..., 2], 0.5)
    u = fu * pts[..., 0] / z + cu
    v = fv * pts[..., 1] / z + cv
    xy = np.stack([u / (W - 1.0) * 2 - 1, v / (H - 1.0) * 2 - 1], -1)
    return xy.astype(np.float32)


def roi_features(P, C, seed):
    rng = np.random.default_rng(seed)
    return np.maximum(rng.normal(0, 1, (P, C)).astype(np.float32), 0)


def mlp_weights(C, H1, H2, seed, scale=None):
    """(W1,b1,W2,b2,w3,b3) xavier-normal like rcnn.py:116-134, non-zero biases to exercise them"""
    rng = np.random.default_rng(seed)
    W1 = rng.normal(0, np.sqrt(2.0 / (C + H1)), (H1, C)).astype(np.float32)
    W2 = rng.normal(0, np.sqrt(2.0 / (H1 + H2)), (H2, H1)).astype(np.float32)
    w3 = rng.normal(0, np.sqrt(2.0 / (H2 + 1)), (H2,)).astype(np.float32)
    b1 = rng.normal(0, 0.05, H1).astype(np.float32)
    b2 = rng.normal(0, 0.05, H2).astype(np.float32)
    b3 = np.float32(rng.normal(0, 0.05))
    return W1, b1, W2, b2, w3, b3


def rpn_output(B, N, seed, z_range=(0.5, 90.0), empty_far=(), empty_near=()):
    """decoded RPN proposals (B, N, 7) [x, y_bottom, z, h, w, l, ry] clustered around N//20 objects per
    frame with depths over z_range (some beyond the 80 m band edge), + distinct scores (B, N).
    Frames listed in empty_far / empty_near get no object in (40, 80] / (0, 40] m."""
    rng = np.random.default_rng(seed)
    props = np.zeros((B, N, 7), dtype=np.float32)
    scores = np.zeros((B, N), dtype=np.float32)
    for b in range(B):
        nc = max(1, N // 20)
        lo, hi = z_range
        if b in empty_far:
            hi = 39.0
        if b in empty_near:
            lo = 41.0
        cz = rng.uniform(lo, hi, nc).astype(np.float32)
        cx = rng.uniform(-30, 30, nc).astype(np.float32)
        which = rng.integers(0, nc, N)
        jit = rng.normal(0, 0.4, (N, 2)).astype(np.float32)
        hwl = np.array([1.526, 1.629, 3.883], dtype=np.float32) * rng.uniform(0.9, 1.1, (N, 3)).astype(np.float32)
        props[b, :, 0] = cx[which] + jit[:, 0]
        props[b, :, 1] = rng.uniform(1.2, 2.0, N).astype(np.float32)
        props[b, :, 2] = cz[which] + jit[:, 1]
        props[b, :, 3:6] = hwl
        props[b, :, 6] = rng.uniform(-np.pi, np.pi, N).astype(np.float32)
        scores[b] = (rng.permutation(N).astype(np.float32) - N / 2) / np.float32(N / 8)   # logits, all distinct
    return scores, props


def image(B, seed, H=384, W=1280, native=(375, 1242)):
    """(B, 3, H, W) N(0,1) image on the 384x1280 canvas, rows / columns beyond the native KITTI size zero
    (kitti_dataset.py:105-106 pads the 375x1242 image)"""
    rng = np.random.default_rng(seed)
    img = rng.standard_normal((B, 3, H, W), dtype=np.float32)
    img[:, :, min(native[0], H):, :] = 0
    img[:, :, :, min(native[1], W):] = 0
    return img


def frames(B, N, seed, H=384, W=1280, native=(375, 1242), dup_frac=0.1, kind="uniform"):
    """(xyz (B,N,3), image (B,3,H,W), pts_xy (B,N,2)): one synthetic KITTI-shaped input batch; kind =
    "uniform" (the crop filled uniformly), "kitti" (density ~ 1/z, ground plane, object clusters) or "packed" (every point in
    one of 16 car-sized boxes: RoIs of >= 512 distinct points)"""
    if kind == "uniform":
        pts = cloud(B, N, seed, dup_frac=dup_frac)
    elif kind == "kitti":
        pts = kitti_like_cloud(B, N, seed, dup_frac=dup_frac)
    elif kind == "packed":
        pts = packed_cloud(B, N, seed)
    else:
        raise ValueError(kind)
    return pts, image(B, seed + 1, H, W, native), pts_xy(pts, W, H)


def seeded_state(shapes, seed):
    """{key: float32 array} for a model state dict given as {key: shape}: a deterministic fill that depends on the sorted keys
    and shapes only, so that two implementations with the same parameter names (the reference's PointRCNN and the engine
    here) get identical weights from a seed instead of from a stored state dict.  Conv / Linear weights ~ N(0, 2 / fan_in),
    1-d tensors: BatchNorm scale and running variance in [0.5, 1.5), running means N(0, 0.3^2), every bias N(0, 0.1^2)."""
    rng = np.random.default_rng(seed)
    out = {}
    for k in sorted(shapes):
        shp = tuple(int(v) for v in shapes[k])
        if k.endswith("num_batches_tracked"):
            continue
        if len(shp) >= 2:
            fan_in = int(np.prod(shp[1:]))
            out[k] = (rng.standard_normal(shp) * np.sqrt(2.0 / max(fan_in, 1))).astype(np.float32)
        elif k.endswith("running_var") or (k.endswith("weight")):
            out[k] = (rng.random(shp) + 0.5).astype(np.float32)
        elif k.endswith("running_mean"):
            out[k] = (rng.standard_normal(shp) * 0.3).astype(np.float32)
        else:
            out[k] = (rng.standard_normal(shp) * 0.1).astype(np.float32)
    return out
