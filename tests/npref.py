"""Independent numpy restatements used to pin the C oracle (tests only).

These are written differently from oracle/jmodt_oracle.c on purpose (vectorised, closed-form
tie rule, float64 polygon clipping) so that an error in one is unlikely to be mirrored in the
other."""
import numpy as np


def fma32(a, b, c):
    """float32 fma via float64 (a*b exact in float64; one extra rounding at the add is below
    float32 resolution except for astronomically rare double-rounding cases)."""
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(np.float32)


def sqdist(p, q):
    """d = fma(dz,dz, fma(dx,dx, dy*dy)) with d* = q - p, float32"""
    d = (q - p).astype(np.float32)
    dx, dy, dz = d[..., 0], d[..., 1], d[..., 2]
    return fma32(dz, dz, fma32(dx, dx, (dy * dy).astype(np.float32)))


def bitrev(v, bits):
    r = np.zeros_like(v)
    for i in range(bits):
        r |= ((v >> i) & 1) << (bits - 1 - i)
    return r


def opt_n_threads(n):
    p = int(np.log(float(n)) / np.log(2.0))
    return max(min(1 << p, 1024), 1)


def fps(xyz, m):
    """closed-form tie rule: among exactly tied maxima the winner minimises
    (bitreverse_{log2 BS}(k mod BS), k)."""
    B, N, _ = xyz.shape
    bs = opt_n_threads(N)
    bits = int(np.log2(bs))
    k = np.arange(N)
    prio = bitrev(k % bs, bits).astype(np.int64) * (N + 1) + k  # smaller = preferred
    out = np.zeros((B, m), dtype=np.int32)
    for b in range(B):
        temp = np.full(N, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, m):
            d = sqdist(xyz[b, old][None, :], xyz[b])
            temp = np.minimum(d, temp)
            mx = temp.max()
            cand = np.nonzero(temp == mx)[0]
            old = int(cand[np.argmin(prio[cand])])
            out[b, j] = old
    return out


def ball_query(radius, nsample, xyz, new_xyz):
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    r2 = np.float32(radius) * np.float32(radius)
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    for b in range(B):
        for i in range(M):
            d = sqdist(xyz[b], new_xyz[b, i][None, :])  # (new - x)
            hits = np.nonzero(d < r2)[0]
            if hits.size:
                h = hits[:nsample]
                idx[b, i, :] = h[0]
                idx[b, i, : h.size] = h
    return idx


def three_nn(unknown, known):
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.full((B, N, 3), np.inf, dtype=np.float32)
    idx = np.zeros((B, N, 3), dtype=np.int32)
    for b in range(B):
        for i in range(N):
            d = sqdist(known[b], unknown[b, i][None, :])
            order = np.argsort(d, kind="stable")[:3]
            d2[b, i, : order.size] = d[order]
            idx[b, i, : order.size] = order
    return d2, idx


# ------------------------------------------------------------------ rotated rectangle overlap
def _corners(box):
    x1, y1, x2, y2, a = [float(v) for v in box]
    cx, cy = (x1 + x2) / 2, (y1 + y2) / 2
    c, s = np.cos(a), np.sin(a)
    pts = []
    for px, py in ((x1, y1), (x2, y1), (x2, y2), (x1, y2)):
        # rotate_around_center (iou3d_kernel.cu:98-102)
        pts.append(((px - cx) * c + (py - cy) * s + cx, -(px - cx) * s + (py - cy) * c + cy))
    return np.array(pts, dtype=np.float64)


def _clip(subject, clipper):
    """Sutherland–Hodgman, float64; clipper must be convex"""
    def area2(p):
        return sum(p[i][0] * p[(i + 1) % len(p)][1] - p[(i + 1) % len(p)][0] * p[i][1] for i in range(len(p)))
    cl = [tuple(p) for p in clipper]
    if area2(cl) < 0:
        cl = cl[::-1]
    out = [tuple(p) for p in subject]
    for i in range(len(cl)):
        a, b = cl[i], cl[(i + 1) % len(cl)]
        inp, out = out, []
        if not inp:
            break
        def side(p):
            return (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
        for j in range(len(inp)):
            p, q = inp[j], inp[(j + 1) % len(inp)]
            sp, sq = side(p), side(q)
            if sp >= 0:
                out.append(p)
            if (sp >= 0) != (sq >= 0):
                t = sp / (sp - sq)
                out.append((p[0] + t * (q[0] - p[0]), p[1] + t * (q[1] - p[1])))
    if len(out) < 3:
        return 0.0
    return abs(sum(out[i][0] * out[(i + 1) % len(out)][1] - out[(i + 1) % len(out)][0] * out[i][1]
                   for i in range(len(out)))) / 2


def overlap_bev(boxes_a, boxes_b):
    out = np.zeros((len(boxes_a), len(boxes_b)), dtype=np.float64)
    ca = [_corners(b) for b in boxes_a]
    cb = [_corners(b) for b in boxes_b]
    for i in range(len(boxes_a)):
        for j in range(len(boxes_b)):
            out[i, j] = _clip(ca[i], cb[j])
    return out


def iou_normal_matrix(b):
    """iou_normal (iou3d_kernel.cu:295-303) vectorised in float32"""
    f = np.float32
    left = np.maximum(b[:, None, 0], b[None, :, 0]); right = np.minimum(b[:, None, 2], b[None, :, 2])
    top = np.maximum(b[:, None, 1], b[None, :, 1]); bottom = np.minimum(b[:, None, 3], b[None, :, 3])
    w = np.maximum((right - left).astype(f), f(0)); h = np.maximum((bottom - top).astype(f), f(0))
    inter = (w * h).astype(f)
    area = ((b[:, 2] - b[:, 0]).astype(f) * (b[:, 3] - b[:, 1]).astype(f)).astype(f)
    den = np.maximum(((area[:, None] + area[None, :]).astype(f) - inter).astype(f), f(1e-8))
    return (inter / den).astype(f)


def greedy_nms(iou_sorted, thresh):
    """textbook greedy NMS on a score-sorted IoU matrix: keep i unless a kept j<i has iou>thr.
    Equivalent to the reference's bitmask reduce (iou3d.cpp:98-114)."""
    n = iou_sorted.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        removed[i + 1:] |= iou_sorted[i, i + 1:] > thresh
    return np.array(keep, dtype=np.int64)
