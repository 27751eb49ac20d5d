"""How much do the discrete outputs north_star wants bit-exact depend on what the reference leaves to nvcc?  (parity #1)

The CPU oracle (oracle/jmodt_oracle.c) and the HIP kernels fix two conventions nobody can check against a CUDA binary here:
squared distances as fma(dz,dz, fma(dx,dx, dy*dy)), and sin / cos / atan2 through include/jm_detmath.h.  This script compiles the
oracle THREE more times into a temporary directory — un-contracted distances (-fmad=false), the other contraction order, the build
host's libm — and counts, on the three benchmark clouds at benchmark size, the FPS picks, ball-query lists, 3-NN rows, roipool3d
in-box index lists and rotated-NMS keep lists that change.  CPU only; nothing of it is in the product or in the checker.

    python tools/parity_exposure.py [frames=2]      -> prints the table of DESIGN.md section 3
"""
import ctypes
import importlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from jmodt_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
VARIANTS = {"checker (fma(dz,dz,fma(dx,dx,dy*dy)); jm_detmath)": [], "un-contracted distances": ["-DORC_EXPOSE_NOFMA"],
            "other contraction order": ["-DORC_EXPOSE_FMA_ALT"], "libm sinf/cosf/atan2f": ["-DORC_EXPOSE_LIBM"]}


def build(tmp, flags, tag):
    so = os.path.join(tmp, f"liborc_{tag}.so")
    fma = ["-mfma"] if " fma " in open("/proc/cpuinfo").read() else []
    subprocess.check_call(["gcc", "-O2", "-std=gnu11", "-ffp-contract=off", "-fno-fast-math", "-fopenmp", "-fPIC"] + fma + flags +
                          ["-shared", "-o", so, os.path.join(ROOT, "oracle", "jmodt_oracle.c"), "-lm"])
    return so


def use(so):
    lib = ctypes.CDLL(so)
    lib.orc_nms.restype = ctypes.c_int
    lib.orc_opt_n_threads.restype = ctypes.c_int
    orc._lib = lib


def run_all(clouds):
    out = {}
    for kind, xyz in clouds.items():
        r = {}
        cur, fps = xyz, []
        for m in (4096, 1024, 256, 64):
            idx = orc.furthest_point_sample(cur, m)
            fps.append(idx)
            cur = np.take_along_axis(cur, idx[..., None].astype(np.int64), axis=1)
        r["fps"] = fps
        r["levels"] = None
        out[kind] = r
    return out


def main():
    clouds = {"uniform": synth.cloud(B, 16384, 1236, dup_frac=0.1), "kitti": synth.kitti_like_cloud(B, 16384, 1236),
              "packed": synth.packed_cloud(B, 16384, 1236)}
    with tempfile.TemporaryDirectory() as tmp:
        sos = {name: build(tmp, flags, str(i)) for i, (name, flags) in enumerate(VARIANTS.items())}
        names = list(VARIANTS)
        use(sos[names[0]])
        base = run_all(clouds)
        # neighbour search / pooling / NMS operands are FIXED to the checker's sampling so that each op is counted on its own
        fixed = {}
        for kind, xyz in clouds.items():
            lv = [xyz]
            for idx in base[kind]["fps"]:
                lv.append(np.take_along_axis(lv[-1], idx[..., None].astype(np.int64), axis=1))
            fixed[kind] = lv
        radii = ((0.1, 16), (0.5, 32), (0.5, 16), (1.0, 32), (1.0, 16), (2.0, 32), (2.0, 16), (4.0, 32))

        def others(kind):
            lv = fixed[kind]
            res = {"ball": [], "nn": [], "pool": None, "nms": None}
            for li in range(4):
                for (rad, ns) in radii[2 * li:2 * li + 2]:
                    res["ball"].append(orc.ball_query(rad, ns, lv[li], lv[li + 1]))
                d3, i3 = orc.three_nn(lv[li], lv[li + 1])
                res["nn"].append(i3)
                res.setdefault("nn_d", []).append(d3)
            boxes = synth.proposals(clouds[kind], 128, 77)
            pf = np.zeros((B, 16384, 4), dtype=np.float32)
            res["pool"] = orc.roipool3d(clouds[kind], pf, orc.enlarge_box3d(boxes, 0.2), 512, return_idx=True)[-1]
            keeps = []
            for seed, n, thr in ((3, 128, 0.1), (4, 1000, 0.1), (5, 1000, 0.8), (6, 6300, 0.85)):
                bev, scores = synth.bev_boxes(n, seed)
                keeps.append(orc.nms(bev, scores, thr, normal=False))
            res["nms"] = keeps
            bev, _ = synth.bev_boxes(300, 9)
            res["iou"] = orc.boxes_iou_bev(bev, bev)
            return res
        base_o = {k: others(k) for k in clouds}
        print(f"{B} frames of 16384 points per cloud; counts = outputs that differ from the checker's convention\n")
        for name in names[1:]:
            use(sos[name])
            print(f"## {name}")
            trig = "libm" in name
            for kind in clouds:
                if not trig:
                    got = run_all({kind: clouds[kind]})[kind]
                    parts = []
                    for li, (a, b) in enumerate(zip(base[kind]["fps"], got["fps"])):
                        diff = int((a != b).sum())
                        first = [int(np.argmax(a[f] != b[f])) if (a[f] != b[f]).any() else -1 for f in range(B)]
                        parts.append(f"L{li + 1} {diff}/{a.size} (first differing pick per frame: {first})")
                    print(f"  {kind:8s} FPS picks (free-running pyramid): " + "; ".join(parts))
                o = others(kind)
                if not trig:
                    bq = [f"{int((a != b).any(-1).sum())}/{a.shape[0] * a.shape[1]}" for a, b in zip(base_o[kind]["ball"], o["ball"])]
                    nn = [f"{int((a != b).any(-1).sum())}/{a.shape[0] * a.shape[1]}" for a, b in zip(base_o[kind]["nn"], o["nn"])]
                    dd = np.concatenate([(a != b).reshape(-1) for a, b in zip(base_o[kind]["nn_d"], o["nn_d"])])
                    print(f"  {kind:8s} ball-query lists (8 scales, checker's centres): {', '.join(bq)};  3-NN rows (4 levels): {', '.join(nn)}"
                          f"  [the convention IS live: {100 * dd.mean():.1f} % of the 3-NN distances differ in the last bit]")
                else:
                    pa, pb = base_o[kind]["pool"], o["pool"]
                    pool = int((pa != pb).any(-1).sum())
                    nm = [f"{'same' if (len(a) == len(b) and (a == b).all()) else f'DIFFERENT ({len(a)} vs {len(b)} kept)'}" for a, b in zip(base_o[kind]["nms"], o["nms"])]
                    di = base_o[kind]["iou"] != o["iou"]
                    print(f"  {kind:8s} roipool3d index lists: {pool}/{pa.shape[0] * pa.shape[1]} RoIs differ;  rotated NMS keep lists "
                          f"(128 @ 0.1, 1000 @ 0.1, 1000 @ 0.8, 6300 @ 0.85): {', '.join(nm)}"
                          f"  [live: {100 * di.mean():.2f} % of 300 x 300 rotated IoUs differ in the last bits, max |diff| {np.abs(base_o[kind]['iou'] - o['iou']).max():.2e}]")
            print()
    importlib.reload(orc)


if __name__ == "__main__":
    main()
