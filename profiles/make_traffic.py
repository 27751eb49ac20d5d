"""profiles/<round>_{detect,sa,ops}_pmc_{FETCH,WRITE}_SIZE.txt -> profiles/<round>_traffic.json

    python profiles/make_traffic.py r04

HBM bytes per launch of the bench rows whose device kernels can be identified in the counter passes:
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 of the AVERAGE dispatch of one (kernel, grid, workgroup) shape.
WRITE_SIZE is exact on known byte counts and FETCH_SIZE reads 0.49x of them on gfx950 (calibration in profiles/README.md),
hence the 2.  bench.py reads this file for `roofline.traffic` / `kernels[].traffic_bytes_per_launch` (it runs un-profiled).

Round 4: attribution is per SHAPE, from passes that ran the headline workload only (`bench.py --headline-only`).  Round 3 took
the maximum over all dispatches of a kernel NAME from runs that also executed the other clouds and the dense-RCNN variant after
the timed region: `roipool3d_canonical` showed 598 MB for a 365 MB dispatch and `rcnn_lift_forward_cnt` 550 MB on a 64 us row
(8.6 TB/s).  A row is only listed when its shape is launched by that row alone; rows whose kernels share a persistent grid with
other rows (conv1d_stack, sa_mlp_wide, the listed set-abstraction kernels) are left out rather than guessed.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# bench row (kernels[].kernel) -> (workload, [(kernel-name substring, grid, workgroup), ...] = the kernels one entry launches)
MAP = {
    "affinity_8x128x128/affinity_forward_batched": ("detect", [("affinity_fused_kernel", "2097152", "1024")]),
    "roipool3d_canonical_cnt": ("detect", [("roipool3d_kernel<true, true>", "65536x8", "512")]),
    "conv3x3_rgb_bias_relu": ("detect", [("conv3x3_rgb_kernel", "2560x384x8", "256")]),
    "fps_pyramid/L1/furthest_point_sampling_xyz": ("detect", [("fps_regs2_kernel<16, 1024, true>", "8192", "1024")]),
    "li_fusion_final/image_fusion_gather": ("detect", [("if_gemm_kernel", "1064960", "64")]),
    "rcnn_lift_forward_cnt": ("detect", [("rcnn_lift_kernel", "524288", "256")]),
    "fp1/three_interpolate": ("detect", [("three_interpolate_lds_kernel", "16384x8", "512")]),
    "li_fusion1/feature_gather": ("detect", [("feature_gather_cl_kernel", "4096x4x8", "256")]),
    "li_fusion2/feature_gather": ("detect", [("feature_gather_cl_kernel", "1024x8x8", "256")]),
    "li_fusion_final/attention_fusion_forward": ("detect", [("attention_fusion_kernel", "1048576", "256")]),
    "proposal_layer/decode_rpn_proposals": ("detect", [("decode_rpn_kernel", "131072", "256")]),
    "fps_pyramid/fp_neighbours/three_nn_ws": None,     # four launches of different sizes under one row: not attributable
    # sa workload (configs[1])
    # (a row with TWO launches per step, one per ball-query scale: the sum of both shapes over the 2 launches = bytes per launch)
    "L2/feat/group_points": ("sa", [("group_points_kernel<true>", "8192x12x8", "256"), ("group_points_kernel<true>", "4096x12x8", "256")], 2),
    # ops workload
    "roipool3d_forward": ("ops", [("roipool3d_kernel<true, false>", "65536x8", "512")]),
    "roipool3d_canonical": ("ops", [("roipool3d_kernel<true, true>", "65536x8", "512")]),
    "FP4/three_interpolate": ("ops", [("three_interpolate_lds_kernel", "16384x8", "512")]),
    "map5_channels_last/feature_gather": ("ops", [("feature_gather_cl_kernel", "16384x1x8", "256")]),
    "single/nms": ("ops", [("nms_mask_kernel<true>", None, None)]),
}


def shape_rows(path):
    """[(kernel, grid, wg, dispatches, avg)] of the per-shape section of a summarize.py --pmc table"""
    rows, on = [], False
    for line in open(path):
        if line.startswith("per dispatch shape"):
            on = True
            continue
        if not on or line.startswith("kernel ") or not line.strip():
            continue
        f = line.split()
        try:
            avg, disp = float(f[-3]), int(f[-4])
        except (ValueError, IndexError):
            continue
        rows.append((" ".join(f[:-7]), f[-7], f[-6], disp, avg))
    return rows


def pick(rows, needle, grid, wg):
    """the ONE shape that matches; None when none or several do (ambiguous: not attributed)"""
    hits = [r for r in rows if needle in r[0] and (grid is None or r[1] == grid) and (wg is None or r[2] == wg)]
    if len(hits) != 1:
        return None
    return hits[0]


def main(rnd):
    out = {"_source": f"profiles/{rnd}_{{detect,sa,ops}}_pmc_{{FETCH,WRITE}}_SIZE.txt, per dispatch SHAPE (kernel, grid, workgroup), "
                      "average over the dispatches of the shape, passes run with bench.py --headline-only; "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "_calibration": "WRITE_SIZE exact on known byte counts (roipool3d 272419 KiB = the 278.9 MB output slab; "
                           "group_points L2 98304 KiB = 100.7 MB); FETCH_SIZE reads 0.49x of known bytes on "
                           "group_points (6487 KiB vs 13.6 MB) -> x2 as MI355X_MICROARCH.md prescribes"}
    cache = {}
    for name, spec in MAP.items():
        if spec is None:
            continue
        wl, parts = spec[0], spec[1]
        launches = spec[2] if len(spec) > 2 else 1
        try:
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                if (wl, c) not in cache:
                    cache[(wl, c)] = shape_rows(os.path.join(HERE, f"{rnd}_{wl}_pmc_{c}.txt"))
        except OSError:
            continue
        f = w = 0.0
        kernels, ok = [], True
        for needle, grid, wg in parts:
            rf, rw = pick(cache[(wl, "FETCH_SIZE")], needle, grid, wg), pick(cache[(wl, "WRITE_SIZE")], needle, grid, wg)
            if rf is None or rw is None:
                ok = False
                break
            f += rf[4]
            w += rw[4]
            kernels.append(f"{rf[0]} @ {rf[1]} / {rf[2]} ({rf[3]} dispatches)")
        if not ok:
            continue
        out[name] = {"kernels": kernels, "launches_per_step": launches, "fetch_kib": round(f / launches, 1), "write_kib": round(w / launches, 1),
                     "bytes": int((2 * f + w) * 1024 / launches)}
    json.dump(out, open(os.path.join(HERE, f"{rnd}_traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(f"{k:<48} {v['bytes'] / 1e6:10.2f} MB   {'; '.join(v['kernels'])}")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r04")
