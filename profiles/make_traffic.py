"""profiles/<round>_{sa,ops}_pmc_{FETCH,WRITE}_SIZE.txt -> profiles/<round>_traffic.json

    python profiles/make_traffic.py r01

Per-launch HBM bytes of the LARGEST-shape dispatch of each main kernel (the `max` column):
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  WRITE_SIZE is exact on known byte counts and
FETCH_SIZE reads 0.49x of them on gfx950 (calibration in profiles/README.md), hence the 2.
bench.py reads this file for `roofline.traffic` (it runs un-profiled itself).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# bench line -> (workload, kernel-name substring in the rocprof tables)
MAP = {
    "fps_L1": ("sa", "fps_regs2_kernel<16, 1024>"),
    "fps_L2": ("sa", "fps_regs2_kernel<4, 1024>"),
    "ball_query_dual_L1": ("sa", "ball_query_kernel<2,"),
    "group_points_feat_L2": ("sa", "group_points_kernel<true>"),
    "roipool3d": ("ops", "roipool3d_kernel"),
    "three_interpolate_FP4": ("ops", "three_interpolate_lds_kernel"),
    "feature_gather_5": ("ops", "feature_gather_rowpair_kernel"),
    "feature_gather_5_channels_last": ("ops", "feature_gather_cl_kernel"),
    "three_nn_FP4": ("ops", "three_nn_kernel"),
    "rcnn_sa1_fused(fps+ball+group+mlp+max)": ("ops", "sa_mlp_kernel"),
    "nms_normal_6300": ("ops", "nms_mask_kernel<true>"),
}


def column(path, needle, col):
    for line in open(path):
        if needle in line:
            return float(line.split()[-3 + col])   # avg, min, max are the last three fields
    return None


def main(rnd):
    out = {"_source": f"profiles/{rnd}_{{sa,ops}}_pmc_{{FETCH,WRITE}}_SIZE.txt (max over dispatches = largest shape); "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "_calibration": "WRITE_SIZE exact on known byte counts (roipool3d 272419 KiB = the 278.9 MB output slab; "
                           "group_points L2 98304 KiB = 100.7 MB); FETCH_SIZE reads 0.49x of known bytes on "
                           "group_points (6487 KiB vs 13.6 MB) -> x2 as MI355X_MICROARCH.md prescribes"}
    for name, (wl, needle) in MAP.items():
        f = column(os.path.join(HERE, f"{rnd}_{wl}_pmc_FETCH_SIZE.txt"), needle, 2)
        w = column(os.path.join(HERE, f"{rnd}_{wl}_pmc_WRITE_SIZE.txt"), needle, 2)
        if f is None or w is None:
            continue
        out[name] = {"kernel": needle, "fetch_kib": f, "write_kib": w, "bytes": int((2 * f + w) * 1024)}
    json.dump(out, open(os.path.join(HERE, f"{rnd}_traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(f"{k:<44} {v['bytes'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
