"""profiles/<round>_{sa,ops}_pmc_{FETCH,WRITE}_SIZE.txt -> profiles/<round>_traffic.json

    python profiles/make_traffic.py r02

Per-launch HBM bytes of the LARGEST-shape dispatch of each main kernel (the `max` column):
bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024.  WRITE_SIZE is exact on known byte counts and
FETCH_SIZE reads 0.49x of them on gfx950 (calibration in profiles/README.md), hence the 2.
bench.py reads this file for `roofline.traffic` (it runs un-profiled itself).
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
# bench row (kernels[].kernel) -> (workload, kernel-name substring in the rocprof tables)
MAP = {
    # default workload (detect): the largest-shape dispatch of each kernel = the row named here
    "rcnn_sa1/sa_mlp_pm_forward": ("detect", "sa_mlp_pm_kernel"),          # largest dispatch = the DENSE RCNN SA1 (bench variant)
    "affinity_8x128x128/affinity_forward_batched": ("detect", ["mlp_gemm_kernel<0", "mlp_gemm_kernel<1"]),   # the entry's two GEMMs
    "rpn_sa1/ball_query_dual_ws": ("detect", ["bq_grid_build_kernel<16", "bq_grid_query_kernel<2>"]),
    "rcnn_lift_forward_cnt": ("detect", "rcnn_lift_kernel"),
    "conv3x3_rgb_bias_relu": ("detect", "conv3x3_rgb_kernel"),
    "conv1d_stack_forward": ("detect", "conv1d_stack_kernel"),
    "fps_pyramid/L1/furthest_point_sampling_xyz": ("detect", "fps_regs2_kernel<16, 1024>"),
    "roipool3d_canonical": ("detect", "roipool3d_kernel"),
    "li_fusion_final/image_fusion_gather": ("detect", "if_gemm_kernel"),
    "rcnn_lift_forward": ("detect", "rcnn_lift_kernel"),
    "li_fusion_final/attention_fusion_forward": ("detect", "attention_fusion_kernel"),
    "fp1/three_interpolate": ("detect", "three_interpolate_lds_kernel"),
    "fp1/three_nn": ("detect", "three_nn_kernel"),
    "rpn_sa1/ball_query_dual": ("detect", "ball_query_kernel<2,"),
    "li_fusion1/feature_gather": ("detect", "feature_gather_cl_kernel"),
    "bias_relu_channels_last": ("detect", "bias_relu_cl_kernel"),
    "rpn_sa4/sa_mlp_forward": ("detect", "sa_mlp_wide_kernel"),
    # sa workload (configs[1])
    "fps_pyramid/L2/furthest_point_sampling_xyz": ("sa", "fps_regs2_kernel<4, 1024>"),
    "L1/ball_query_dual_ws": ("sa", ["bq_grid_build_kernel<16", "bq_grid_query_kernel<2>"]),
    "L2/feat/group_points": ("sa", "group_points_kernel<true>"),
    # ops workload
    "roipool3d_forward": ("ops", "roipool3d_kernel"),
    "FP4/three_interpolate": ("ops", "three_interpolate_lds_kernel"),
    "map5_nchw/feature_gather": ("ops", "feature_gather_rowpair_kernel"),
    "map5_channels_last/feature_gather": ("ops", "feature_gather_cl_kernel"),
    "FP4/three_nn": ("ops", "three_nn_kernel"),
    "single/nms": ("ops", "nms_mask_kernel<true>"),
}


def column(path, needle, col):
    """the avg / min / max field (col 0 / 1 / 2) of the first row naming `needle` (the per-name table comes first); a LIST of
    needles = the kernels one entry launches: their sum"""
    if isinstance(needle, (list, tuple)):
        vals = [column(path, n, col) for n in needle]
        return None if any(v is None for v in vals) else sum(vals)
    for line in open(path):
        if needle in line:
            return float(line.split()[-3 + col])   # avg, min, max are the last three fields
    return None


def main(rnd):
    out = {"_source": f"profiles/{rnd}_{{detect,sa,ops}}_pmc_{{FETCH,WRITE}}_SIZE.txt (max over dispatches = largest shape); "
                      "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024",
           "_calibration": "WRITE_SIZE exact on known byte counts (roipool3d 272419 KiB = the 278.9 MB output slab; "
                           "group_points L2 98304 KiB = 100.7 MB); FETCH_SIZE reads 0.49x of known bytes on "
                           "group_points (6487 KiB vs 13.6 MB) -> x2 as MI355X_MICROARCH.md prescribes"}
    for name, (wl, needle) in MAP.items():
        try:
            f = column(os.path.join(HERE, f"{rnd}_{wl}_pmc_FETCH_SIZE.txt"), needle, 2)
            w = column(os.path.join(HERE, f"{rnd}_{wl}_pmc_WRITE_SIZE.txt"), needle, 2)
        except OSError:
            continue
        if f is None or w is None:
            continue
        out[name] = {"kernel": needle if isinstance(needle, str) else " + ".join(needle), "fetch_kib": f, "write_kib": w,
                     "bytes": int((2 * f + w) * 1024)}
    json.dump(out, open(os.path.join(HERE, f"{rnd}_traffic.json"), "w"), indent=1)
    for k, v in out.items():
        if isinstance(v, dict):
            print(f"{k:<44} {v['bytes'] / 1e6:10.2f} MB")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r03")
