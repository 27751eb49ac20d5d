// Build shim for oracle/_ref: compiles the reference's OWN CPU point-in-box / roipool code
// (jmodt/ops/roipool3d/src/roipool3d.cpp:82-195 — pt_in_box3d_cpu, pts_in_boxes3d_cpu,
// roipool3d_cpu) from where it lies under /root/reference.  No reference source is copied.
//
// roipool3d.cpp also contains the two GPU wrappers, which reference roipool3dLauncher{,_slow}
// defined in roipool3d_kernel.cu (CUDA — unbuildable here: no nvcc).  They are declared WEAK
// below, before the reference's own (compatible) declarations, so the shared object loads with
// those two symbols null.  No definition / stand-in is provided: calling `forward` or
// `forward_slow` on this build is an error by construction; only the CPU entry points are used.
//
// TEST INFRASTRUCTURE ONLY (see oracle/jmodt_oracle.c header for the rules).
void roipool3dLauncher_slow(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                            const float* xyz, const float* boxes3d, const float* pts_feature,
                            float* pooled_features, int* pooled_empty_flag) __attribute__((weak));
void roipool3dLauncher(int batch_size, int pts_num, int boxes_num, int feature_in_len, int sampled_pts_num,
                       const float* xyz, const float* boxes3d, const float* pts_feature, float* pooled_features,
                       int* pooled_empty_flag) __attribute__((weak));

#include "/root/reference/jmodt/ops/roipool3d/src/roipool3d.cpp"
