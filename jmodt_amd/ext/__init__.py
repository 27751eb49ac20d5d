"""Drop-in replacements for the reference's three pybind extension modules.

`pointnet2_cuda`, `iou3d_cuda`, `roipool3d_cuda` expose exactly the `m.def` tables of
jmodt/ops/pointnet2/src/pointnet2_api.cpp:10-24, jmodt/ops/iou3d/src/iou3d.cpp:170-175 and
jmodt/ops/roipool3d/src/roipool3d.cpp:198-203 (same names, argument order and meaning), so the
reference's own Python wrappers run unmodified on top of libjmodt_hip.so.
"""
