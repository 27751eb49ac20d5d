"""Build jmodt_amd/csrc/libjmodt_hip.so for gfx950 with hipcc (cross-compiles without a GPU).

    python -m jmodt_amd.csrc.build [--force] [--save-temps] [--tools]

--tools builds tools/bin/libjmodt_hip_tools.so instead: the same sources compiled with -DJM_TOOLS_BUILD (the JM_*
environment switches of the experiment scripts under tools/ become live) plus tools/csrc/*.hip (kernels that were
measured and rejected, kept for the A/B scripts).  The product library contains neither.

One translation unit per op family, compiled in parallel, linked into ONE shared library with a
flat C ABI (include/jmodt_hip.h).  -ffp-contract=off: the kernels spell out every fused
operation they want (see oracle/jmodt_oracle.c header for the floating-point conventions).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["capi.hip", "fps.hip", "ball_query.hip", "ball_query_grid.hip", "three_nn_grid.hip", "pointnet2_gather.hip", "roipool3d.hip", "iou3d.hip",
           "feature_gather.hip", "affinity.hip", "affinity_fused.hip", "affinity_train.hip", "affinity_x3.hip", "sa_mlp.hip", "sa_mlp_pm.hip", "sa_dedupe.hip", "sa_mlp_wide.hip", "sa_groups.hip", "sa_xyz.hip", "li_fusion.hip", "image_fusion.hip", "elementwise.hip", "rcnn_lift.hip", "conv1d_stack.hip", "conv1d_stack64.hip", "conv_rgb.hip", "conv_wino.hip", "sort.hip", "proposal.hip", "detections.hip", "points_gemm.hip", "rows_gemm.hip", "rows_ops.hip", "rows_chain.hip"]
HEADERS = ["jm_common.h", "jm_mfma.h", "jm_grid.h", "fps_common.h", "jm_rows.h", os.path.join(ROOT, "include", "jmodt_hip.h"), os.path.join(ROOT, "include", "jm_detmath.h")]
LIB = os.path.join(HERE, "libjmodt_hip.so")
OBJ_DIR = os.path.join(HERE, "build")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-fvisibility=hidden", "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function", "-DJM_BUILDING"]


def _mtime(p):
    return os.path.getmtime(p) if os.path.exists(p) else 0.0


TOOLS_DIR = os.path.join(ROOT, "tools")
TOOLS_LIB = os.path.join(TOOLS_DIR, "bin", "libjmodt_hip_tools.so")


def _compile(path, obj_dir, flags, force, save_temps):
    obj = os.path.join(obj_dir, os.path.basename(path).replace(".hip", ".o"))
    deps = [path] + [h if os.path.isabs(h) else os.path.join(HERE, h) for h in HEADERS]
    if not force and _mtime(obj) >= max(_mtime(d) for d in deps):
        return obj, False
    cmd = [HIPCC] + flags + ["-c", path, "-o", obj]
    if save_temps:
        cmd += ["-save-temps=obj"]
    subprocess.check_call(cmd, cwd=obj_dir)
    return obj, True


def build(force=False, save_temps=False, verbose=True, tools=False):
    obj_dir = os.path.join(HERE, "build_tools") if tools else OBJ_DIR
    lib = TOOLS_LIB if tools else LIB
    flags = FLAGS + (["-DJM_TOOLS_BUILD", "-I", HERE, "-I", os.path.join(TOOLS_DIR, "csrc"), "-I", os.path.join(ROOT, "include")] + os.environ.get("JM_TOOLS_DEFS", "").split() if tools else [])
    paths = [os.path.join(HERE, s) for s in SOURCES]
    if tools:
        extra = os.path.join(TOOLS_DIR, "csrc")
        paths += sorted(os.path.join(extra, f) for f in os.listdir(extra) if f.endswith(".hip"))
    os.makedirs(obj_dir, exist_ok=True)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(8, len(paths))) as ex:
        res = list(ex.map(lambda s: _compile(s, obj_dir, flags, force, save_temps), paths))
    objs = [o for o, _ in res]
    if force or any(ch for _, ch in res) or _mtime(lib) < max(_mtime(o) for o in objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
        subprocess.check_call(cmd)
        if verbose:
            print(f"built {lib}")
    elif verbose:
        print(f"up to date: {lib}")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, save_temps="--save-temps" in sys.argv, tools="--tools" in sys.argv)
