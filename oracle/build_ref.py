"""Compile oracle/_ref/roipool3d_ref.so from the reference's roipool3d.cpp (CPU functions).

Only runs where /root/reference exists (the authoring container).  Uses g++ directly with the
include/link flags of the installed torch; the reference's own setup.py is NOT run.
TEST INFRASTRUCTURE ONLY.
"""
import os
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
REF_SRC = "/root/reference/jmodt/ops/roipool3d/src/roipool3d.cpp"
OUT = os.path.join(HERE, "_ref", "roipool3d_ref.so")


def main() -> int:
    if not os.path.isfile(REF_SRC):
        print("reference tree absent: skipping oracle/_ref")
        return 0
    src = os.path.join(HERE, "ref_roipool3d_build.cpp")
    if os.path.isfile(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(src), os.path.getmtime(REF_SRC)):
        return 0
    import torch
    from torch.utils import cpp_extension

    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    inc = cpp_extension.include_paths()
    inc.append(sysconfig.get_paths()["include"])
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-DTORCH_EXTENSION_NAME=roipool3d_ref",
           "-DTORCH_API_INCLUDE_EXTENSION_H", f"-D_GLIBCXX_USE_CXX11_ABI={int(torch._C._GLIBCXX_USE_CXX11_ABI)}"]
    cmd += [f"-I{p}" for p in inc]
    cmd += [src, "-o", OUT, f"-L{torch_lib}", f"-Wl,-rpath,{torch_lib}",
            "-ltorch", "-ltorch_cpu", "-lc10", "-ltorch_python"]
    print(" ".join(cmd))
    subprocess.check_call(cmd)
    return 0


if __name__ == "__main__":
    sys.exit(main())
