"""run a script against ANOTHER build of the library (A/B experiments: tools/bin/*.so built with JM_TOOLS_DEFS=...):
    python tools/ab_lib.py tools/bin/libjmodt_hip_deep.so bench.py --no-cpu-baseline --headline-only"""
import os, runpy, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from jmodt_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
