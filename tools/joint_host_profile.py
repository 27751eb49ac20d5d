"""Where the HOST time of a joint-mode training step goes (rows route): cProfile over a few steps on the GPU box, the main thread and
the autograd engine's worker thread (threading.setprofile) alike.

    python tools/joint_host_profile.py [steps] > gpurun_out/joint_host_profile.txt

The backward is run on the CALLING thread for the profiled steps (the autograd engine's worker threads are not Python threads and
would not be profiled): the image convolutions then meet MIOpen's find mode anew on that thread — ignore the seconds under
aten.convolution_backward in the listing, the Python side of the backward is what this is for.  Measured (round 5): forward 8.6 ms
of host time per step, backward 7.6 ms on the engine's thread, no single hot spot (the library-call wrapper 2 ms over ~170 calls).
"""
import cProfile
import io
import os
import pstats
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench        # noqa: E402


def main(steps):
    dev = torch.device("cuda:0")
    st = bench.make_joint_state(4, 0, dev)
    for _ in range(4):
        bench.train_step(st, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bench.train_step(st, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"un-profiled: host {1e3 * (t1 - t0) / steps:.2f} ms / step, step {1e3 * (t2 - t0) / steps:.2f} ms")
    profs = []

    def hook(frame, event, arg):      # every thread that starts running Python code gets its own profiler
        p = cProfile.Profile()
        profs.append((threading.current_thread().name, p))
        sys.setprofile(None)
        p.enable()
    threading.setprofile(hook)
    # the autograd engine's worker threads are not Python threads: run the backward on the calling thread so that it is profiled
    torch.autograd.grad_mode.set_multithreading_enabled(False)
    main_p = cProfile.Profile()
    main_p.enable()
    for _ in range(steps):
        bench.train_step(st, None)
    main_p.disable()
    torch.cuda.synchronize()
    for name, p in [("main", main_p)] + profs:
        p.disable()
        s = io.StringIO()
        ps = pstats.Stats(p, stream=s)
        print(f"==== thread {name}: total {ps.total_tt * 1e3 / steps:.2f} ms / step")
        ps.sort_stats("tottime").print_stats(45)
        print("\n".join(s.getvalue().splitlines()[6:56]))
        s = io.StringIO()
        pstats.Stats(p, stream=s).sort_stats("cumulative").print_stats(60)
        print("\n".join(s.getvalue().splitlines()[6:70]))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5)
