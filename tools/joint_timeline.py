"""Per-dispatch timeline of the LAST joint-mode training step of a rocprofv3 kernel trace (rocpd SQLite database):

    cd /tmp && rocprofv3 --kernel-trace -d /tmp/jt -o jt -- python $REPO/bench.py --workload train --joint --steps 4 --warmup 3 --no-cpu-baseline
    python tools/joint_timeline.py /tmp/jt/.../jt_results.db gpurun_out/joint_timeline.csv

One row per dispatch: start (us after the step's first dispatch), duration, queue, kernel, grid, workgroup.  Steps are cut at the
fused Adam launches (multi_tensor_apply).  The table answers what a stats summary cannot: which dispatches are on the chain the
step waits for, and how long each shape of a kernel takes inside the composed step."""
import sqlite3
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/profiles")
from summarize import _shape_cols, short        # noqa: E402


def main(db, out):
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    grid, wg = _shape_cols(cols)
    q = [c for c in cols if c.lower() in ("queue_id", "queue", "stream_id", "stream")]
    sel = ", ".join([name, "start", "end"] + (grid or []) + (wg or []) + q[:2])
    rows = cur.execute(f"select {sel} from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "multi_tensor_apply" in r[0]]
    # the last step = dispatches between the last two groups of Adam launches
    cuts = [i for j, i in enumerate(adam) if j == 0 or i - adam[j - 1] > 50]
    lo, hi = (cuts[-2], cuts[-1]) if len(cuts) >= 2 else (0, len(rows))
    step = rows[lo:hi]
    t0 = step[0][1]
    ng = len(grid or [])
    with open(out, "w") as f:
        f.write("start_us,dur_us,queue,kernel,grid,wg\n")
        for r in step:
            g = "x".join(str(v) for v in r[3:3 + ng] if v not in (1, None)) or "1"
            w = "x".join(str(v) for v in r[3 + ng:3 + 2 * ng] if v not in (1, None)) or "1"
            qq = "/".join(str(v) for v in r[3 + 2 * ng:])
            f.write(f"{(r[1] - t0) / 1e3:.2f},{(r[2] - r[1]) / 1e3:.2f},{qq},{short(r[0]).replace(',', ';')},{g},{w}\n")
    print(f"{len(step)} dispatches, {(step[-1][2] - t0) / 1e6:.3f} ms; columns available: {cols}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
