#!/bin/bash
# per-stream timeline of ONE joint-mode training step (rocprofv3 kernel trace): which stream runs what when, 0.5 ms bins.
#   gpurun -- 'bash tools/joint_timeline.sh'
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 python "$REPO/bench.py" --workload train --joint --steps 3 --warmup 2 --no-cpu-baseline > /tmp/warm.log 2>&1
rm -rf /tmp/prof_jtl
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_jtl -o jtl -- python "$REPO/bench.py" --workload train --joint --steps 6 --warmup 2 --no-cpu-baseline > /tmp/prof_jtl.log 2>&1
db=$(find /tmp/prof_jtl -name '*.db' | head -1)
python - "$db" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
fps = [i for i, r in enumerate(rows) if "fps_regs2_kernel<16" in r[0]]
# one step = between two consecutive Adam launches in the timed region
adam = [i for i, r in enumerate(rows) if "fused_adam" in r[0].lower() or "FusedAdam" in r[0]]
print(len(rows), "dispatches;", len(adam), "adam launches")
bursts = [adam[0]]
for i, j in zip(adam, adam[1:]):
    if rows[j][1] - rows[i][1] > 5_000_000:
        bursts.append(j)
lasts = [max(k for k in adam if k < nxt) for nxt in bursts[1:]] + [adam[-1]]      # last adam launch of every burst
a, b = lasts[-3], lasts[-2]
t0, t1 = rows[a][2], rows[b][2]
sel = [r for r in rows[a + 1:b + 1]]
print(f"step window {(t1 - t0) / 1e6:.2f} ms, {len(sel)} dispatches")
def short(n):
    n = n.split("(")[0]
    for k in ("rows_gemm_small", "rows_gemm_kernel<0", "rows_gemm_kernel<1", "rows_gemm_kernel<2", "wgrad_reduce", "fps_regs", "sa_rows", "conv3x3_wino", "conv3x3_rgb", "elementwise", "reduce_kernel", "three_nn", "bq_grid", "ball_query", "train_gemm", "Col2Im", "transpose"):
        if k in n:
            return k
    if "gkgs" in n or "_mh" in n or "igemm" in n or "Conv" in n or "conv" in n or "ck::" in n or "xdl" in n or "pta" in n:
        return "MIOpen-conv"
    if "Cijk" in n or "ROn1" in n or "USL1" in n or "GROn" in n:
        return "Tensile"
    return n[-22:]
streams = collections.OrderedDict()
for r in sel:
    streams.setdefault((r[3], r[4]), []).append(r)
for key, rs in streams.items():
    busy = sum(r[2] - r[1] for r in rs)
    names = collections.Counter()
    for r in rs:
        names[short(r[0])] += r[2] - r[1]
    top = ", ".join(f"{k} {v / 1e6:.2f}" for k, v in names.most_common(6))
    print(f"stream {key}: {len(rs)} kernels, busy {busy / 1e6:.2f} ms, first {(rs[0][1] - t0) / 1e6:.2f} last end {(rs[-1][2] - t0) / 1e6:.2f}  [{top}]")
BIN = 500000
nb = int((t1 - t0) / BIN) + 1
print("\n0.5 ms bins: per stream busy fraction (tenths) and its dominant kernel")
for i in range(nb):
    lo, hi = t0 + i * BIN, t0 + (i + 1) * BIN
    cells = []
    for key, rs in streams.items():
        acc = collections.Counter()
        for r in rs:
            ov = min(r[2], hi) - max(r[1], lo)
            if ov > 0:
                acc[short(r[0])] += ov
        tot = sum(acc.values())
        cells.append(f"{int(10 * tot / BIN):2d} {acc.most_common(1)[0][0][:18] if acc else '':18s}")
    print(f"{i * 0.5:5.1f} | " + " | ".join(cells))
PY
