#!/bin/bash
# why does a second FPS pyramid in flight slow the 4-frame step down?  kernel trace of `bench.py --workload train --prefetch-depth D`:
# per stream / queue the L1 sampling kernels and the largest idle gaps of the busiest stream.   gpurun -- 'bash tools/prefetch_depth_timeline.sh 2 train'
export TMPDIR=/tmp
REPO=$(pwd)
D=${1:-2}; W=${2:-train}
cd /tmp
rm -rf /tmp/prof_pd
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_pd -o pd -- python "$REPO/bench.py" --workload $W --prefetch-depth $D --steps 6 --warmup 2 --headline-only --no-cpu-baseline > /tmp/prof_pd.log 2>&1
tail -c 300 /tmp/prof_pd.log
db=$(find /tmp/prof_pd -name '*.db' | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
fps = [r for r in rows if "fps_regs2_kernel<16" in r[0] or "fps_regs2_kernel<8" in r[0] or "fps_regs2_kernel<4" in r[0]]
print(len(rows), "dispatches;", len(fps), "L1-sized FPS launches")
import os, csv
out = os.environ.get("PD_DUMP")
if out:
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        for r in rows[-3000:]:
            w.writerow([r[0][:120], r[1], r[2], r[3], r[4]])
big = [r for r in rows if "fps_regs" in r[0] and r[2] - r[1] > 2e6]
t0 = big[max(0, len(big) - 16)][1]
print("long FPS kernels (start ms, end ms, dur ms, stream, queue):")
for r in big[-16:]:
    print(f"  {(r[1]-t0)/1e6:9.3f} {(r[2]-t0)/1e6:9.3f} {(r[2]-r[1])/1e6:7.3f}  stream {r[3]} queue {r[4]}")
streams = {}
for r in rows:
    if r[1] >= t0:
        streams.setdefault((r[3], r[4]), []).append(r)
for k, v in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    busy = sum(r[2] - r[1] for r in v)
    print(f"stream {k[0]} queue {k[1]}: {len(v)} kernels, busy {busy/1e6:.2f} ms; first {v[0][0].split('(')[0][-36:]}")
main = max(streams.items(), key=lambda kv: len(kv[1]))[1]
gaps = sorted(((b[1] - a[2], a, b) for a, b in zip(main, main[1:])), key=lambda g: -g[0])[:14]
print("largest gaps on the busiest stream:")
for g, a, b in gaps:
    print(f"  {g/1e3:8.1f} us at {(a[2]-t0)/1e6:8.3f} ms after {a[0].split('(')[0][-40:]} before {b[0].split('(')[0][-40:]}")
PY
