#!/bin/bash
# PMC counters of conv3x3_wino_kernel (one counter set per pass): gpurun -- 'bash tools/wino_pmc.sh "SET1" "SET2" ...'
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
for set in "$@"; do
  rm -rf /tmp/prof_wn
  timeout 300 rocprofv3 --pmc $set --kernel-trace -d /tmp/prof_wn -o wn -- env B=8 PYTHONPATH=$REPO python "$REPO/tools/conv_wino_bench.py" > /tmp/prof_wn.log 2>&1
  db=$(find /tmp/prof_wn -name '*.db' | head -1)
  python - "$db" <<'PY'
import sqlite3, sys, collections
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, counter_name, value, duration, grid_size from counters_collection where kernel_name like '%conv3x3_wino_kernel%'").fetchall()
agg = collections.defaultdict(list)
for k, c, v, d, g in rows:
    agg[(g, c)].append((v, d))
for (g, c), vs in sorted(agg.items()):
    v = sum(x[0] for x in vs) / len(vs); d = sum(x[1] for x in vs) / len(vs)
    print(f"grid {g:9d} {c:32s} {v:16.1f}  per ns {v / d:10.3f}   dur {d / 1e3:8.1f} us  (n={len(vs)})")
PY
done
