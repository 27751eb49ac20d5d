#!/bin/bash
# effective shader clock of the fused SA kernel = GRBM_GUI_ACTIVE / kernel duration (MI355X_MICROARCH.md, DVFS give-back)
#   gpurun -- 'bash tools/kernel_clock.sh [JM_TOOLS_LIB]'
export TMPDIR=/tmp
REPO=$(pwd)
export JM_TOOLS_LIB=${1:-$REPO/tools/bin/libjmodt_hip_tools.so}
cd /tmp
rm -rf /tmp/prof_clk
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d /tmp/prof_clk -o clk -- python "$REPO/tools/sa_exp.py" > /tmp/prof_clk.log 2>&1
tail -3 /tmp/prof_clk.log
db=$(find /tmp/prof_clk -name '*.db' | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select kernel_name, value, duration from counters_collection where counter_name='GRBM_GUI_ACTIVE' and duration > 200000").fetchall()
for k, v, d in rows:
    print(f"{k.split('(')[0][-40:]:40s} {d / 1e6:8.3f} ms  GRBM_GUI_ACTIVE {v:14.0f}  -> {v / d:7.3f} counts/ns  (/8 XCDs = {v / d / 8:.3f} GHz)")
PY
