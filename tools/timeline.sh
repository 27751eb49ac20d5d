#!/bin/bash
# where is the GPU idle inside a step?  rocprofv3 kernel trace of the default bench -> union of the kernel intervals,
# idle gaps and the kernels either side of the largest ones.   gpurun -- 'bash tools/timeline.sh'
export TMPDIR=/tmp
REPO=$(pwd)
cd /tmp
timeout 600 python "$REPO/tools/plain_steps.py" > /tmp/warm.log 2>&1
rm -rf /tmp/prof_tl
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_tl -o tl -- python "$REPO/tools/plain_steps.py" > /tmp/prof_tl.log 2>&1
db=$(find /tmp/prof_tl -name '*.db' | head -1)
python - "$db" <<'PY'
import sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
rows = cur.execute("select name, start, end, stream_id, queue_id, scratch_size, lds_size, vgpr_count, grid_x, workgroup_x from kernels order by start").fetchall()
print(len(rows), "kernel dispatches")
# the timed region = the last 6 steps: take the last 6/9 of the dispatches by the L1 FPS kernel markers
fps = [i for i, r in enumerate(rows) if "fps_regs2_kernel<16" in r[0]]
print("FPS L1 launches:", len(fps))
lo = rows[fps[-6]][1] if len(fps) >= 7 else rows[0][1]
sel = [r for r in rows if r[1] >= lo]
t0, t1 = sel[0][1], max(r[2] for r in sel)
# union of intervals
iv = sorted((r[1], r[2]) for r in sel)
busy = 0; gaps = []; cs, ce = iv[0]
for s, e in iv[1:]:
    if s > ce:
        gaps.append((s - ce, ce, s)); busy += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy += ce - cs
print(f"window {(t1 - t0) / 1e6:.3f} ms, some kernel running {busy / 1e6:.3f} ms ({100 * busy / (t1 - t0):.1f} %), idle {(t1 - t0 - busy) / 1e6:.3f} ms in {len(gaps)} gaps")
# excluding the FPS side stream (8 CUs busy is not 'the GPU busy')
iv2 = sorted((r[1], r[2]) for r in sel if "fps_regs" not in r[0])
busy2 = 0; gaps2 = []; cs, ce = iv2[0]
for s, e in iv2[1:]:
    if s > ce:
        gaps2.append((s - ce, ce, s)); busy2 += ce - cs; cs, ce = s, e
    else:
        ce = max(ce, e)
busy2 += ce - cs
print(f"without the FPS kernels: busy {busy2 / 1e6:.3f} ms ({100 * busy2 / (t1 - t0):.1f} %), idle {(t1 - t0 - busy2) / 1e6:.3f} ms in {len(gaps2)} gaps; per step idle {(t1 - t0 - busy2) / 6e6:.3f} ms")
hist = {}
for g, a, b in gaps2:
    k = "<5us" if g < 5000 else "<20us" if g < 20000 else "<100us" if g < 100000 else ">=100us"
    hist.setdefault(k, [0, 0]); hist[k][0] += 1; hist[k][1] += g
print({k: (v[0], round(v[1] / 1e6, 3)) for k, v in hist.items()})
names = lambda t: [r[0].split("(")[0][-40:] for r in sel if r[2] == t or r[1] == t]
for g, a, b in sorted(gaps2, reverse=True)[:12]:
    print(f"  gap {g / 1e3:8.1f} us after {names(a)[:1]} before {names(b)[:1]}")
cand = [x for x in sorted(gaps2, reverse=True) if any("attention_fusion" in n for n in names(x[1]))]
g, a, b = cand[1] if len(cand) > 1 else sorted(gaps2, reverse=True)[0]
print("context of an attention_fusion -> conv1d_stack gap (name, stream, start-a us, end-a us):")
for r in sel:
    if r[1] > a - 300000 and r[1] < b + 250000:
        print(f"   {r[0].split('(')[0][-44:]:44s} stream {r[3]} queue {r[4]}  {(r[1] - a) / 1e3:9.1f} {(r[2] - a) / 1e3:9.1f}  scratch {r[5]} lds {r[6]} vgpr {r[7]} grid {r[8]} wg {r[9]}")
PY
