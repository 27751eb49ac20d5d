"""per (kernel, dispatch shape): every counter of the round-6 PMC passes on one line, with the ratios that answer "why":

    python profiles/pmc_digest.py profiles/r06_packed_pmc_ > profiles/r06_packed_counters_digest.txt

SQ_WAVE_CYCLES ~ WAIT_ANY (parked: s_waitcnt / barrier) + WAIT_INST_ANY (issue stall: MFMA dependency / busy pipe) + ACTIVE_INST_ANY
(MI355X_MICROARCH.md: disjoint, quad-cycles); mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel time x clock) is the
matrix pipe's share of the kernel's duration; lds_conflict = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; l2_hit = TCC_HIT / (HIT + MISS)."""
import collections
import re
import sys

prefix = sys.argv[1]
files = {"waves": "sq_waves.txt", "insts": "sq_insts.txt", "lds": "sq_lds.txt", "l2": "l2_hit.txt", "l1": "l1.txt"}
rows = collections.OrderedDict()
for tag, f in files.items():
    try:
        txt = open(prefix + f).read()
    except OSError:
        continue
    sect = txt.split("per dispatch shape")[1] if "per dispatch shape" in txt else ""
    for ln in sect.splitlines()[2:]:
        m = re.match(r"(.{56}) +(\S+) +(\S+) +(\S+) +(\d+) +([\d.]+) ", ln)
        if not m:
            continue
        k, grid, wg, ctr, n, avg = m.groups()
        k = k.strip().replace("void ", "").replace("jm::", "")
        if "jm::" not in ln:
            continue
        rows.setdefault((k, grid, wg), {})[ctr] = float(avg)
print(f"{'kernel':<36} {'grid':>12} {'wg':>5}  parked  stall  active  lds_stall  lds_conflict  l2_hit  waves  mfma_busy_cyc  valu/mfma_insts")
for (k, g, w), c in sorted(rows.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = c.get("SQ_WAVE_CYCLES")
    if not wc:
        continue
    pct = lambda x: f"{100 * c.get(x, 0) / wc:5.1f}%"     # noqa: E731
    conf = c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"] if c.get("SQ_LDS_IDX_ACTIVE") else 0.0
    hit = c.get("TCC_HIT_sum", 0) / (c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)) if c.get("TCC_HIT_sum") else 0.0
    print(f"{k[:36]:<36} {g:>12} {w:>5}  {pct('SQ_WAIT_ANY')} {pct('SQ_WAIT_INST_ANY')} {pct('SQ_ACTIVE_INST_ANY')}   {pct('SQ_WAIT_INST_LDS')}     {100 * conf:6.1f}%   {100 * hit:5.1f}%  {c.get('SQ_WAVES', 0):6.0f}  {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0):13.3g}  "
          f"{c.get('SQ_INSTS_VALU', 0):.3g}/{c.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0):.3g}")
