"""Turn rocprofv3 (ROCm 7.2, rocpd SQLite) output into the small text tables committed here.

    python profiles/summarize.py <results.db> [out.txt]            per-kernel time stats, then the same per SHAPE:
                                                                   one row per (kernel, grid, workgroup), so that a kernel
                                                                   launched at several shapes in one workload (sa_mlp_pm:
                                                                   RPN SA2 x2, RCNN SA1, SA2) has each shape's own average
    python profiles/summarize.py --pmc <results.db> [out.txt]      per-kernel PMC counter averages (+ per shape)
"""
import sqlite3
import sys


def short(name):
    """jm:: kernels keep their (short) template arguments; library kernels with page-long template lists (composable_kernel
    convolutions picked by MIOpen's find mode) become `prefix<..>#hash` instead of the meaningless last 64 characters"""
    base = name.split("(")[0]
    if base.startswith("_Z") and len(base) > 64:        # a mangled symbol: spell out its leading nested name
        import re
        import zlib
        parts, rest = [], base[3:] if base.startswith("_ZN") else base[2:]
        while True:
            m = re.match(r"(\d+)", rest)
            if not m:
                break
            n = int(m.group(1))
            parts.append(rest[len(m.group(1)):len(m.group(1)) + n])
            rest = rest[len(m.group(1)) + n:]
        return f"{'::'.join(parts)[-52:]}<..>#{zlib.crc32(base.encode()) & 0xffff:04x}"
    if len(base) > 64 and "<" in base and "jm::" not in base:
        import zlib
        head = base.split("<")[0].split(" ")[-1]
        return f"{head[-44:]}<..>#{zlib.crc32(base.encode()) & 0xffff:04x}"
    return base[-64:]


def stats(db_path, out_path=None):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    # MIOpen's find mode times every applicable solver once per process and shape, its reference kernel included
    # (naive_conv_*: ~0.15 s each): start-up probes, not part of any step — listed at the end, outside the percentages
    probes = [r for r in rows if r[0].startswith("naive_conv")]
    rows = [r for r in rows if not r[0].startswith("naive_conv")]
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<64} {'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f"{short(n):<64} {c:>6} {s / 1e3:>12.1f} {a / 1e3:>11.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * s / total:>6.2f}")
    for n, c, s, a, mn, mx in probes:
        lines.append(f"(start-up probe of MIOpen find mode, not in pct) {short(n)[-40:]}: {c} calls, {s / 1e3:.1f} us")
    lines += ["", shape_table(cur, cols, name_col)]
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


def _shape_cols(cols):
    """the grid / workgroup columns of a rocpd view, whatever this ROCm calls them"""
    def pick(*stems):
        found = []
        for axis in ("x", "y", "z"):
            hit = [c for c in cols if any(c.lower() in (f"{st}_{axis}", f"{st}_size_{axis}", f"{st}{axis}") for st in stems)]
            if not hit:
                return None
            found.append(hit[0])
        return found
    return pick("grid"), pick("workgroup", "block", "wg")


def shape_table(cur, cols, name_col, top=160):
    grid, wg = _shape_cols(cols)
    if not grid or not wg:
        return f"(no grid / workgroup columns in the kernels view: {cols})"
    key = ", ".join([name_col] + grid + wg)
    rows = cur.execute(f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                       f"where {name_col} not like 'naive_conv%' group by {key} order by sum(end-start) desc limit {top}").fetchall()
    lines = [f"per dispatch shape (top {top} by total time; grid = work-items as rocprofv3 records them)",
             f"{'kernel':<56} {'grid':>18} {'wg':>12} {'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>10} {'max_us':>10}"]
    for r in rows:
        n, g, w = r[0], r[1:4], r[4:7]
        c, s, a, mn, mx = r[7:]
        gs = "x".join(str(v) for v in g if v not in (1, None)) or "1"
        ws = "x".join(str(v) for v in w if v not in (1, None)) or "1"
        lines.append(f"{short(n)[-56:]:<56} {gs:>18} {ws:>12} {c:>6} {s / 1e3:>12.1f} {a / 1e3:>11.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f}")
    return "\n".join(lines)


def pmc(db_path, out_path=None):
    cur = sqlite3.connect(db_path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    if not cols:
        print("no counters_collection view; tables:", [r[0] for r in cur.execute("select name from sqlite_master")][:80])
        return
    kcol = [c for c in cols if "kernel" in c and "name" in c] or [c for c in cols if c == "name"]
    ccol = [c for c in cols if "counter" in c and "name" in c]
    vcol = [c for c in cols if c in ("value", "counter_value")]
    if not (kcol and ccol and vcol):
        print("unexpected schema:", cols)
        return
    rows = cur.execute(f"select {kcol[0]}, {ccol[0]}, count(*), avg({vcol[0]}), min({vcol[0]}), max({vcol[0]}) "
                       f"from counters_collection group by {kcol[0]}, {ccol[0]} order by {kcol[0]}").fetchall()
    lines = [f"{'kernel':<64} {'counter':<14} {'dispatches':>10} {'avg':>16} {'min':>16} {'max':>16}"]
    for k, c, n, a, mn, mx in rows:
        lines.append(f"{short(k):<64} {c:<14} {n:>10} {a:>16.1f} {mn:>16.1f} {mx:>16.1f}")
    grid, wg = _shape_cols(cols)
    if grid and wg:
        key = ", ".join([kcol[0]] + grid + wg + [ccol[0]])
        rows = cur.execute(f"select {key}, count(*), avg({vcol[0]}), min({vcol[0]}), max({vcol[0]}) from counters_collection "
                           f"group by {key} order by avg({vcol[0]}) desc").fetchall()
        # every shape of the hand-written (jm::) kernels — profiles/make_traffic.py attributes counters per (kernel, grid,
        # workgroup) — and the 40 largest library shapes
        lib_rows = [r for r in rows if "jm::" not in r[0]][:40]
        rows = [r for r in rows if "jm::" in r[0]] + lib_rows
        lines += ["", "per dispatch shape (every jm:: shape, then the 40 largest library shapes, by counter average)",
                  f"{'kernel':<56} {'grid':>18} {'wg':>12} {'counter':<14} {'dispatches':>10} {'avg':>16} {'min':>16} {'max':>16}"]
        for r in rows:
            gs = "x".join(str(v) for v in r[1:4] if v not in (1, None)) or "1"
            ws = "x".join(str(v) for v in r[4:7] if v not in (1, None)) or "1"
            lines.append(f"{short(r[0])[-56:]:<56} {gs:>18} {ws:>12} {r[7]:<14} {r[8]:>10} {r[9]:>16.1f} {r[10]:>16.1f} {r[11]:>16.1f}")
    else:
        lines += ["", f"(no grid / workgroup columns in counters_collection: {cols})"]
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0] == "--pmc":
        pmc(args[1], args[2] if len(args) > 2 else None)
    else:
        stats(args[0], args[1] if len(args) > 1 else None)
