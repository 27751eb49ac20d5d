"""Turn a rocprofv3 (ROCm 7.2, rocpd SQLite) kernel trace into the per-kernel stats table that is
committed under profiles/.   usage: python profiles/summarize.py <results.db> [out.txt]"""
import sqlite3
import sys


def main(db_path, out_path=None):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = [f"{'kernel':<64} {'calls':>6} {'total_us':>12} {'avg_us':>11} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for n, c, s, a, mn, mx in rows:
        short = n.split("(")[0][-64:]
        lines.append(f"{short:<64} {c:>6} {s / 1e3:>12.1f} {a / 1e3:>11.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100 * s / total:>6.2f}")
    text = "\n".join(lines) + "\n"
    if out_path:
        open(out_path, "w").write(text)
    print(text)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
