#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace: per-kernel calls / total / avg /
min / max duration and share of GPU kernel time -- the same table `--stats` prints in CSV mode.
usage: python tools/rocpd_stats.py <results.db> [out.md]"""
import re
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = next(t for t in tabs if t.startswith("rocpd_kernel_dispatch"))
    ks = next(t for t in tabs if t.startswith("rocpd_info_kernel_symbol"))
    cols = [r[1] for r in c.execute(f"pragma table_info({kd})")]
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = (f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc")
    rows = c.execute(q).fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["| kernel | calls | total ms | avg us | min us | max us | % |", "|---|---|---|---|---|---|---|"]
    for name, n, tot, mn, mx in rows:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = short if len(short) < 110 else short[:107] + "..."
        lines.append(f"| `{short}` | {n} | {tot / 1e6:.3f} | {tot / n / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | "
                     f"{100.0 * tot / total:.2f} |")
    lines.append(f"\ntotal kernel time: {total / 1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches "
                 f"(columns: {', '.join(cols[:6])}...)")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:3])
