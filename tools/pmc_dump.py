#!/usr/bin/env python
"""Generic per-kernel dump of one rocprofv3 --pmc pass (ROCm 7.2 rocpd sqlite): launches, average duration and the mean of
every collected counter per launch, plus per-SIMD-cycle shares where SQ_WAVE_CYCLES / GRBM_GUI_ACTIVE are present.
usage: python tools/pmc_dump.py <results.db> [out.md] [name filter regex]"""
import re
import sqlite3
import sys


def main(db, out=None, flt=None):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = c.execute(f"select d.dispatch_id, s.{name_col}, d.grid_size_x, d.end - d.start, i.name, sum(e.value) from {kd} d "
                     f"join {ks} s on d.kernel_id = s.id join {pe} e on e.event_id = d.event_id "
                     f"join {ip} i on e.pmc_id = i.id group by d.dispatch_id, i.name order by d.dispatch_id").fetchall()
    agg, counters = {}, []
    for did, name, gx, dur, cname, val in rows:
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(G160Params\)|\(AttnParams\)|\(GemmParams\)", "", name)
        if flt and not re.search(flt, name):
            continue
        key = (name, gx)
        a = agg.setdefault(key, {"ids": set(), "ns": 0.0})
        if did not in a["ids"]:
            a["ids"].add(did)
            a["ns"] += dur
        a[cname] = a.get(cname, 0.0) + val
        if cname not in counters:
            counters.append(cname)
    lines = ["| kernel | grid | launches | avg us | " + " | ".join(counters) + " |", "|---|---|---|---|" + "---|" * len(counters)]
    for (name, gx), a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        n = len(a["ids"])
        lines.append(f"| `{name[:60]}` | {gx} | {n} | {a['ns'] / n / 1e3:.1f} | " +
                     " | ".join(f"{a.get(k, 0.0) / n:.4g}" for k in counters) + " |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")   # ("" = print only)


if __name__ == "__main__":
    main(*sys.argv[1:4])
