#!/usr/bin/env python
"""Per-kernel SQ counter summary from one rocprofv3 --pmc pass (ROCm 7.2 rocpd sqlite):
MFMA-busy share and LDS bank-conflict share.  Derived columns (gfx94x-style formulas, MI355X_MICROARCH.md):
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel duration x CLOCK_GHZ x 1024 SIMDs)   [clock assumed, see header]
  lds_confl  = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  valu_share = SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES, wait_share = SQ_WAIT_ANY / SQ_WAVE_CYCLES (quad-cycle units, ratios only)
usage: python tools/pmc_sq.py <results.db> [out.md] [skip_fraction]"""
import re
import sqlite3
import sys

CLOCK_GHZ = 2.4


def main(db, out=None, skip=0.5):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    rows = c.execute(f"select d.dispatch_id, s.{name_col}, d.end - d.start, i.name, sum(e.value) from {kd} d "
                     f"join {ks} s on d.kernel_id = s.id join {pe} e on e.event_id = d.event_id "
                     f"join {ip} i on e.pmc_id = i.id group by d.dispatch_id, i.name order by d.dispatch_id").fetchall()
    ids = sorted({r[0] for r in rows})
    first = ids[int(len(ids) * float(skip))] if ids else 0     # drop the warm-up pass of the replay
    agg = {}
    for did, name, dur, cname, val in rows:
        if did < first:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        name = re.sub(r"\(G160Params\)|\(AttnParams\)|\(GemmParams\)", "", name)
        a = agg.setdefault(name, {"ids": set(), "ns": 0.0})
        if did not in a["ids"]:
            a["ids"].add(did)
            a["ns"] += dur
        a[cname] = a.get(cname, 0.0) + val
    lines = ["| kernel | launches | avg us | mfma_busy | lds_confl | valu_share | wait_share |", "|---|---|---|---|---|---|---|"]
    for name, a in sorted(agg.items(), key=lambda kv: -kv[1]["ns"]):
        n = len(a["ids"])
        g = lambda k: a.get(k, 0.0)  # noqa: E731
        mfma = g("SQ_VALU_MFMA_BUSY_CYCLES") / max(1.0, a["ns"] * CLOCK_GHZ * 1024)
        lds = g("SQ_LDS_BANK_CONFLICT") / max(1.0, g("SQ_LDS_IDX_ACTIVE"))
        wc = max(1.0, g("SQ_WAVE_CYCLES"))
        lines.append(f"| `{name[:70]}` | {n} | {a['ns'] / n / 1e3:.1f} | {mfma:.3f} | {lds:.3f} | "
                     f"{g('SQ_ACTIVE_INST_VALU') / wc:.3f} | {g('SQ_WAIT_ANY') / wc:.3f} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + f"\n\n(assumed clock {CLOCK_GHZ} GHz for mfma_busy; PMC passes run at lower clocks, so it is a lower bound)\n")


if __name__ == "__main__":
    main(*sys.argv[1:4])
