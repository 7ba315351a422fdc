#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; ROCm 7.2 rocpd
sqlite).  Consecutive dispatches of one kernel with one grid are a "run" (the selftest bench
launches every shape 3 + 20 times back to back); the table gives the mean counter value per
launch for every run.  Counters are in KB (x1024); on gfx950 FETCH_SIZE tallies 128-B requests
at 64 B for wide coalesced streams (MI355X_MICROARCH.md, HBM), so reads are also shown x2.
usage: python tools/rocpd_pmc.py <fetch.db> <write.db> [out.json]"""
import json
import re
import sqlite3
import sys


def runs(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = (f"select d.dispatch_id, s.{name_col}, d.grid_size_x, d.grid_size_y, d.grid_size_z, d.end - d.start, sum(e.value) "
         f"from {kd} d join {ks} s on d.kernel_id = s.id join {pe} e on e.event_id = d.event_id "
         f"join {ip} i on e.pmc_id = i.id where i.name = ? group by d.dispatch_id order by d.dispatch_id")
    out, cur = [], None
    for did, name, gx, gy, gz, dur, val in c.execute(q, (counter,)):
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        key = (name, gx, gy, gz)
        if cur is None or cur["key"] != key:
            cur = {"key": key, "n": 0, "sum": 0.0, "dur": 0.0}
            out.append(cur)
        cur["n"] += 1
        cur["sum"] += val
        cur["dur"] += dur
    return out


def main(fdb, wdb, out=None):
    fr, wr = runs(fdb, "FETCH_SIZE"), runs(wdb, "WRITE_SIZE")
    rows = []
    for f, w in zip(fr, wr):
        if f["key"] != w["key"]:
            continue
        name, gx, gy, gz = f["key"]
        rows.append({"kernel": name, "grid": [gx, gy, gz], "launches": f["n"],
                     "fetch_bytes_raw": f["sum"] / f["n"] * 1024, "fetch_bytes_x2": f["sum"] / f["n"] * 2048,
                     "write_bytes": w["sum"] / w["n"] * 1024, "avg_us_profiled": f["dur"] / f["n"] / 1e3})
    print("| kernel | grid | launches | FETCH raw MB | FETCH x2 MB | WRITE MB | avg us (profiled) |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        k = r["kernel"] if len(r["kernel"]) < 70 else r["kernel"][:67] + "..."
        print(f"| `{k}` | {r['grid']} | {r['launches']} | {r['fetch_bytes_raw'] / 1e6:.2f} | "
              f"{r['fetch_bytes_x2'] / 1e6:.2f} | {r['write_bytes'] / 1e6:.2f} | {r['avg_us_profiled']:.1f} |")
    if out:
        # aggregate per bench.py bucket name: mean HBM bytes per launch (reads x2-corrected + writes)
        def bucket(name):
            m = re.match(r"void gemm160_kernel<(\d), (\d), (true|false)", name)
            if m:
                tile = {"44": "256x160", "24": "128x160", "22": "64x160"}[m.group(1) + m.group(2)]
                return f"gemm160_kernel<{m.group(1)},{m.group(2)}{',conv' if m.group(3) == 'true' else ''}>({tile})"
            m = re.match(r"void gemm_conv_kernel<(\d), (\d), (true|false)>", name)
            if m:
                return f"gemm_conv_kernel<{m.group(1)},{m.group(2)},{m.group(3)}>"
            if name.startswith("void attention_kernel"):
                return "attention_kernel"
            return None
        agg = {}
        for r in rows:
            b = bucket(r["kernel"])
            if b and r["launches"] >= 10:
                agg.setdefault(b, []).append(r["fetch_bytes_x2"] + r["write_bytes"])
        res = {b: sum(v) / len(v) for b, v in agg.items()}
        res["_note"] = ("mean over the representative UNet shapes of the torch-free selftest bench (rocprofv3 --pmc "
                        "FETCH_SIZE / WRITE_SIZE in separate passes; FETCH_SIZE x2 per MI355X_MICROARCH.md HBM section); "
                        "per-shape values in the .md table next to this file")
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
