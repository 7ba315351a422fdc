#!/usr/bin/env python
"""HBM bytes per launch, per bench.py profiler bucket, from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE;
separate runs, ROCm 7.2 rocpd sqlite) over `selftest --replay <launch list>`: the torch-free replay of every
GEMM/conv launch of one UNet forward.  The replay runs the list twice; only the second pass is counted.
Counters are in KB; FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B, MI355X_MICROARCH.md).
usage: python tools/pmc_bucket.py <fetch.db> <write.db> <out.json> [out.md]"""
import json
import re
import sqlite3
import sys


def dispatches(db, counter):
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    t = lambda p: next(x for x in tabs if x.startswith(p))  # noqa: E731
    kd, ks, pe, ip = t("rocpd_kernel_dispatch"), t("rocpd_info_kernel_symbol"), t("rocpd_pmc_event"), t("rocpd_info_pmc")
    scols = [r[1] for r in c.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scols else "kernel_name"
    q = (f"select d.dispatch_id, s.{name_col}, d.grid_size_x, d.end - d.start, sum(e.value) from {kd} d "
         f"join {ks} s on d.kernel_id = s.id join {pe} e on e.event_id = d.event_id join {ip} i on e.pmc_id = i.id "
         f"where i.name = ? group by d.dispatch_id order by d.dispatch_id")
    return [(re.sub(r"\(anonymous namespace\)::", "", n), g, dur, v) for _, n, g, dur, v in c.execute(q, (counter,))]


def bucket(name):
    m = re.match(r"void gemm160_kernel<(\d), (\d), (true|false)", name)
    if m:
        # bench.py's buckets are keyed by the ROW count of the tile (256 / 128 / 64): the 8-wave forms of round 3
        # (<4,2> = 128 rows, <4,1> = 64 rows) and the 256 x 320 GEGLU tile (<4,4,...,10>) share the bucket of their rows
        rows = int(m.group(1)) * int(m.group(2)) * 16
        key = {256: ("4,4", "256x160"), 128: ("2,4", "128x160"), 64: ("2,2", "64x160")}[rows]
        return f"gemm160_kernel<{key[0]}{',conv' if m.group(3) == 'true' else ''}>({key[1]})"
    m = re.match(r"void gemm160ws_kernel<(true|false)", name)
    if m:
        return f"gemm160_kernel<4,4{',conv' if m.group(1) == 'true' else ''}>(256x160)"
    m = re.match(r"void gemm_conv_kernel<(\d), (\d), (true|false)>", name)
    if m:
        return f"gemm_conv_kernel<{m.group(1)},{m.group(2)},{m.group(3)}>"
    if "conv3x3_patch" in name:
        return "conv3x3_patch_kernel(256x160)"
    if "conv3x3_narrow" in name:
        return "conv3x3_narrow_kernel(N<=16)"
    if "splitk_reduce" in name:       # plain, statistics-emitting and fused-GroupNorm forms: one bench.py bucket since round 6
        return "splitk_reduce(+epilogue / GroupNorm)"
    return None


def main(fdb, wdb, out, md=None):
    f, w = dispatches(fdb, "FETCH_SIZE"), dispatches(wdb, "WRITE_SIZE")
    assert len(f) == len(w) and all(a[0] == b[0] and a[1] == b[1] for a, b in zip(f, w)), "passes differ"
    half = len(f) // 2
    agg = {}
    for (name, grid, dur, fv), (_, _, _, wv) in zip(f[half:], w[half:]):
        b = bucket(name)
        if b:
            a = agg.setdefault(b, {"n": 0, "fetch": 0.0, "write": 0.0, "us": 0.0})
            a["n"] += 1
            a["fetch"] += fv * 2048.0
            a["write"] += wv * 1024.0
            a["us"] += dur / 1e3
    res = {b: (a["fetch"] + a["write"]) / a["n"] for b, a in agg.items()}
    res["_note"] = ("HBM bytes per launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from two rocprofv3 --pmc passes over a "
                    "torch-free replay (selftest --replay profiles/unet_c2_gemm_shapes.txt) of the exact GEMM/conv launch "
                    "list of one UNet forward at config C2 (UNet batch 8, 64x64 latent); FETCH_SIZE doubled per "
                    "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B)")
    json.dump(res, open(out, "w"), indent=1)
    lines = ["| bucket | launches per UNet pass | HBM read MB / launch | HBM write MB / launch | avg us (under PMC) |",
             "|---|---|---|---|---|"]
    for b, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        lines.append(f"| `{b}` | {a['n']} | {a['fetch'] / a['n'] / 1e6:.1f} | {a['write'] / a['n'] / 1e6:.1f} | {a['us'] / a['n']:.1f} |")
    print("\n".join(lines))
    if md:
        open(md, "w").write("\n".join(lines) + "\n\n" + res["_note"] + "\n")


if __name__ == "__main__":
    main(*sys.argv[1:5])
