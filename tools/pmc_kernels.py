#!/usr/bin/env python
"""Per-kernel HBM traffic and bandwidth from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, ROCm 7.2
rocpd sqlite) over the same torch-free command (selftest --bench-attn / --bench-gn): for every kernel name the launches,
average duration, HBM MB read / written per launch and the resulting GB/s.  FETCH_SIZE is doubled (gfx950 tallies 128-byte
requests at 64 B, MI355X_MICROARCH.md); durations are those of the FETCH pass (PMC passes run at lower clocks than an
un-profiled run, so GB/s is a lower bound).
usage: python tools/pmc_kernels.py <fetch.db> <write.db> <out.md> [<json to merge 'name -> bytes per launch' into>]"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_bucket import dispatches  # noqa: E402


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:90]


def main(fdb, wdb, out, merge=None):
    f, w = dispatches(fdb, "FETCH_SIZE"), dispatches(wdb, "WRITE_SIZE")
    assert len(f) == len(w), "passes differ"
    agg = {}
    for (name, grid, dur, fv), (_, _, _, wv) in zip(f, w):
        a = agg.setdefault(short(name), {"n": 0, "fetch": 0.0, "write": 0.0, "us": 0.0})
        a["n"] += 1
        a["fetch"] += fv * 2048.0
        a["write"] += wv * 1024.0
        a["us"] += dur / 1e3
    lines = ["| kernel | launches | avg us (under PMC) | HBM read MB / launch | HBM write MB / launch | HBM GB/s |",
             "|---|---|---|---|---|---|"]
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        byt = (a["fetch"] + a["write"]) / a["n"]
        us = a["us"] / a["n"]
        lines.append(f"| `{k}` | {a['n']} | {us:.1f} | {a['fetch'] / a['n'] / 1e6:.2f} | {a['write'] / a['n'] / 1e6:.2f} | "
                     f"{byt / us / 1e3:.0f} |")
    txt = "\n".join(lines)
    print(txt)
    open(out, "w").write(txt + "\n\n(HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE, KB counters, two separate rocprofv3 --pmc passes)\n")
    if merge:
        d = json.load(open(merge)) if os.path.exists(merge) else {}
        for k, a in agg.items():
            if "attention" in k:
                key = "attention_kernel"
            elif k.startswith("gn_") or "gn_" in k:
                key = "groupnorm kernels (per kernel launch: stats, apply or small)"
            elif "layernorm" in k or "ln_rowstats" in k:
                key = "layernorm_kernel"
            else:
                continue
            e = d.setdefault("_per_kernel_" + key, {"n": 0, "bytes": 0.0})
            e["n"] += a["n"]
            e["bytes"] += a["fetch"] + a["write"]
        for k in list(d):
            if k.startswith("_per_kernel_") and isinstance(d[k], dict):
                d[k[len("_per_kernel_"):]] = d[k]["bytes"] / max(1, d[k]["n"])
                del d[k]
        json.dump(d, open(merge, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:5])
