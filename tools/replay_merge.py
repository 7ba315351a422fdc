#!/usr/bin/env python
"""Merge `selftest --replay-time <shapes> <tile>` logs (one per forced tile code) into a per-problem
table: auto-selected time vs the best forced variant.  usage: replay_merge.py <dir> <tile> [<tile> ...]"""
import collections
import sys


def load(path):
    seen, out = collections.Counter(), {}
    for line in open(path):
        if '|' not in line or 'us/launch' in line:
            continue
        k, v = line.split('|')
        k, v = ' '.join(k.split()), v.split()
        seen[(k, int(v[0]))] += 1
        out[(k, int(v[0]), seen[(k, int(v[0]))])] = (float(v[1]), float(v[4]))
    return out


def main(d, tiles, top=45):
    data = {t: load(f'{d}/replay_t{t}.log') for t in tiles}
    rows = []
    for key, (us0, ideal) in data[tiles[0]].items():
        alt = {t: data[t][key][0] for t in tiles[1:] if key in data[t]}
        best = min(alt, key=alt.get) if alt else tiles[0]
        gain = key[1] * max(0.0, us0 - alt.get(best, us0))
        rows.append((gain, key, us0, ideal, best, alt))
    rows.sort(reverse=True)
    for gain, key, us0, ideal, best, alt in rows[:top]:
        print(f"{key[0]:36s} n={key[1]:2d} auto {us0:6.1f} ideal {ideal:5.1f} best {best}:{alt.get(best, us0):6.1f} "
              f"gain {gain:6.1f} | " + ' '.join(f"{t}:{u:.0f}" for t, u in alt.items()))
    print("sum of gains, us per pass:", round(sum(r[0] for r in rows), 1))


if __name__ == '__main__':
    main(sys.argv[1], [int(t) for t in sys.argv[2:]])
