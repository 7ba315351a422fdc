#!/usr/bin/env python
"""Static LDS bank-conflict model of the GEMM epilogue's staging image (gemm_glds.hip: epilogue_stage / epilogue_store),
after the per-instruction banking rules of MI355X_MICROARCH.md (LDS section): lane groups per instruction, bank =
(addr / 4) mod 64 for ds_read_b64 / b128 and mod 32 for every ds_write; identical addresses broadcast; every extra
distinct address on a busy bank inside a lane group costs one more LDS cycle.

    python tools/lds_bank_model.py            # today's row stride and the candidates

Prints, per access pattern, LDS-array cycles per wave instruction: ideal / modelled.  No GPU needed; the numbers to
confirm on the GPU are SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the short-K linears (profiles/r04_pmc_sq_gemm.md)."""
import sys

G_B128 = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
          [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
G_B128 = G_B128 + [[l + 32 for l in g] for g in G_B128]
G_16 = [list(range(16 * i, 16 * i + 16)) for i in range(4)]      # ds_write_b64: contiguous 16-lane groups
G_32 = [list(range(32)), list(range(32, 64))]                    # ds_write_b32
G_8 = [list(range(8 * i, 8 * i + 8)) for i in range(8)]          # ds_write_b128


def cycles(addrs, groups, nbytes, mod):
    """addrs: byte address per lane (None = inactive).  -> LDS-array cycles of the wave instruction"""
    total = 0
    for grp in groups:
        per_bank = {}
        for l in grp:
            a = addrs[l]
            if a is None:
                continue
            for d in range(nbytes // 4):
                dw = a // 4 + d
                per_bank.setdefault(dw % mod, set()).add(dw)
        total += max([len(v) for v in per_bank.values()] or [1])
    return total


def stage_write(RS, wm_rows, NT=5):
    """ds_write_b64 of epilogue_stage's plain path: lane (l15, g) writes 4 f16 at row l15, byte (wn*16*NT + 4 g) * 2 + j * 32"""
    worst = 0
    for j in range(NT):
        addrs = [((l & 15)) * RS + (4 * (l >> 4)) * 2 + j * 32 for l in range(64)]
        worst = max(worst, cycles(addrs, G_16, 8, 32))
    return 4, worst


def geglu_write(RS, NT=5):
    addrs = [(l & 15) * RS + (2 * (l >> 4)) * 2 for l in range(64)]
    return 2, cycles(addrs, G_32, 4, 32)


def store_read_plain(RS, CPR, wave=0, it=0, nthreads=256):
    addrs = []
    for l in range(64):
        c = wave * 64 + l + it * nthreads
        addrs.append((c // CPR) * RS + (c % CPR) * 16)
    return 4, cycles(addrs, G_B128, 16, 64)


def store_read_lnout(RS, j=0):
    addrs = [(l >> 2) * RS + ((l & 3) + 4 * j) * 16 for l in range(64)]
    return 4, cycles(addrs, G_B128, 16, 64)


def store_read_gn(RS):
    addrs = [(l // 20) * RS + (l % 20) * 16 if l < 60 else None for l in range(64)]
    return 4, cycles(addrs, G_B128, 16, 64)


def report(RS, cols=160):
    CPR = cols // 8
    w = stage_write(RS, 0)
    rp = [store_read_plain(RS, CPR, wave=wv, it=it)[1] for wv in range(4) for it in range(3)]
    rl = [store_read_lnout(RS, j)[1] for j in range(CPR // 4)]
    print(f"row stride {RS:4d} B ({cols} columns + {RS - 2 * cols} B pad): staging ds_write_b64 {w[0]} -> {w[1]} cycles | "
          f"store-pass ds_read_b128 plain 4 -> {min(rp)}..{max(rp)} (mean {sum(rp) / len(rp):.2f}) | "
          f"LayerNorm-statistics form 4 -> {min(rl)}..{max(rl)} | GroupNorm-statistics form 4 -> {store_read_gn(RS)[1]}")


if __name__ == "__main__":
    print("160-column tiles (UNet widths):")
    for pad in (16, 0, 8, 32, 48, 64, 80, 96, 112):
        report(320 + pad, 160)
    print("128-column tiles (VAE / Swin / SeeCoder widths):")
    for pad in (16, 0, 32, 64):
        report(256 + pad, 128)
    print("GEGLU staging (80-column image, ds_write_b32):")
    for pad in (16, 0, 32, 48, 64, 96):
        print(f"  row stride {160 + pad}: 2 -> {geglu_write(160 + pad)[1]} cycles")
