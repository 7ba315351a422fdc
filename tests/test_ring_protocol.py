"""CPU model of the slot / wait arithmetic of the register-operand ring kernel (csrc/gemm_glds.hip, gemm160ar_kernel; forced
variants 27 / 45 / 85 / 86).  The kernel cannot run here (no GPU), and its loads are asm statements the compiler does not
count, so the ONLY thing that makes a consumed slot valid is the hand-written `s_waitcnt vmcnt(KEEP)` in front of each barrier.
This restates the loop's bookkeeping -- which slot a step issues into, which slot it consumes, how many VMEM instructions are
younger than the ones it needs -- and checks, for every loop length, that
  * a slot is never refilled before the step that reads it has passed the barrier behind which the refill is issued,
  * the tile a step consumes is the tile that was issued into its slot,
  * the counted wait (or the tail's vmcnt(0)) leaves nothing outstanding that the step reads.
VMEM instructions complete in issue order (loads and LDS-DMA alike), which is what `vmcnt(N)` relies on."""
import pytest


def simulate(nsteps, nbuf, per_step):
    depth = nbuf - 1
    keep = per_step * (depth - 1)
    issued = []            # VMEM instruction stream: the tile each instruction belongs to, in issue order
    slot_tile = {}         # slot -> tile held / in flight
    next_tile = 0

    def issue(slot):
        nonlocal next_tile
        slot_tile[slot] = next_tile
        issued.extend([next_tile] * per_step)   # A registers first, then the weight pieces of the same tile
        next_tile += 1

    for s in range(depth):                       # prologue
        if s < nsteps:
            issue(s)
    consumed = []
    st0 = 0
    while st0 < nsteps:
        for u in range(nbuf):
            st = st0 + u
            if st >= nsteps:
                break
            allowed = keep if st + depth - 1 < nsteps else 0      # the s_waitcnt in front of the barrier
            landed = issued[:max(0, len(issued) - allowed)]       # in-order completion: all but the youngest `allowed`
            needed = [i for i, t in enumerate(issued) if t == st]
            assert needed and max(needed) < len(landed), (nsteps, nbuf, st, "tile not landed behind the wait")
            # refill behind the barrier: the slot of step st - 1, which every wave has left
            if st + depth < nsteps:
                slot = (u + depth) % nbuf
                assert slot != u, "refill must not touch the slot being read"
                if st >= 1:
                    assert slot == (st - 1) % nbuf
                issue(slot)
            assert slot_tile[u] == st, (nsteps, nbuf, st, slot_tile)   # slot u = st % nbuf holds tile st
            consumed.append(slot_tile[u])
        st0 += nbuf
    assert consumed == list(range(nsteps))
    assert next_tile == nsteps                    # every tile issued exactly once, none past the end


@pytest.mark.parametrize("nbuf,per_step", [(7, 9), (7, 5), (7, 7), (3, 7), (5, 9)])
def test_ring_slots_and_counted_wait(nbuf, per_step):
    # per_step: VMEM instructions per wave and K tile = 2 WMB register loads + ceil(20 / waves) weight pieces
    assert per_step * (nbuf - 2) < 64             # vmcnt is 6 bits
    for nsteps in range(1, 40):
        simulate(nsteps, nbuf, per_step)
