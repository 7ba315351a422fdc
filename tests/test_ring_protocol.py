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


# ---------------------------------------------------------------------------------------------------------------------
# conv3x3_patch_fl_kernel (forced variant 95): 4 loader waves and 8 consumer waves hand the taps over through progress
# words in LDS instead of a block-wide barrier.  Model: every wave is a little state machine, a scheduler picks which
# wave moves next (randomly, or adversarially: one loader / one consumer starved as long as the protocol lets the others
# run).  Checked at every consumer read and every loader refill:
#   * a consumer reads tap T only when ALL four loaders' pieces of tap T have landed and the stage still holds tap T,
#   * a loader refills the stage of tap T - 1 (with tap T + 1) only when ALL eight consumers have read tap T - 1.
# `shared_counter=True` is the first draft (one `ready` and one `done` counter, waits on 4 (T + 1) / 8 T): the starved
# schedules break it, which is why the kernel publishes one word per wave.
import random


def _simulate_handover(ntaps, shared_counter, pick):
    NL, NC = 4, 8
    ready = [0] * NL          # per-loader progress words (or summed, for the counter form)
    done = [0] * NC
    landed = [[False] * ntaps for _ in range(NL)]      # loader lw's pieces of tap T are in LDS
    stage_tap = [[None, None] for _ in range(NL)]      # what each loader's quarter of weight stage s holds
    read_done = [[False] * ntaps for _ in range(NC)]
    lpc, ltap = ["land"] * NL, [0] * NL                # loader program counter / current tap
    cpc, ctap = ["wait"] * NC, [0] * NC
    for lw in range(NL):                               # prologue: tap 0 issued into stage 0
        stage_tap[lw][0] = 0

    def ready_ok(T):       # consumer may start tap T (0-based)
        return sum(ready) >= NL * (T + 1) if shared_counter else all(r >= T + 1 for r in ready)

    def done_ok(T):        # loader at tap T may refill the stage of tap T - 1
        return sum(done) >= NC * T if shared_counter else all(d >= T for d in done)

    steps = 0
    while any(t < ntaps for t in ltap) or any(t < ntaps for t in ctap):
        steps += 1
        assert steps < 100000, "deadlock"
        runnable = []
        for lw in range(NL):
            if ltap[lw] < ntaps and (lpc[lw] == "land" or done_ok(ltap[lw])):
                runnable.append(("L", lw))
        for w in range(NC):
            if ctap[w] < ntaps and (cpc[w] == "read" or ready_ok(ctap[w])):
                runnable.append(("C", w))
        assert runnable, "deadlock"
        kind, i = pick(runnable)
        if kind == "L":
            T = ltap[i]
            if lpc[i] == "land":                        # vmcnt(0): this loader's pieces of tap T are in LDS; publish
                assert stage_tap[i][T % 2] == T
                landed[i][T] = True
                ready[i] = T + 1
                lpc[i] = "refill"
            else:                                       # passed the wait on `done`: refill the other stage with tap T + 1
                if T >= 1:
                    bad = [w for w in range(NC) if not read_done[w][T - 1]]
                    if bad:
                        return f"loader {i} overwrites tap {T - 1} while consumers {bad} still read it"
                if T + 1 < ntaps:
                    stage_tap[i][(T + 1) % 2] = T + 1
                lpc[i] = "land"
                ltap[i] += 1
        else:
            T = ctap[i]
            if cpc[i] == "wait":                        # passed the wait on `ready`: read the fragments of tap T
                for lw in range(NL):
                    if not landed[lw][T] or stage_tap[lw][T % 2] != T:
                        return f"consumer {i} reads tap {T} before loader {lw}'s pieces landed / after they were overwritten"
                cpc[i] = "read"
            else:                                       # lgkmcnt(0): fragments in registers; publish
                read_done[i][T] = True
                done[i] = T + 1
                cpc[i] = "wait"
                ctap[i] += 1
    return None


def _starve(victim):
    def pick(runnable):
        others = [r for r in runnable if r != victim]
        return others[0] if others else runnable[0]
    return pick


def test_patch_handover_per_wave_words_hold_under_any_schedule():
    rng = random.Random(7)
    for ntaps in (1, 2, 9, 27):
        for _ in range(200):
            assert _simulate_handover(ntaps, False, lambda r: rng.choice(r)) is None
        for victim in [("L", 3), ("L", 0), ("C", 7), ("C", 0)]:
            assert _simulate_handover(ntaps, False, _starve(victim)) is None


def test_patch_handover_shared_counters_race_when_one_wave_lags():
    # the first draft: three loaders two hand-overs ahead make `ready >= 4 (T + 1)` true without the fourth
    assert _simulate_handover(9, True, _starve(("L", 3))) is not None
    assert _simulate_handover(9, True, _starve(("C", 7))) is not None
