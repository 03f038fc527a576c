"""Discrete model of the experimental 3-stage igemm schedule (mofa_video_amd/csrc/igemm_ring3.inc): replays one wave's
program order (waits, barriers, refills, fragment reads, epilogue operations, bias loads) for a walk of several tiles with
an in-order VMEM queue and random landing times, and checks the invariants the kernel relies on:
  1. when a K step reads ring buffer b, the refill that carries that step's data has retired (per the wait in front of it),
  2. a refill never targets a buffer that is read in the same step or holds a not-yet-consumed stage,
  3. the epilogue's slab buffer has no refill in flight,
  4. the queue never exceeds the 6-bit vmcnt range.
    python tools/ring3_schedule_model.py        (pure Python, no GPU)"""
import random

OPS = 6


def run(nk, ntiles, epi_ops, bias_ops, exact, seed, compiler_drain=True):
    rng = random.Random(seed)
    q = []                      # in-order VMEM queue: (kind, tile, stage, buf)
    buf_holds = [None] * 3      # (tile, stage) whose data the buffer holds or is being refilled with
    landed = set()

    def issue(tile, stage, buf, real=True):
        assert buf_holds[buf] is None or buf_holds[buf] in consumed, f"refill of buffer {buf} over live {buf_holds[buf]}"
        buf_holds[buf] = (tile, stage) if real else ("dummy", tile, stage)
        for _ in range(OPS):
            q.append(("dma", tile, stage, buf))
        assert len(q) <= 63, len(q)

    def wait_le(n):
        n -= n % 2                                          # wait_vmcnt_le_fine rounds down to even
        while len(q) > n:
            k = q.pop(0)
            if k[0] == "dma":
                landed.add(k[1:])
        # anything still queued may or may not have landed: model the adversarial case (not landed)

    consumed = set()

    def read(tile, stage, buf):
        assert buf_holds[buf] == (tile, stage), f"tile {tile} step {stage} reads buffer {buf} holding {buf_holds[buf]}"
        pending = [k for k in q if k[0] == "dma" and k[1:3] == (tile, stage)]
        assert not pending, f"tile {tile} step {stage}: {len(pending)} refill pieces still in flight"
        consumed.add((tile, stage))

    nb = lambda b: 0 if b == 2 else b + 1
    for _ in range(bias_ops):
        q.append(("bias", 0))
    issue(0, 0, 0)
    issue(0, 1, 1)
    cb, allow = 0, OPS
    for t in range(ntiles):
        has_next = t + 1 < ntiles
        cur, nxt = cb, nb(nb(cb))
        for ks in range(nk):
            if ks < 2:
                wait_le(allow)
            else:
                wait_le(OPS)
            # barrier: every wave has finished the previous step's reads
            if ks == 1 and compiler_drain:
                wait_le(0)                                   # hipcc's vmcnt(0) at the bias touch (seen in the ISA)
            assert nxt != cur
            if ks < nk - 2:
                issue(t, ks + 2, nxt)
            else:
                issue(t + 1, ks - (nk - 2), nxt, real=has_next)
            read(t, ks, cur)
            last = cur
            cur, nxt = nb(cur), nb(nxt)
        # epilogue: slabs in `last`
        inflight_bufs = {k[3] for k in q if k[0] == "dma"}
        assert last not in inflight_bufs, (last, inflight_bufs)
        n_epi = epi_ops if exact else rng.randint(0, epi_ops)  # ragged tiles issue fewer memory operations
        for _ in range(n_epi):
            q.append(("epi", t))
        assert len(q) <= 63, len(q)
        if not has_next:
            break
        for _ in range(bias_ops):
            q.append(("bias", t + 1))
        allow = OPS + ((epi_ops + bias_ops) if exact else 0)
        cb = nb(last)
    wait_le(0)
    return True


if __name__ == "__main__":
    n = 0
    for nk in (4, 5, 6, 7, 9, 20, 40):
        for epi_ops, bias_ops in ((8, 2), (8, 0), (16, 2), (24, 2), (40, 2), (4, 4)):
            for exact in (True, False):
                for seed in range(3):
                    for drain in (True, False):              # the counted waits must hold without hipcc's extra drain too
                        run(nk, 5, epi_ops, bias_ops, exact, seed, drain)
                        n += 1
    print(f"ring3 schedule model: {n} configurations, all invariants hold")
