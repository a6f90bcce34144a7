#!/usr/bin/env python3
"""Random-schedule simulation of k_inverse_flow's hand-off protocol (fpv_kernels.hip): one chain
wave and two i/o waves that meet only through counters.  Loads are modelled adversarially both
ways: they LAND only when the issuing wave's counted wait retires them (so a reader that trusts
a counter too early is caught), and they may OVERWRITE their slot as soon as they are issued (so
a slot refilled before its reader is done is caught).  Checks: no deadlock, the chain reads only
landed residuals of the right superstep, the ring never loses an un-emitted superstep, a plane
slot is never refilled before its superstep is emitted.
  python tools/exp/flow_protocol_sim.py [runs]"""
import random, sys

A, SLOTS = 6, 8

def run(nsteps, cap, nk, seed):
    rnd = random.Random(seed)
    F = {"chain": 0, "res": [0, 0], "out": [0, 0]}
    slot_issued = [None] * SLOTS          # per wave identical: superstep whose residuals were last ISSUED into the slot
    slot_landed = [[None] * SLOTS, [None] * SLOTS]
    ring_emitted = [0, 0]                 # supersteps really emitted per i/o wave
    chain_done = 0
    class IO:
        def __init__(s, w):
            s.w, s.i, s.pc, s.q = w, 0, 0, []   # q: outstanding ops in issue order: ("r", j) / ("l", j) / ("s", j)
            s.issued_r = [None] * SLOTS
            for j in range(min(A, nsteps)):
                s.issued_r[j % SLOTS] = j; slot_landed[w][j % SLOTS] = j   # prologue drained
            s.lslot = [0 if nsteps > 0 else None, None]
            s.lland = [0 if nsteps > 0 else None, None]
        def retire(s, keep):
            while len(s.q) > keep:
                k, j = s.q.pop(0)
                if k == "r": slot_landed[s.w][j % SLOTS] = j
                if k == "l": s.lland[j & 1] = j
        def step(s):   # returns False if blocked
            i = s.i
            if i > nsteps: return None
            if s.pc == 0:
                if i >= 2 and i + A - 1 < nsteps:
                    s.retire(5 * nk); landed = i + A - 1
                else:
                    s.retire(0); landed = min(i + A, nsteps)
                for j in range(landed):   # everything the counter promises must really have landed (or been consumed)
                    if j >= chain_done_ref[0]:
                        assert slot_landed[s.w][j % SLOTS] == j, ("published unlanded residual", s.w, i, j, slot_landed[s.w])
                F["res"][s.w] = landed
                s.pc = 1; return True
            if s.pc == 1:
                if i >= 1:
                    if F["chain"] < i: return False
                    assert s.lland[(i - 1) & 1] == i - 1, ("planes not landed", s.w, i, s.lland)
                    s.q += [("s", i - 1)] * nk
                    ring_emitted[s.w] = i
                    F["out"][s.w] = i
                s.pc = 2; return True
            if s.pc == 2:
                if i + 1 < nsteps:
                    assert ring_emitted[s.w] >= i, "plane slot refilled before its superstep was emitted"
                    s.lslot[(i + 1) & 1] = i + 1; s.q += [("l", i + 1)] * (3 * nk)
                if i + A < nsteps:
                    j = i + A
                    old = s.issued_r[j % SLOTS]
                    assert old is None or old < chain_done_ref[0], ("residual slot overwritten before the chain read it", j, old, chain_done_ref[0])
                    s.issued_r[j % SLOTS] = j; slot_landed[s.w][j % SLOTS] = None; s.q += [("r", j)] * nk
                s.i += 1; s.pc = 0; return True
    chain_done_ref = [0]
    ios = [IO(0), IO(1)]
    t = 0
    def chain_step():
        nonlocal t
        if t >= nsteps: return None
        need_out = max(0, t - cap)
        if not (min(F["res"]) > t and min(F["out"]) >= need_out): return False
        for w in range(2):
            assert slot_landed[w][t % SLOTS] == t, ("chain reads wrong residuals", t, w, slot_landed[w])
            assert ios[w].issued_r[t % SLOTS] == t, ("residual slot already refilled", t, w)
            assert ring_emitted[w] >= t - cap, ("ring overwritten", t, ring_emitted)
        t += 1; chain_done_ref[0] = t; F["chain"] = t
        return True
    agents = [chain_step, ios[0].step, ios[1].step]
    idle = 0
    while True:
        k = rnd.choice([0, 0, 1, 2]) if rnd.random() < 0.5 else rnd.randrange(3)
        r = agents[k]()
        if r: idle = 0
        else:
            idle += 1
            if idle > 200:
                rs = [a() for a in agents]
                if all(x is None for x in rs): return True
                if not any(rs): raise AssertionError(("deadlock", nsteps, cap, t, ios[0].i, ios[1].i, F))
                idle = 0

runs = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 0
for seed in range(runs):
    rnd = random.Random(seed)
    run(rnd.choice([0, 1, 2, 3, 5, 6, 7, 8, 9, 13, 40, 799]), rnd.choice([1, 2, 3, 5]), rnd.choice([1, 2, 3]), seed)
    n += 1
print("flow protocol: %d random schedules ok" % n)
