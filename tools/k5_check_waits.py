#!/usr/bin/env python3
"""Dev tool (CPU only): checks the counted `s_waitcnt vmcnt(N)` of csrc/xattn.hip (K5) against the ISA hipcc emitted — the same
check tools/k7b_check_waits.py makes for tattn2.hip, on K5's schedule (per head: five q steps of one unit, one attention step
of three units; then the output projection of tattn2).  A wait is exact only while every load / store / DMA the source counts
is exactly one instruction in the emitted stream, between the same two barriers as in the model.
Usage:  hipcc ... --cuda-device-only -S xattn.hip -o xattn.s ;  k5_check_waits.py xattn.s"""
import re
import sys

HEADS, KM, KVU, NU = 5, 5, 3, 5
HSTEPS, UPH = KM + 1, KM + KVU
P1S = HEADS * HSTEPS
NSTEP = P1S + 3 * HEADS
RS0 = P1S + HEADS


def kind(s): return (0 if (s % HSTEPS) < KM else 1) if s < P1S else 2


def ub1(s):
    if s <= P1S:
        return UPH * (s // HSTEPS) + (s % HSTEPS)
    v = s - P1S
    c, m = divmod(v, HEADS)
    return UPH * HEADS + (2 * HEADS * c + 2 * m if c < 2 else 4 * HEADS + (v - 2 * HEADS))


NUNITS = ub1(NSTEP)


def ub(s): return ub1(s) if s <= NSTEP else NUNITS + ub1(s - NSTEP)
def hm(s): return ub(s + 1) + NU
def xp(s): return 15 if s in (RS0, RS0 + 1) else 0
def nt(c): return 8 if c < 2 else 4
def first(c): return P1S + c * HEADS
def nbq(s): return 4 if (kind(s) == 1 and s + 1 < P1S) or s == NSTEP - 1 else 0
def nbias(s): return next((nt(c) for c in range(3) if s == first(c) - 1), 0)
def nres(s): return next((3 * nt(c) // 2 for c in range(3) if s == first(c) + 1), 0)
def nst(s): return next((3 * nt(c) // 2 for c in range(3) if s == first(c) + HEADS - 1), 0)
def prev(s): return NSTEP - 1 if s == 0 else s - 1
def younger(s): return xp(prev(s)) + nst(prev(s)) + nbq(s) + nbias(s) + nres(s)
def inflight(s): return 2 * (hm(s - 1) - ub(s + 2)) + younger(s)


def check(text):
    m = re.search(r"^(_ZN\S*xattn_kernelILi320E\S*):", text, re.M)
    if not m:
        print("kernel not found")
        return 1
    body = text[m.end():text.index("s_endpgm", m.end())].split("\n")
    segs, cur, lastwait = [], dict(dma=0, ld=0, st=0, vm=None, other=0), None
    for ln in body:
        t = ln.strip()
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            lastwait = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
        if t.startswith("global_load_lds"):
            cur["dma"] += 1
        elif t.startswith("global_load"):
            cur["ld"] += 1
        elif t.startswith("global_store"):
            cur["st"] += 1
        elif t.startswith(("flat_", "scratch_", "buffer_")):
            cur["other"] += 1
        if t.startswith("s_barrier"):
            cur["vm"] = lastwait
            segs.append(cur)
            cur, lastwait = dict(dma=0, ld=0, st=0, vm=None, other=0), None
    segs.append(cur)
    if len(segs) != NSTEP + 2:
        print("expected", NSTEP + 1, "barriers, found", len(segs) - 1)
        return 1
    model = [inflight(s) for s in range(NSTEP)]
    rot = max(range(NSTEP), key=lambda o: sum(1 for i in range(1, NSTEP + 1) if segs[i]["vm"] == model[(i - 1 + o) % NSTEP]))
    bad = 0
    for s in range(NSTEP):
        i = ((s - rot) % NSTEP) + 1
        seg, ps = segs[i], prev(s)
        exp = dict(vm=model[s], dma=2 * (hm(ps) - hm(ps - 1)) + xp(ps), ld=nbq(s) + nbias(s) + nres(s), st=nst(ps), other=0)
        if i == 1:        # the seam of the rotated loop: the segment continues at the end of the text
            seg = {k: (seg[k] + segs[-1][k] if k != "vm" else seg[k]) for k in seg}
        if any(seg[k] != exp[k] for k in exp):
            print(f"  step {s}: ISA {seg}  model {exp}")
            bad += 1
    print(f"xattn_kernel<320>: {NSTEP} steps, {NUNITS} units, loop rotated by {rot}, {bad} mismatching; prologue {segs[0]}")
    return bad


if __name__ == "__main__":
    sys.exit(1 if check(open(sys.argv[1]).read()) else 0)
