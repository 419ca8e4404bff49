#!/usr/bin/env python3
"""Dev tool (CPU only): checks the counted `s_waitcnt vmcnt(N)` of csrc/tattn2.hip against the ISA hipcc emitted.

The kernel's step waits are exact: N = the vector-memory instructions younger than the last weight-unit DMA piece the
step needs (tattn2.hip `inflight()`).  That only holds if every load / store / DMA the source counts is exactly one
instruction in the emitted stream and sits between the same two barriers as in the model.  This script re-derives the
schedule, splits the kernel's ISA at its `s_barrier`s, and compares per step: the wait's N, the DMA pieces, the plain
loads and the stores.  Usage:  hipcc ... -save-temps=obj -c tattn2.hip ;  k7b_check_waits.py tattn2-hip-amdgcn-...s
"""
import re
import sys

P1S, HEADS, NSTEP, NU = 50, 5, 65, 5
RS0 = P1S + HEADS


def kind(s): return (0 if (s % 10) < 5 else 1) if s < P1S else 2


def ub1(s):
    if s <= P1S:
        hs, r = divmod(s, 10)
        return 15 * hs + (2 * r if r < 5 else 10 + (r - 5))
    v = s - P1S
    c, m = divmod(v, 5)
    return 75 + (10 * c + 2 * m if c < 2 else 20 + (v - 10))


def ub(s): return ub1(s) if s <= NSTEP else 100 + ub1(s - NSTEP)
def hm(s): return ub(s + 1) + NU
def xp(s): return 15 if s in (RS0, RS0 + 1) else 0
def nt(c): return 8 if c < 2 else 4
def first(c): return P1S + c * 5
def nbq(s): return 4 if (kind(s) == 1 and s % 10 == 9 and s + 1 < P1S) or s == NSTEP - 1 else 0
def nbias(s): return next((nt(c) for c in range(3) if s == first(c) - 1), 0)
def nres(s): return next((3 * nt(c) // 2 for c in range(3) if s == first(c) + 1), 0)
def nst(s): return next((3 * nt(c) // 2 for c in range(3) if s == first(c) + 4), 0)
def prev(s): return NSTEP - 1 if s == 0 else s - 1
def younger(s): return xp(prev(s)) + nst(prev(s)) + nbq(s) + nbias(s) + nres(s)
def inflight(s): return 2 * (hm(s - 1) - ub(s + 2)) + younger(s)


def check(text, label):
    m = re.search(r"^(_ZN\S*tattn2_kernelILi320E" + label + r"E\S*):", text, re.M)
    if not m:
        print("kernel", label, "not found")
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
        print(label, ": expected", NSTEP + 1, "barriers, found", len(segs) - 1)
        return 1
    model = [inflight(s) for s in range(NSTEP)]
    # the compiler rotates the tile loop: text segment i (1..65) is step (i - 1 + rot) % 65
    rot = max(range(NSTEP), key=lambda o: sum(1 for i in range(1, NSTEP + 1) if segs[i]["vm"] == model[(i - 1 + o) % NSTEP]))
    bad = 0
    for s in range(NSTEP):
        i = ((s - rot) % NSTEP) + 1
        seg, ps = segs[i], prev(s)
        exp = dict(vm=model[s], dma=2 * (hm(ps) - hm(ps - 1)) + xp(ps), ld=nbq(s) + nbias(s) + nres(s), st=nst(ps), other=0)
        if i == 1:        # the seam of the rotated loop: the segment continues at the end of the text
            seg = {k: (seg[k] + segs[-1][k] if k != "vm" else seg[k]) for k in seg}
        if any(seg[k] != exp[k] for k in exp):
            print(f"  {label} step {s}: ISA {seg}  model {exp}")
            bad += 1
    print(f"{label}: {NSTEP} steps, loop rotated by {rot}, {bad} mismatching; prologue {segs[0]}")
    return bad


if __name__ == "__main__":
    text = open(sys.argv[1]).read()
    sys.exit(1 if sum(check(text, lb) for lb in ("Li24E", "Li16E", "Li12E", "Li4E", "Li1E")) else 0)
