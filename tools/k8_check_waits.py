#!/usr/bin/env python3
"""Dev tool (CPU only): checks the counted `s_waitcnt vmcnt(N)` of csrc/ff_fused.hip's chunk loop against the ISA hipcc
emitted.  A step's wait is N = the vector-memory instructions younger than the last weight-unit DMA piece the next step
needs (`inflight()`); N may be smaller than the true count (it then waits for more than it must) but never larger, and
that only holds if every load the source counts is exactly one instruction, in the segment the model puts it in.
Usage:  hipcc ... -save-temps=obj -c ff_fused.hip ;  k8_check_waits.py ff_fused-hip-amdgcn-...s"""
import re
import sys

KM, NCGF, NCG, NU = 5, 2, 3, 5
CSTEPS, UPC = KM + NCG, 15


def ubl(r): return 2 * r if r <= KM else 2 * KM + (2 * (r - KM) if r - KM <= NCGF else 2 * NCGF + (r - KM - NCGF))
def ub(s): return UPC * (s // CSTEPS) + ubl(s % CSTEPS)
def hm(s): return NU if s < 0 else ub(s + 1) + NU
def n_b1(s): return 8 if s % CSTEPS == KM else 0
def inflight(s): return 2 * (hm(s - 1) - ub(s + 2)) + n_b1(s)


# ---- PO: the tail (proj_out) — tattn2's output-projection schedule, 15 steps, 25 units, then the next tile's chunk 0
TSTEPS, TUNITS, TRS0 = NCG * KM, 2 * KM * NCGF + KM * (NCG - NCGF), KM


def nt(c): return 8 if c < NCGF else 4


def tub(t):
    if t >= TSTEPS:
        return TUNITS + ubl(t - TSTEPS)
    c, m = divmod(t, KM)
    return 2 * KM * c + 2 * m if c < NCGF else 2 * KM * NCGF + (t - NCGF * KM)


def thm(t): return NU if t < 0 else tub(t + 1) + NU
def txp(t): return 15 if t in (TRS0, TRS0 + 1) else 0
def tn_bias(t): return next((nt(c) for c in range(1, NCG) if t == c * KM - 1), 0)
def tn_res(t): return next((3 * nt(c) // 2 for c in range(NCG) if t == c * KM + 1), 0)
def tn_st(t): return next((3 * nt(c) // 2 for c in range(NCG) if t == c * KM + KM - 1), 0)
def tyounger(t): return 0 if t == 0 else txp(t - 1) + tn_st(t - 1) + tn_bias(t) + tn_res(t)
def tinflight(t): return 2 * (thm(t - 1) - tub(t + 2)) + tyounger(t)


def segments(textpart):
    segs, cur = [], dict(dma=0, ld=0, st=0, waits=[])
    for ln in textpart.split("\n"):
        t = ln.strip()
        if t.startswith("s_waitcnt") and "vmcnt" in t:
            cur["waits"].append(int(re.search(r"vmcnt\((\d+)\)", t).group(1)))
        if t.startswith("global_load_lds"):
            cur["dma"] += 1
        elif t.startswith("global_load"):
            cur["ld"] += 1
        elif t.startswith(("global_store", "flat_", "scratch_", "buffer_")):
            cur["st"] += 1
        if t.startswith("s_barrier"):
            segs.append(cur)
            cur = dict(dma=0, ld=0, st=0, waits=[])
    segs.append(cur)
    return segs


def check_kernel(text, label, po):
    m = re.search(r"^(_ZN\S*ff_fused_kernelILi320E" + label + r"\S*):", text, re.M)
    if not m:
        print("kernel", label, "not found")
        return 1
    body = text[m.end():text.index("s_endpgm", m.end())]
    blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
    big = max(blocks, key=lambda b: b.count("v_mfma"))
    # the chunk loop = the text up to the first branch behind its 360 MFMAs; with PO the tail follows it in the same block
    pos, n = 0, 0
    for mm in re.finditer(r"v_mfma", big):
        n += 1
        if n == 360:
            pos = mm.end()
            break
    cut = big.index("s_cbranch", pos)
    loop, rest = big[:cut], big[cut:]
    segs = segments(loop)
    bad = 0
    if len(segs) != CSTEPS + 1:
        print(label, ": expected", CSTEPS, "barriers in the chunk loop, found", len(segs) - 1)
        return 1
    for s in range(CSTEPS):
        dma = 2 * (hm(s - 1) - hm(s - 2)) if s > 0 else 0
        exp = dict(dma=dma, ld=n_b1(s), st=0)
        got = dict(dma=segs[s]["dma"], ld=segs[s]["ld"], st=segs[s]["st"])
        w = segs[s]["waits"]
        ok = got == exp and w and w[-1] == inflight(s) and all(x <= inflight(s) for x in w)
        if not ok:
            print(f"  {label} step {s}: ISA {got} waits {w}   model {exp} wait {inflight(s)}")
            bad += 1
    tail = segs[CSTEPS]
    if tail["dma"] != 2 * (hm(CSTEPS - 1) - hm(CSTEPS - 2)) or tail["ld"] or tail["st"]:
        print("  after the last barrier of the chunk loop:", tail)
        bad += 1
    print(f"{label} chunk loop: {CSTEPS} steps, {bad} mismatching")
    if not po:
        return bad
    tsegs = segments(rest)
    if len(tsegs) != TSTEPS + 1:
        print(label, ": expected", TSTEPS, "barriers in the tail, found", len(tsegs) - 1)
        return bad + 1
    tbad = 0
    for t in range(TSTEPS):
        # segment t: from barrier t-1 (exclusive) to barrier t: the pieces issued after barrier t-1 (weights + row pieces), the stores
        # of the epilogue at the end of step t-1, the loads at the top of step t (t = 0: y's 30 residual loads + 8 bias loads)
        exp = dict(dma=(2 * (thm(t - 1) - thm(t - 2)) + txp(t - 1)) if t > 0 else 0,
                   ld=(tn_bias(t) + tn_res(t)) if t > 0 else 38, st=tn_st(t - 1) if t > 0 else 0)
        got = dict(dma=tsegs[t]["dma"], ld=tsegs[t]["ld"], st=tsegs[t]["st"])
        w = tsegs[t]["waits"]
        # (t = 0: the compiler's own partial waits for y's residual loads sit in this segment; the LAST wait gates the barrier)
        ok = got == exp and w and w[-1] == tinflight(t) and (t == 0 or all(x <= tinflight(t) for x in w))
        if not ok:
            print(f"  {label} tail step {t}: ISA {got} waits {w}   model {exp} wait {tinflight(t)}")
            tbad += 1
    last = tsegs[TSTEPS]
    if last["dma"] != 2 * (thm(TSTEPS - 1) - thm(TSTEPS - 2)) or last["st"] != tn_st(TSTEPS - 1):
        print("  after the last barrier of the tail:", last)
        tbad += 1
    print(f"{label} tail: {TSTEPS} steps, {tbad} mismatching")
    return bad + tbad


def main(path):
    text = open(path).read()
    return check_kernel(text, "Lb0E", False) + check_kernel(text, "Lb1E", True)


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
