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


def main(path):
    text = open(path).read()
    m = re.search(r"^(_ZN\S*ff_fused_kernel\S*):", text, re.M)
    body = text[m.end():text.index("s_endpgm", m.end())]
    # the chunk loop = the basic block that holds the MFMAs
    blocks = re.split(r"^\.LBB\d+_\d+:.*$", body, flags=re.M)
    loop = max(blocks, key=lambda b: b.count("v_mfma"))
    loop = loop[:loop.index("s_cbranch", loop.rindex("v_mfma"))]          # (the code behind the loop's branch has no label)
    segs, cur = [], dict(dma=0, ld=0, st=0, waits=[])
    for ln in loop.split("\n"):
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
    if len(segs) != CSTEPS + 1:
        print("expected", CSTEPS, "barriers in the chunk loop, found", len(segs) - 1)
        return 1
    bad = 0
    for s in range(CSTEPS):
        # segment s: from barrier s-1 (exclusive) to barrier s: the pieces issued after barrier s-1, the loads at the top of
        # step s, the wait of step s
        dma = 2 * (hm(s - 1) - hm(s - 2)) if s > 0 else 0
        exp = dict(dma=dma, ld=n_b1(s), st=0)
        got = dict(dma=segs[s]["dma"], ld=segs[s]["ld"], st=segs[s]["st"])
        w = segs[s]["waits"]
        ok = got == exp and w and w[-1] == inflight(s) and all(x <= inflight(s) for x in w)
        if not ok:
            print(f"  step {s}: ISA {got} waits {w}   model {exp} wait {inflight(s)}")
            bad += 1
    tail = segs[CSTEPS]
    if tail["dma"] != 2 * (hm(CSTEPS - 1) - hm(CSTEPS - 2)) or tail["ld"] or tail["st"]:
        print("  after the last barrier:", tail)
        bad += 1
    print(f"chunk loop: {CSTEPS} steps, {bad} mismatching")
    return bad


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1]) else 0)
