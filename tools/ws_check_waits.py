#!/usr/bin/env python3
"""Counted-wait check of the weights-stationary GEMM (csrc/gemm_ws.hip) against the emitted ISA.

`Ws::sync(k)` waits with `s_waitcnt vmcnt(n)`, n = the vector-memory operations the wave has issued SINCE the LDS-DMA of the
chunk it is about to read: (PF-1) DMA batches of PPW instructions, PF epilogues of MT stores and — RES + PIPE, round 6 — PF
residual requests of MT loads.  That is only right if the compiled steady-state loop issues exactly those operations per
chunk: hipcc merging two loads, splitting a store or spilling to scratch (scratch traffic counts in vmcnt too) would let a
chunk be read before it has landed — invisible at parity sizes (the K8 bug of round 3).  This tool disassembles the kernels
and checks, per instantiation: no scratch / buffer operation anywhere, and the function's totals of `global_load_lds`, stores and
plain loads against what the program's structure issues (`expected`): a merged, split or duplicated operation shows up there.

    python tools/ws_check_waits.py            (exit 1 on a mismatch; CPU only: hipcc cross-compiles)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "decentralised-verification-and-distributed-execution-of-large-scale-video-diffusion-models_amd", "csrc", "gemm_ws.hip")


def expected(K, NW, ROWS, GEGLU, RES, PIPE, WSET, PF):
    """Totals over the whole function, from the program's structure (Ws::run): PIPE = step 0, a loop body of two steps, a tail
    step and a drain in either tail branch; otherwise one loop of sync / step / drain.  DMA: PF initial batches + one per
    sync site; stores: one epilogue (MT) per step that has a predecessor + per drain; plain loads: the weight slice (2 KS per
    load_weights site, + 2 bias vectors with weight sets, whose steps reload), the bias vector, MT residual rows per site."""
    ppw, mt, ks = ROWS * K * 2 // 1024 // NW, ROWS // 16, K // 32
    steps = 4 if PIPE else 1
    dma = (PF + steps) * ppw
    st = (3 + 2) * mt if PIPE else mt
    res_sites = (steps if PIPE else 1) if RES else 0
    lw_sites = 1 + (steps if WSET else 0)
    ld = lw_sites * (2 * ks + (2 if WSET else 0)) + 1 + res_sites * mt
    return dma, st, ld


def main():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "ws.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-S", "--cuda-device-only", SRC, "-o", out],
                       check=True, capture_output=True)
        text = open(out).read()
    bad = 0
    for m in re.finditer(r"\n(_ZN12_GLOBAL__N_114gemm_ws_kernelILi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELb(\d)ELb(\d)ELb(\d)EEEvNS_3WsPE):.*?\n\.Lfunc_end", text, re.S):
        K, NW, ROWS = (int(m.group(i)) for i in (2, 3, 4))
        GEGLU, RES, PIPE, WSET = (m.group(i) == "1" for i in (5, 6, 7, 8))
        body = m.group(0)
        lds = ROWS * K * 2
        ns = 4 if (K == 640 and 4 * lds + (8192 if GEGLU else 0) + (NW * 64 * 32 if WSET else 0) <= 160 * 1024) else 3     # Ws::NS
        want = expected(K, NW, ROWS, GEGLU, RES, PIPE, WSET, ns - 1)
        scratch = len(re.findall(r"\n\s+(scratch_|buffer_)", body))
        got = (len(re.findall(r"\n\s+global_load_lds", body)), len(re.findall(r"\n\s+global_store", body)),
               len(re.findall(r"\n\s+global_load_dword", body)))
        # hipcc may tail-merge the two drains of the PIPE tail (stores: 4 MT .. 5 MT) and may duplicate a whole step (tail
        # duplication: every count grows by one step's worth) — neither changes what a step issues
        ppw, mt = ROWS * K * 2 // 1024 // NW, ROWS // 16
        per_step = (ppw, mt, (mt if (RES and PIPE) else 0) + ((2 * (K // 32) + 2) if WSET else 0))
        ok = False
        for dup in range(0, 3):
            w = tuple(want[i] + dup * per_step[i] for i in range(3))
            if got[0] == w[0] and got[2] == w[2] and (w[1] - (mt if PIPE else 0) - dup * 0 <= got[1] <= w[1]):
                ok = True
        ok = ok and scratch == 0
        tag = f"K={K} NW={NW} ROWS={ROWS} geglu={int(GEGLU)} res={int(RES)} pipe={int(PIPE)} wset={int(WSET)} ring={ns}"
        print(f"{'ok  ' if ok else 'FAIL'} {tag:66s} DMA {got[0]} (want {want[0]}), stores {got[1]} (want {want[1]}), plain loads {got[2]} "
              f"(want {want[2]}), scratch/buffer ops {scratch}")
        bad += not ok
    if bad:
        print(f"ws_check_waits: {bad} instantiation(s) do not issue what Ws::sync counts")
        sys.exit(1)
    print("ws_check_waits: ok")


if __name__ == "__main__":
    main()
