#!/usr/bin/env python3
"""Per-launch time guard for the dominant kernels (an 11 % regression of the 3x3-conv kernel went unnoticed for
half a round).  Compares the rocprofv3 --kernel-trace --stats averages of a fresh run with a committed profile and FAILS
(exit 1) when a guarded kernel's average launch got more than --tol (4 %) slower than the BOX, or 2 x tol outright.  Boxes
of the pool differ by several per cent as a whole.  Two box factors: the median ratio of the guarded kernels (rounds 4-5) and,
when both bench lines carry one, the ratio of the two runs' MFMA PROBES (`box.mfma_probe_tflops`, round 6: a fixed instruction
stream measured in the bench process — independent of the kernels being judged); a kernel fails when it is slower than the box
by BOTH (the probe alone mispredicted a box by ~3 % in round 6).

    python tools/perf_guard.py gpurun_out/rNN/kernel_stats.csv profiles/r03_kernel_stats.csv [--tol 0.04]
                               [--bench-new gpurun_out/rNN/bench.json --bench-ref profiles/r0M_bench.json]"""
import argparse
import csv
import json
import sys

GUARDED = (
    "gemm_kernel<256, 320, 4, 2, 1, false, true, 0>",      # 3x3 conv, levels 0-2 (the dominant kernel)
    "gemm_kernel<256, 320, 4, 2, 2, false, true, 0>",      # temporal conv
    "gemm_kernel<256, 320, 4, 2, 0, false, false, 0>",     # tiled Linear
    "flash_attn_kernel<2, false, true>",                   # spatial self-attention
    "tattn2_kernel<320, 24>",                              # K7, second design
    "ff_fused_kernel<320>",                                # K8 (rounds 3-4)
    "ff_fused_kernel<320, true>",                          # K8 with proj_out as its tail (round 5)
    "xattn_kernel<320>",                                   # K5 (round 5)
    "conv3x3_gn_kernel",                                   # K1 (level-0 3x3 convolutions, round 4)
    "tconv_gn_kernel<12>",                                 # K3 (level-0 temporal convolutions at 24 frames, round 4)
)


def load(path):
    rows = {}
    for r in csv.DictReader(open(path)):
        rows[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]))
    return rows


def probe_of(path):
    """box.mfma_probe_tflops of a bench line file (the last JSON line in it), or None."""
    if not path:
        return None
    try:
        lines = [ln for ln in open(path).read().splitlines() if ln.lstrip().startswith("{")]
        return float(json.loads(lines[-1])["box"]["mfma_probe_tflops"])
    except (OSError, ValueError, KeyError, IndexError, TypeError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("new")
    ap.add_argument("ref")
    ap.add_argument("--tol", type=float, default=0.04)
    ap.add_argument("--bench-new", default=None, help="bench line (JSON file) of the new run: its box.mfma_probe_tflops")
    ap.add_argument("--bench-ref", default=None, help="bench line of the reference run")
    a = ap.parse_args()
    new, ref = load(a.new), load(a.ref)
    rows = []
    for g in GUARDED:
        kn = [k for k in new if g in k]
        kr = [k for k in ref if g in k]
        if not kn or not kr:
            print(f"  {g:58s} not in {'the new run' if not kn else 'the reference'}: skipped")
            continue
        (cn, tn), (cr, tr) = new[kn[0]], ref[kr[0]]
        same_mix = cn % cr == 0 or cr % cn == 0       # else: a different mix of shapes behind one instantiation
        rows.append((g, tn, tr, cn, cr, same_mix))
    # the boxes of the pool differ by a few per cent as a whole: the BOX factor is the median ratio over the guarded
    # kernels with an unchanged shape mix; a kernel fails when it is > tol slower than the box, or > 2 tol slower outright
    ratios = sorted(tn / tr for _, tn, tr, _, _, ok in rows if ok)
    box = ratios[len(ratios) // 2] if ratios else 1.0
    probes = [probe_of(a.bench_new), probe_of(a.bench_ref)]
    box_p = None
    if all(probes):
        # A kernel fails only when it is slower than the box by BOTH measures: the probe is independent of the kernels judged
        # but is one instruction stream (round 6: two boxes whose probes said 0.990 ran every kernel 1-3.6 % apart the other
        # way — HBM and clocks under mixed load differ too); the median moves with a uniform regression.  Either alone misjudges.
        box_p = probes[1] / probes[0]
        print(f"  box factor from the MFMA probes (reference {probes[1]:.1f} / new {probes[0]:.1f} TFLOP/s): {box_p:.3f};  "
              f"median ratio of the guarded kernels: {box:.3f}")
    else:
        print(f"  box factor (median ratio of the guarded kernels; no MFMA probe in {'either' if not any(probes) else 'one'} bench line): {box:.3f}")
    bad = 0
    for g, tn, tr, cn, cr, ok in rows:
        raw, rel = tn / tr - 1.0, tn / tr / box - 1.0
        rel_p = tn / tr / box_p - 1.0 if box_p else rel
        fail = ok and ((rel > a.tol and rel_p > a.tol) or raw > 2 * a.tol)
        bad += fail
        note = "" if ok else f"  (calls {cn} vs {cr}: shape mix differs, not judged)"
        vs_p = f"  vs probe {100 * rel_p:+6.1f} %" if box_p else ""
        print(f"  {g:58s} {tn / 1e3:9.1f} us vs {tr / 1e3:9.1f} us  {100 * raw:+6.1f} %  vs box {100 * rel:+6.1f} %{vs_p}  {'FAIL' if fail else 'ok'}{note}")
    if bad:
        print(f"perf_guard: {bad} guarded kernel(s) more than {100 * a.tol:.0f} % slower than {a.ref} (box-normalised; or {200 * a.tol:.0f} % outright)")
        sys.exit(1)
    print("perf_guard: ok")


if __name__ == "__main__":
    main()
