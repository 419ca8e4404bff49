#!/usr/bin/env python3
"""Copy the outputs of tools/round_profile.sh (gpurun_out/<tag>/) into profiles/ under the round's names.
    python tools/install_profile.py gpurun_out/r02c r02"""
import json
import os
import shutil
import sys

src, rnd = sys.argv[1], sys.argv[2]
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
shutil.copy(os.path.join(src, "pmc_current.json"), os.path.join(P, "pmc_current.json"))
shutil.copy(os.path.join(src, "pmc_current.json"), os.path.join(P, f"{rnd}_pmc.json"))
shutil.copy(os.path.join(src, "kernel_stats.csv"), os.path.join(P, f"{rnd}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "tools.txt"), os.path.join(P, f"{rnd}_tools.txt"))
meta = json.load(open(os.path.join(src, "pmc_current.json")))["_meta"]
per_fw = meta["hbm_bytes_all_kernels"] / meta["forwards"] / 1e9
hdr = f"""# {rnd} — PMC summary: `python bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile` (2 forwards), 1x MI355X

Separate rocprofv3 --pmc passes (FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE), merged by tools/pmc_summary.py (tools/round_profile.sh);
per-launch bytes in pmc_current.json (= {rnd}_pmc.json), which records the hash of the kernel sources it was taken on; bench.py reports
`roofline.traffic` from it only while that hash matches.  FETCH_SIZE doubled per MI355X_MICROARCH.md.  Sum over all kernels:
{per_fw:.0f} GB of HBM traffic per forward (r01: 445).  Kernel sources {meta['source_sha']}.

"""
open(os.path.join(P, f"{rnd}_pmc_summary.md"), "w").write(hdr + open(os.path.join(src, "pmc_summary.md")).read())
print(f"installed {src} as profiles/{rnd}_*  (sources {meta['source_sha']}, {per_fw:.0f} GB per forward)")
