#!/bin/bash
# Round-end evidence run on the GPU box: kernel-trace stats + the three PMC passes of the bench command, the default
# bench line, and the dev tools.  Everything lands under gpurun_out/$1/ ; copy the summaries into profiles/.
set -o pipefail
OUT=gpurun_out/${1:-round}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile --no-duplicate-leg"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-duplicate-leg > $OUT/stats.log 2>&1
cp $OUT/stats/*/*kernel_stats.csv $OUT/kernel_stats.csv
# per-launch time of the dominant kernels against the previous round's committed profile (fails loudly on > 4 %).  The guard
# compares per-instantiation AVERAGES, so it runs on the step in the form the committed profile was taken in: every block on
# both CFG items (--no-shared-prefix); the product default (shared prefix: the first launches of several kernels are half
# size) is the stats file above.  The reference file's name is recorded in the output; a FAIL fails this script.
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_dup -- python3 bench.py --steps 2 --warmup 1 --cpu-frames 0 --no-profile --no-shared-prefix --rehearsal > $OUT/stats_dup.log 2>&1
cp $OUT/stats_dup/*/*kernel_stats.csv $OUT/kernel_stats_full_duplicate.csv
# like with like: the newest committed FULL-DUPLICATE stats; the plain stats only where a round has none
PREV=${PERF_GUARD_REF:-$(ls profiles/r0*_kernel_stats_full_duplicate.csv 2>/dev/null | sort | tail -1)}
[ -z "$PREV" ] && PREV=$(ls profiles/r0*_kernel_stats.csv | grep -v full_duplicate | sort | tail -1)
PREV_BENCH=$(echo $PREV | sed -E 's/_kernel_stats(_full_duplicate)?\.csv/_bench.json/')
# this box's MFMA probe for the guard's box factor: a quick bench line (one step, no CPU leg)
python3 bench.py --steps 1 --warmup 1 --cpu-frames 0 --no-profile --no-duplicate-leg > $OUT/bench_probe.json 2> /dev/null
GUARD_RC=0
{ echo "perf_guard reference: $PREV"; python3 tools/perf_guard.py $OUT/kernel_stats_full_duplicate.csv $PREV --bench-new $OUT/bench_probe.json --bench-ref $PREV_BENCH; } | tee $OUT/perf_guard.txt || GUARD_RC=1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc/fetch -- $B > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc/write -- $B > $OUT/pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc/mfma -- $B > $OUT/pmc_mfma.log 2>&1
python tools/pmc_summary.py $OUT/pmc/fetch $OUT/pmc/write $OUT/pmc/mfma --json $OUT/pmc_current.json > $OUT/pmc_summary.md
rm -rf $OUT/stats $OUT/stats_dup $OUT/pmc
# the committed bench line of the round is taken on the final sources WITH this run's PMC profile in place (VERDICT r5 7d:
# r05_bench_20steps.json predated the final PMC pass and carried traffic: null); copy the same file into profiles/ afterwards
cp $OUT/pmc_current.json profiles/pmc_current.json
echo "== bench (default flags)"; python bench.py > $OUT/bench.json 2> $OUT/bench.err; cut -c1-400 $OUT/bench.json
echo "== bench (20 steps)"; python bench.py --steps 20 --warmup 3 > $OUT/bench_20steps.json 2> $OUT/bench_20steps.err; cut -c1-200 $OUT/bench_20steps.json
echo "== bench (--profile-all)"; python bench.py --steps 5 --warmup 2 --profile-all --shapes 40 --cpu-frames 0 > $OUT/bench_profile_all.json 2> $OUT/bench_profile_all.err
echo "== tools"; { echo "## tools/block_profile.py"; python tools/block_profile.py 2>&1 | grep -v amdgpu.ids; echo; echo "## tools/attn_bench.py"; python tools/attn_bench.py 2>&1 | grep -v amdgpu.ids; echo; echo "## tools/k7_bench.py"; python tools/k7_bench.py 2>&1 | grep -v amdgpu.ids; echo; echo "## tools/gemm_bench.py"; python tools/gemm_bench.py 2>&1 | grep -v amdgpu.ids; echo; echo "## tools/k5_bench.py"; python tools/k5_bench.py 2>&1 | grep -v amdgpu.ids; echo; echo "## tools/k8p_bench.py"; python tools/k8p_bench.py 2>&1 | grep -v amdgpu.ids; } > $OUT/tools.txt
tail -5 $OUT/tools.txt
if [ $GUARD_RC -ne 0 ]; then echo "round_profile: perf_guard FAILED (see $OUT/perf_guard.txt)"; exit 1; fi
