"""CPU suite: the bench line's contract (fields the driver and the judge read), checked on the line committed from
the round-end profiled run (profiles/r01_bench_round_end.json) and on bench.py's own helpers."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_committed_bench_line_has_the_contract_fields():
    d = json.load(open(os.path.join(ROOT, "profiles", "r01_bench_round_end.json")))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["unit"] == "steps/s" and d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f16" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]              # N = 1: one 24-frame window per step
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3
    assert r["traffic"] is None or r["traffic"] > 0
    # achieved = algorithmic flops per launch / average launch time (both in the line)
    assert abs(r["achieved"] - r["algorithmic_gflop_per_launch"] / r["avg_launch_ms"]) < 0.01 * r["achieved"]


def test_pmc_traffic_is_reported_only_for_the_build_that_was_profiled(tmp_path, monkeypatch):
    """`roofline.traffic` comes from a committed PMC profile; a profile taken on other kernel sources must give
    None (with a note), never a stale number (round-1 finding)."""
    sys.path.insert(0, ROOT)
    import bench
    import vdx  # noqa: F401
    from vdx._lib import source_sha
    k = "gemm_kernel<256, 320, 4, 2, 1, false, true, 0>"
    prof = {k: {"launches": 70, "hbm_read_bytes_per_launch": 6.0e8, "hbm_write_bytes_per_launch": 1.7e8, "mfma_busy": 0.5},
            "_meta": {"source_sha": source_sha(), "forwards": 2, "hbm_bytes_all_kernels": 8.0e11, "kernels": [k]}}
    f = tmp_path / "pmc.json"
    f.write_text(json.dumps(prof))
    monkeypatch.setattr(bench, "PMC_JSON", str(f))
    t, per_step, note = bench.pmc_traffic(k)
    assert t == 770000000 and per_step == 400000000000 and note is None
    assert bench.pmc_traffic("no_such_kernel")[0] is None
    prof["_meta"]["source_sha"] = "0000000000000000"
    f.write_text(json.dumps(prof))
    t, per_step, note = bench.pmc_traffic(k)
    assert t is None and per_step is None and "0000000000000000" in note
    monkeypatch.setattr(bench, "PMC_JSON", str(tmp_path / "missing.json"))
    assert bench.pmc_traffic(k)[0] is None


def test_bench_selects_the_baseline_configuration_of_the_world_size():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.select_config(1) == ("cfg2", 24, "mono")
    assert bench.select_config(2) == ("cfg3", 24, "fsdp")
    assert bench.select_config(4) == ("cfg4", 48, "hybrid")
    assert bench.select_config(8) == ("cfg5", 96, "hybrid_ctx")
    assert bench.select_config(3) == ("generic-3", 36, "hybrid_ctx")
    assert bench.select_config(1, "cfg5") == ("cfg5", 96, "hybrid_ctx")
    # the planner on those: SURVEY a1 known answers
    from vdx.planner import plan
    assert plan(24, 2, 0, 4, no_chunking=True).ranges == ((0, 24), (0, 24))
    assert plan(48, 4, 0, 4).ranges == ((0, 16), (12, 28), (24, 40), (36, 48))
    assert plan(96, 8, 0, 4).ranges[-2:] == ((72, 88), (84, 96)) and len(plan(96, 8, 0, 4).ranges) == 8


def test_perf_guard_flags_a_slower_dominant_kernel(tmp_path):
    """tools/perf_guard.py: a guarded kernel whose average launch is > 4 % slower than the committed profile fails the run."""
    import subprocess
    import sys
    hdr = '"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
    row = '"void {}(X)",{},1,{},1.0,1,1,0\n'
    ref, new_ok, new_bad = tmp_path / "ref.csv", tmp_path / "ok.csv", tmp_path / "bad.csv"
    k1, k2 = "gemm_kernel<256, 320, 4, 2, 1, false, true, 0>", "flash_attn_kernel<2, false, true>"
    ref.write_text(hdr + row.format(k1, 105, 900000.0) + row.format(k2, 45, 2098000.0))
    new_ok.write_text(hdr + row.format(k1, 105, 925000.0) + row.format(k2, 45, 2050000.0))
    new_bad.write_text(hdr + row.format(k1, 105, 995000.0) + row.format(k2, 45, 2050000.0))
    tool = os.path.join(ROOT, "tools", "perf_guard.py")
    ok = subprocess.run([sys.executable, tool, str(new_ok), str(ref)], capture_output=True, text=True)
    bad = subprocess.run([sys.executable, tool, str(new_bad), str(ref)], capture_output=True, text=True)
    assert ok.returncode == 0 and "perf_guard: ok" in ok.stdout, ok.stdout
    assert bad.returncode == 1 and "FAIL" in bad.stdout and "+10.6 %" in bad.stdout, bad.stdout
    # a box that is 5 % slower as a whole is not a regression of any kernel
    slow = tmp_path / "slow.csv"
    slow.write_text(hdr + row.format(k1, 105, 945000.0) + row.format(k2, 45, 2203000.0))
    r = subprocess.run([sys.executable, tool, str(slow), str(ref)], capture_output=True, text=True)
    assert r.returncode == 0 and "box factor" in r.stdout, r.stdout


def test_bare_bench_gpus_n_launches_itself(tmp_path):
    """`python bench.py --gpus N` with no launcher: the parent re-runs itself under torch.distributed.run
    and relays the outcome.  Without a GPU the ranks die in `torch.cuda.set_device`; what is checked here is that they were
    started as N ranks (each names its rank), that the parent's exit code is the launcher's, and that no JSON line is
    invented.  (The two-rank run that completes is the -m gpu test `test_bench_self_launch_two_ranks_on_one_gpu`.)"""
    import subprocess
    sys.path.insert(0, ROOT)
    import bench
    cmd = bench.self_launch_command(["--gpus", "2", "--steps", "1"], 2, 29999)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "2", "--steps", "1"]
    assert cmd[cmd.index("29999") + 1] == os.path.join(ROOT, "bench.py")
    import torch
    if torch.cuda.is_available():
        return
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "local_rank: 1" in r.stderr or "rank      : 1" in r.stderr or "rank: 1" in r.stderr, r.stderr[-2000:]


def test_round6_bench_line_identifies_its_box_and_carries_a_parity_number():
    """the line carries what identifies the box's speed (a fixed MFMA stream, a fixed copy, the shader
    clock over the timed steps) and the HIP-vs-oracle distance at the benchmark's own spatial extent; the long-standing path
    fields carry EXECUTED FLOPs.  Two committed lines from DIFFERENT boxes of the pool (raw step times 3.4 % apart) agree within
    2 % once the step's TFLOP/s is divided by the box's probe."""
    a = json.load(open(os.path.join(ROOT, "profiles", "r06_bench.json")))
    b = json.load(open(os.path.join(ROOT, "profiles", "r06_bench_other_box.json")))
    for d in (a, b):
        box = d["box"]
        for k in ("mfma_probe_tflops", "hbm_probe_gbs", "sclk_mhz_mean", "step_tflops_over_probe"):
            assert k in box, k
        assert 800 < box["mfma_probe_tflops"] < 2500 and 2000 < box["hbm_probe_gbs"] < 8000
        assert box["sclk_mhz_mean"] is None or 500 < box["sclk_mhz_mean"] < 3000
        cb = d["cpu_baseline"]
        assert cb["kind"] == "port" and cb["cores"] >= 1 and 0 < cb["rel_l2_vs_hip"] < 4e-3
        assert d["path_mfma_frac"] == d["path_mfma_frac_executed"] < d["path_mfma_frac_reference_equivalent"]
        assert abs(d["path_tflops_per_gpu"] - d["tflop_per_step_executed"] * 1e3 / d["ms_per_step"]) < 0.01 * d["path_tflops_per_gpu"]
        assert abs(box["step_tflops_over_probe"] - d["path_tflops_per_gpu"] / box["mfma_probe_tflops"]) < 1e-3
        assert d["roofline"]["traffic"] is None or d["roofline"]["traffic"] > 0
    assert abs(a["box"]["mfma_probe_tflops"] / b["box"]["mfma_probe_tflops"] - 1) > 0.02          # really two boxes
    assert abs(a["ms_per_step"] / b["ms_per_step"] - 1) > 0.02
    assert abs(a["box"]["step_tflops_over_probe"] / b["box"]["step_tflops_over_probe"] - 1) < 0.02
    assert a["roofline"]["traffic"] is not None and a["hbm_traffic_bytes_per_step"] > 1e11        # the round's line has its PMC profile
