"""CPU suite: the oracle (and the product's host logic) against what the REFERENCE'S OWN CODE computed when it was
executed in the build container (tests/golden/make_ref_fixtures.py: `fsdp_chunked_coherent.py`, `fsdp_chunked.py`,
`chunk_only.py` run unmodified as `__main__`, world sizes 1-8 over gloo, on stand-in model objects — diffusers stays
absent, so this pins the rows the reference owns: a1 planner + chunk -> rank, a2 noise + slicing, a3 global context,
a4 the `_denoise` call sequence, a7 the FSDP wrap arguments, a8 gather order, a9 ramp blend, the CSV row of a10).

Bit for bit: every comparison below is `==` / `torch.equal`."""
import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from ref_exec_standins import ExactUNet, frames_u8, oracle_unet, text_table  # noqa: E402

EXACT = ("exact_hybrid_ctx_w2", "exact_hybrid_ctx_w3", "exact_hybrid_w4", "exact_hybrid_ctx_w8", "exact_chunk_w1",
         "exact_chunk_only_cfg1", "exact_fsdp_chunked_w2", "exact_fsdp_mode_w2")
ORACLE = ("oracle_hybrid_ctx_w2", "oracle_chunk_only_cfg1")
NUMERIC = EXACT + ORACLE


def planner_rows():
    return json.load(open(os.path.join(GOLD, "ref_exec_planner.json")))


def rule_of(row):
    return "coherent" if row["file"] == "coherent" else "third"


def test_fixture_covers_the_sweep_and_every_baseline_configuration():
    rows = planner_rows()
    assert len(rows) >= 1500 and {r["world"] for r in rows} == {1, 2, 3, 4, 8}
    assert {r["file"] for r in rows} == {"coherent", "fsdp_chunked", "chunk_only"}
    key = lambda r: (r["file"], r["mode"], r["T"], r["world"], r["chunk_size"], r["overlap"])     # noqa: E731
    by = {key(r): r for r in rows}
    # SURVEY §8 a1's hand-executed known answers, now the reference's executed ones
    assert by[("chunk_only", "-", 8, 1, 0, 4)]["ranges"] == [[0, 8], [6, 8]] and by[("chunk_only", "-", 8, 1, 0, 4)]["ov"] == 2       # cfg1
    assert by[("coherent", "fsdp", 24, 2, 0, 4)]["ranges"] == [[0, 24], [0, 24]] and by[("coherent", "fsdp", 24, 2, 0, 4)]["ov"] == 0   # cfg3
    assert by[("coherent", "hybrid", 24, 2, 0, 4)]["ranges"] == [[0, 16], [12, 24]]                                                     # cfg3 hybrid
    assert by[("coherent", "hybrid", 48, 4, 0, 4)]["ranges"] == [[0, 16], [12, 28], [24, 40], [36, 48]]                                 # cfg4
    r5 = by[("coherent", "hybrid_ctx", 96, 8, 0, 4)]                                                                                    # cfg5
    assert r5["cs"] == 16 and r5["ranges"][-2:] == [[72, 88], [84, 96]] and len(r5["ranges"]) == 8
    assert by[("coherent", "hybrid", 32, 1, 0, 4)]["ranges"] == [[0, 16], [12, 28], [24, 32]]
    r3 = by[("coherent", "hybrid", 32, 3, 0, 4)]
    assert r3["cs"] == 10 and len(r3["ranges"]) == 6 and r3["ranges"][-1] == [30, 32]
    assert sum(1 for r in rows if r.get("hang")) > 0


def test_oracle_planner_equals_the_executed_reference():
    from oracle.pipeline_ref import PlannerHang, my_ranges, plan_chunks
    for r in planner_rows():
        args = (r["T"], r["world"], r["chunk_size"], r["overlap"], r["mode"] == "fsdp", rule_of(r))
        if r.get("hang"):
            with pytest.raises(PlannerHang):
                plan_chunks(*args)
            continue
        cs, ov, ranges = plan_chunks(*args)
        assert [list(x) for x in ranges] == r["ranges"], r
        assert (cs, ov) == (r["cs"], r["ov"]), r
        # :194 on rank 0's share (:184): frames x channels x 2, no spatial extent
        assert sum((e - s) * 4 * 2 for s, e in my_ranges(ranges, r["world"], 0)) == r["network_bytes"], r


def test_product_planner_equals_the_executed_reference():
    import vdx  # noqa: F401
    from vdx.planner import PlannerError, plan
    for r in planner_rows():
        args = dict(total=r["T"], world=r["world"], chunk_size=r["chunk_size"], overlap=r["overlap"], no_chunking=r["mode"] == "fsdp",
                    overlap_rule=rule_of(r))
        if r.get("hang"):
            with pytest.raises(PlannerError):
                plan(**args)
            continue
        cp = plan(**args)
        assert [list(x) for x in cp.ranges] == r["ranges"] and (cp.chunk, cp.overlap) == (r["cs"], r["ov"]), r
        # chunk -> rank: the fixture's ranges were rebuilt from what each rank put into all_gather_object
        for rank in range(r["world"]):
            assert [list(x) for x in cp.for_rank(rank)] == r["ranges"][rank::r["world"]]


@pytest.mark.parametrize("name", NUMERIC)
def test_oracle_pipeline_equals_the_executed_reference(name):
    """`oracle/pipeline_ref.py` on the same stand-in model objects reproduces what the reference's `__init__` / `_denoise` /
    `__call__` produced: the context tensor, the first UNet input (cat + ctx injection), the timestep sequence, every
    denoised chunk in gather order, and every blended frame as it reached `vae.decode` (lat[:, :, i] / 0.18215).
    "exact_*" fixtures (elementwise stand-in UNet): BIT FOR BIT, whole chain — world sizes 1, 2, 3, 4 and 8, up to the
    reference's default 50 steps.  "oracle_*" fixtures (fp32 oracle UNet: float sums depend on thread count and CPU): the
    denoised chunks within rel-L2 1e-3 (measured ~1e-4), and the blend — fed the FIXTURE's chunks — bit for bit."""
    from oracle.ddim_ref import DDIMSchedulerRef
    from oracle.pipeline_ref import base_noise, denoise, global_context, my_ranges, plan_chunks, ramp_blend
    g = np.load(os.path.join(GOLD, f"ref_exec_{name}.npz"))
    T, hw, steps, world = int(g["T"]), int(g["hw"]), int(g["steps"]), int(g["world"])
    mode, ref_file = str(g["mode"]), str(g["ref_file"])
    rule = "coherent" if ref_file == "fsdp_chunked_coherent.py" else "third"
    cs, ov, ranges = plan_chunks(T, world, int(g["chunk_size_arg"]), int(g["overlap_arg"]), mode == "fsdp", rule)
    assert (cs, ov) == (int(g["cs"]), int(g["ov"])) and [list(r) for r in ranges] == g["ranges"].tolist()
    sched = DDIMSchedulerRef()
    sched.set_timesteps(steps)
    assert sched.timesteps.tolist() == g["timesteps"].tolist()
    exact = str(g["unet_kind"]) == "exact"
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    unet = ExactUNet() if exact else oracle_unet()
    emb = text_table()
    uncond, cond = emb[1:], emb[:1]                 # reference :103: cond_emb, uncond_emb = emb[:1], emb[1:]
    base = base_noise(T, 4, hw, hw, sched.init_noise_sigma)
    ctx = global_context(T, 4, hw, hw, sched.init_noise_sigma) if mode == "hybrid_ctx" else None
    if ctx is not None:
        assert torch.equal(ctx, torch.from_numpy(g["ctx"]))
    # first UNet input of rank 0's first chunk (:133-137)
    s0, e0 = my_ranges(ranges, world, 0)[0]
    x = torch.cat([base[:, :, s0:e0]] * 2)
    if ctx is not None:
        x = x + 0.35 * ctx.repeat(1, 1, e0 - s0, 1, 1)
    assert torch.equal(x, torch.from_numpy(g["x_first"])) and int(g["t_first"]) == int(sched.timesteps[0])
    gathered = []
    for rank in range(world):
        for k, (s, e) in enumerate(my_ranges(ranges, world, rank)):
            den = denoise(unet, sched, base[:, :, s:e].clone(), uncond, cond, 7.5, ctx, 0.35)
            want = torch.from_numpy(g[f"den_r{rank}_k{k}"])
            if exact:
                assert torch.equal(den, want), (name, rank, k)
            else:
                err = float((den.double() - want.double()).norm() / want.double().norm())
                assert err <= 1e-3, (name, rank, k, err)
                den = want
            gathered.append((s, e, den))
    # one UNet call per (chunk, step) and nothing else: rank 0 of the reference's run made exactly its share
    assert int(g["unet_calls"]) == len(my_ranges(ranges, world, 0)) * steps and len(unet.calls) == len(ranges) * steps
    lat = ramp_blend(gathered, T, ov, base)
    assert lat.dtype == torch.float32
    z_ref = torch.from_numpy(g["z"])
    for i in range(T):
        z = lat[:, :, i]
        if ref_file == "chunk_only.py":
            z = z.to(torch.float16)             # chunk_only.py:154 casts before the division
        assert torch.equal((z / 0.18215)[0], z_ref[i]), (name, i)
    # the first and last frame of the video are exactly zero (every chunk's ramp starts and ends at 0) — when ov > 0
    if ov > 0:
        assert float(z_ref[0].abs().max()) == 0.0 and float(z_ref[-1].abs().max()) == 0.0


@pytest.mark.parametrize("name", NUMERIC)
def test_result_row_contract_equals_the_executed_reference(name):
    """CSV header, `network_bytes` formula and the boundary L1 metric (`temp_instab`) of the product's host code against
    the row the reference wrote."""
    import vdx  # noqa: F401
    from vdx import metrics
    g = np.load(os.path.join(GOLD, f"ref_exec_{name}.npz"))
    assert str(g["csv_header"]).split(",") == metrics.CSV_HEADER
    ranges = [tuple(r) for r in g["ranges"].tolist()]
    world = int(g["world"])
    assert int(g["network_bytes"]) == sum((e - s) * 4 * 2 for s, e in ranges[0::world])
    # the stand-in decoder of the fixture run, applied to the recorded latents -> the frames the reference measured
    frames = frames_u8(torch.from_numpy(g["z"]))
    want = float(g["temp_instab"])
    ref_file, mode = str(g["ref_file"]), str(g["mode"])
    got = metrics.boundary_l1(frames, ranges)
    if ref_file == "fsdp_chunked_coherent.py" and mode == "fsdp":
        assert np.isnan(want)                   # :229 `and not cfg.no_chunking`
    elif np.isnan(want):
        assert got is None
    else:
        assert got == want


def test_fsdp_wrap_arguments_of_the_executed_reference():
    """Row a7: what the reference passes to `FullyShardedDataParallel` (:63-88), as recorded while it ran, and its
    `wrap_policy` on three probes (a container never wraps; >= 10 M un-wrapped parameters wrap)."""
    g = np.load(os.path.join(GOLD, "ref_exec_exact_hybrid_ctx_w2.npz"))
    calls = json.loads(str(g["fsdp_kwargs"]))
    assert [c[0] for c in calls] == ["ExactUNet", "TextEncoder", "Linear"]        # unet, text_encoder, each trainable VAE child
    for _, kw in calls:
        assert kw["sharding_strategy"] == "ShardingStrategy.FULL_SHARD" and kw["use_orig_params"] == "False"
        assert "offload_params=True" in kw["cpu_offload"] and kw["device_id"] == "0" and kw["auto_wrap_policy"] == "wrap_policy"
        assert kw["mixed_precision"].count("torch.float16") == 3
    assert g["wrap_policy"].tolist() == [False, True, False]
    assert json.loads(str(g["from_pretrained_kwargs"])) == {"torch_dtype": "torch.float16", "low_cpu_mem_usage": "True",
                                                            "use_safetensors": "False", "device_map": "None"}
    # `--mode chunk` of the same file and chunk_only.py wrap nothing
    assert json.loads(str(np.load(os.path.join(GOLD, "ref_exec_exact_chunk_only_cfg1.npz"))["fsdp_kwargs"])) == []
    assert json.loads(str(np.load(os.path.join(GOLD, "ref_exec_exact_chunk_w1.npz"))["fsdp_kwargs"])) == []
    # fsdp_chunked.py passes the same arguments (device_id = local_rank instead of current_device())
    for _, kw in json.loads(str(np.load(os.path.join(GOLD, "ref_exec_exact_fsdp_chunked_w2.npz"))["fsdp_kwargs"])):
        assert kw["sharding_strategy"] == "ShardingStrategy.FULL_SHARD" and "offload_params=True" in kw["cpu_offload"]


def test_fsdp_py_loop_equals_the_executed_reference():
    """`fsdp.py` — the strategy file BASELINE cfg3 names (`FSDPBenchmark.run`, :108-215): its own CFG loop (:141-153: literal
    7.5, no `scale_model_input`), noise drawn unseeded (the harness seeded the global generator), per-frame decode through
    `pipe.decode_latents`.  The oracle's `denoise` from the same start reproduces the final latent bit for bit; the CSV row
    says mode "fsdp", chunk_size 0, overlap 0, network_bytes 0, empty boundary metrics."""
    from oracle.ddim_ref import DDIMSchedulerRef
    from oracle.pipeline_ref import denoise
    import vdx  # noqa: F401
    from vdx import metrics
    g = np.load(os.path.join(GOLD, "ref_exec_exact_fsdp_file_w2.npz"))
    T, hw, steps = int(g["T"]), int(g["hw"]), int(g["steps"])
    # the start latent is what the script drew (:133-137; the generator also served the stand-in modules' constructors, so
    # it is taken from the recorded first UNet input, whose two halves are the same tensor: `torch.cat([lat, lat])`, :143)
    x_first = torch.from_numpy(g["x_first"])
    lat0 = x_first[:1].clone()
    assert tuple(x_first.shape) == (2, 4, T, hw, hw) and torch.equal(x_first[0], x_first[1]) and 0.9 < float(lat0.float().std()) < 1.1
    sched = DDIMSchedulerRef()
    sched.set_timesteps(steps)
    assert sched.timesteps.tolist() == g["timesteps"].tolist() and int(g["unet_calls"]) == steps
    emb = text_table()
    lat = denoise(ExactUNet(), sched, lat0.clone(), emb[1:], emb[:1], 7.5, None)
    z = torch.from_numpy(g["z"])                                               # (T, 4, h, w) fp32: lat.cpu().float()[:, :, i]
    assert z.dtype == torch.float32 and torch.equal(lat[0].float().permute(1, 0, 2, 3), z)
    assert str(g["csv_header"]).split(",") == metrics.CSV_HEADER and str(g["csv_mode"]) == "fsdp"
    assert (int(g["cs"]), int(g["ov"]), int(g["network_bytes"]), str(g["csv_temp_instab"])) == (0, 0, 0, "")
    calls = json.loads(str(g["fsdp_kwargs"]))
    assert [c[0] for c in calls] == ["ExactUNet", "TextEncoder", "Linear"]
    assert "offload_params=True" in calls[0][1]["cpu_offload"] and "offload_params=False" in calls[2][1]["cpu_offload"]     # fsdp.py:96: the VAE children stay on the GPU
