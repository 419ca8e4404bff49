"""-m gpu: the reference's call sequence on the import shims (diffusers / pynvml / cv2 stand-ins) with its exact FSDP
wrap arguments (`fsdp_chunked_coherent.py:63-88`) around the parameter-less HIP modules — one process, world-1 RCCL."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_call_sequence_runs_on_the_shims_with_fsdp_wrap(gpu, tmp_path):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
               PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out = str(tmp_path / "out.mp4")
    r = subprocess.run([sys.executable, "-m", "vdx.compat.run", os.path.join(ROOT, "tests", "compat_reference_style.py"), out],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("COMPAT-OK")]
    assert line, r.stdout[-2000:]
    assert "unet FullyShardedDataParallel" in line[0] and "in_channels 4" in line[0] and "frames 5" in line[0]
    assert os.path.getsize(out) > 1000


def test_from_pretrained_directory_end_to_end(gpu, tmp_path):
    """The one branch a user with real weights takes (`fsdp_chunked_coherent.py:55-61`): a checkpoint directory in
    diffusers layout (written here at narrow widths: config.json files, .safetensors, scheduler_config.json) loaded
    through the diffusers shim; the UNet it builds, run on the GPU, against the oracle holding the same state dict;
    text encoder and VAE decode run on the loaded weights."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ckpt_dir
    import vdx  # noqa: F401
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig as RefCfg
    from vdx.compat import diffusers_shim as d
    root = str(tmp_path / "ckpt")
    ckpt_dir.write(root)
    pipe = d.DiffusionPipeline.from_pretrained(root, torch_dtype=torch.float16, low_cpu_mem_usage=True, use_safetensors=False,
                                               device_map=None)
    ref_m = UNet3DConditionModelRef(RefCfg.tiny(ch=ckpt_dir.UNET_CH, cross=ckpt_dir.CROSS, in_heads=8)).eval()
    ref_m.load_state_dict({k: v.half().float() for k, v in ckpt_dir.unet_state_dict().items()})
    g = torch.Generator().manual_seed(5)
    sample = torch.randn(2, 4, 4, 16, 24, generator=g).half()
    ehs = torch.randn(2, 77, ckpt_dir.CROSS, generator=g).half()
    with torch.no_grad():
        ref = ref_m(sample.float(), torch.tensor(500), ehs.float()).sample
    out = pipe.unet(sample.to(gpu), torch.tensor(500, device=gpu), encoder_hidden_states=ehs.to(gpu)).sample
    err = float((out.float().cpu().double() - ref.double()).norm() / ref.double().norm())
    print(f"from_pretrained(dir) unet rel-L2 {err:.3e}")
    assert err <= 4e-3
    ids = pipe.tokenizer(["a red panda", ""], padding="max_length", max_length=pipe.tokenizer.model_max_length, truncation=True,
                         return_tensors="pt").input_ids.to(gpu)
    emb = pipe.text_encoder(ids)[0]
    assert emb.shape == (2, 77, 128) and bool(torch.isfinite(emb.float()).all())
    img = pipe.vae.decode(torch.randn(1, 4, 8, 8, device=gpu, dtype=torch.float16) / pipe.vae.config.scaling_factor).sample
    assert img.shape == (1, 3, 64, 64) and bool(torch.isfinite(img.float()).all())


def test_front_end_writes_the_reference_csv_row(gpu, tmp_path):
    """`python -m vdx.pipeline` with the reference's flags (fsdp_chunked_coherent.py:281-300) on tiny synthetic weights: the job
    runs end to end (text tower, chunked denoising, exchange, blend, VAE decode, boundary metrics, mp4, memory) and appends
    the reference's CSV row; `--emu_*` sleep as the reference does (:195-199, 257-258); both exchanges give the same frames."""
    import csv
    import time
    import vdx  # noqa: F401
    from vdx import metrics
    from vdx.pipeline import main
    from vdx.planner import plan
    out_csv, mp4 = str(tmp_path / "r.csv"), str(tmp_path / "o.mp4")
    base = ["--model_id", "synthetic:tiny", "--num_frames", "10", "--steps", "2", "--height", "128", "--width", "256", "--chunk_size", "6",
            "--overlap", "2", "--mode", "hybrid_ctx", "--out_csv", out_csv, "--out_video", mp4, "--noise_device", "cpu"]
    assert main(base) == 0
    assert main(base + ["--emu_rtt_ms", "300", "--emu_bw_mbps", "0.001", "--exchange", "halo"]) == 0
    rows = list(csv.DictReader(open(out_csv)))
    assert list(rows[0].keys()) == metrics.CSV_HEADER and len(rows) == 2
    cp = plan(10, 1, 6, 2)
    for r in rows:
        assert r["mode"] == "hybrid_ctx" and int(r["world_size"]) == 1 and int(r["num_frames"]) == 10
        assert (int(r["chunk_size"]), int(r["overlap"])) == (cp.chunk, cp.overlap)
        assert int(r["network_bytes"]) == sum((e - s) * 4 * 2 for s, e in cp.ranges)          # :194
        assert float(r["latency_s"]) > 0 and float(r["throughput_fps"]) > 0 and int(r["peak_vram_mb"]) > 0
        assert r["temp_instab"] != "" and r["flow_err"] != ""                                 # two chunks: one boundary
    assert rows[0]["temp_instab"] == rows[1]["temp_instab"]          # halo and allgather blend to the same frames
    assert os.path.getsize(mp4) > 1000
    # the sleeps the job took, as it reports them: gauss(300, 0) ms + payload / (0.001 Mbps) = 96 B / 125 B/s before the exchange,
    # 300 ms before the reduction (:195-199, 257-258); wall-clock around them
    from vdx.pipeline import build_arg_parser, config_from_args, run_job
    a = build_arg_parser().parse_args(base + ["--emu_rtt_ms", "300", "--emu_bw_mbps", "0.001"])
    t0 = time.time()
    res = run_job(config_from_args(a), out_video=None)
    wall = time.time() - t0
    payload = sum((e - s) * 4 * 2 for s, e in cp.ranges)
    assert abs(res["emu_gather_delay_s"] - (0.3 + payload / 125.0)) < 1e-9 and res["emu_reduce_delay_s"] == 0.3
    assert wall > res["emu_gather_delay_s"] + res["emu_reduce_delay_s"]


def test_front_end_two_ranks_under_torchrun(gpu, tmp_path):
    """The front end as the reference's sweep script launches its own (`torchrun --nproc_per_node=2 …`, full_experiments_ZeroscopeXL.sh):
    two real processes (both on this box's one GPU: gloo + VDX_SHARE_GPU, rehearsal aids), `--mode hybrid_ctx`: parameters sharded
    1/2 per rank, ctx broadcast, one window per rank, all-gather + blend, rank 0 writes the row."""
    import csv
    out_csv, mp4 = str(tmp_path / "r2.csv"), str(tmp_path / "o2.mp4")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(VDX_DIST_BACKEND="gloo", VDX_SHARE_GPU="1", PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29577", "-m", "vdx.pipeline", "--model_id", "synthetic:tiny", "--num_frames", "12", "--steps", "2",
                        "--height", "128", "--width", "256", "--overlap", "2", "--mode", "hybrid_ctx", "--out_csv", out_csv,
                        "--out_video", mp4, "--noise_device", "cpu", "--emu_rtt_ms", "10"],
                       capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    rows = list(csv.DictReader(open(out_csv)))
    assert len(rows) == 1 and int(rows[0]["world_size"]) == 2 and rows[0]["mode"] == "hybrid_ctx"
    import vdx  # noqa: F401
    from vdx.planner import plan
    cp = plan(12, 2, 0, 2)
    assert (int(rows[0]["chunk_size"]), int(rows[0]["overlap"])) == (cp.chunk, cp.overlap)
    assert int(rows[0]["network_bytes"]) == sum((e - s) * 4 * 2 for s, e in cp.for_rank(0))          # rank 0's own chunk list (:194)
    assert rows[0]["temp_instab"] != "" and os.path.getsize(mp4) > 1000
