#!/usr/bin/env python3
"""bench.py — denoising-step throughput of the HIP path on Zeroscope-XL shapes (BASELINE.json).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A *step* = one iteration of `fsdp_chunked_coherent.py:132-142` for every chunk a rank owns, lock-step
across ranks: cat/ctx-inject -> UNet3D forward on the CFG batch of 2 -> guidance combine -> DDIM update.
The workload is the BASELINE.json configuration of the world size (`config.workload` names it):
  N=1  cfg2: Zeroscope-XL, 24 frames @576x1024 (latent 72x128), monolithic, weights resident.
  N=2  cfg3: the same 24-frame clip, `--mode fsdp` (fsdp.py:130-153, planner :150-151 + padding :174-177): BOTH ranks
       denoise the full clip; only parameter memory is saved (1/2 of every unit per GPU, per-unit RCCL all-gather).
  N=4  cfg4: 48 frames, planner's 4 x 16-frame windows (overlap 4), FSDP + chunked (`hybrid`), one window per GPU.
  N=8  cfg5: 96 frames, 7 x 16 + 1 x 12-frame windows, `hybrid_ctx` (global-context injection), one window per GPU.
  other N: 12*N frames through the planner's automatic chunking, `hybrid_ctx`.
UNet parameters are sharded 1/N per GPU for N > 1 (per-unit all-gather prefetched on a side stream); there is no other
collective inside a step; ctx broadcast before and the chunk exchange after the loop are outside the timed steps.
value = (USEFUL frames of the video / 24) * steps / max-over-ranks time: 24-frame-equivalent denoising steps per
second (FLOPs are linear in the frame count; frames that two windows both compute count once — SURVEY §8d(ii)).
`lockstep_steps_per_s` is the plain steps / time of the job.
`roofline` describes the kernel with the most time in the timed region among those that carry the step's FLOPs
(every GEMM / implicit-GEMM family and flash attention have HIP events around each launch, on the launch stream):
achieved = their algorithmic FLOPs / their summed launch time, against the dense fp16 MFMA peak.
Weights are synthetic (diffusers-shaped, seeded); inputs are synthetic noise/text embeddings.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TFLOP_PER_STEP_24F = 156.97      # algorithmic FLOPs of one CFG step at 24 f, XL
PMC_JSON = os.path.join(ROOT, "profiles", "pmc_current.json")
MONO_PEAKS_JSON = os.path.join(ROOT, "profiles", "monolithic_peaks.json")


def pmc_traffic(kernel):
    """(HBM bytes per launch of `kernel`, HBM bytes per step over all kernels, note) from the committed rocprofv3 PMC
    passes of this same workload (FETCH_SIZE x2-corrected + WRITE_SIZE, tools/pmc_summary.py --json): counters cannot
    be read from inside the bench process.  The profile records the hash of the kernel sources it was taken on;
    when the sources this process runs hash differently (or the kernel is not in the profile) the answer is None —
    a stale profile must not pass for a measurement of this build."""
    try:
        from vdx._lib import source_sha
        prof = json.load(open(PMC_JSON))
        meta = prof.get("_meta", {})
        if meta.get("source_sha") != source_sha():
            return None, None, f"profiles/pmc_current.json was taken on kernel sources {meta.get('source_sha')}, this build is {source_sha()}"
        e = prof.get(kernel)
        per_step = meta.get("hbm_bytes_all_kernels", 0) / max(meta.get("forwards", 2), 1)
        return (round(e["hbm_read_bytes_per_launch"] + e["hbm_write_bytes_per_launch"]) if e else None,
                round(per_step) if per_step else None, None if e else "kernel not in the profile")
    except (OSError, ValueError, KeyError) as ex:
        return None, None, f"no usable PMC profile: {ex}"


def select_config(world, name=None):
    """BASELINE.json configuration for a world size -> (name, total frames T, mode, description)."""
    table = {"cfg2": (24, "mono"), "cfg3": (24, "fsdp"), "cfg4": (48, "hybrid"), "cfg5": (96, "hybrid_ctx")}
    if name is None:
        name = {1: "cfg2", 2: "cfg3", 4: "cfg4", 8: "cfg5"}.get(world)
    if name is None:
        return f"generic-{world}", 12 * world, "hybrid_ctx"
    T, mode = table[name]
    return name, T, mode


PEAK_MFMA_TFLOPS = 2500.0        # gfx950 dense fp16/bf16 matrix peak (MI355X_MICROARCH.md)


def shared_prefix_tflop(frames: int, hw: int = 72 * 128) -> float:
    """Algorithmic TFLOP (2 x MACs, SURVEY §8d's counting) of the blocks the CFG-shared prefix computes for ONE batch item
    instead of two (unet3d.forward): conv_in, transformer_in, down_blocks.0.resnets.0 / temp_convs.0 and the first spatial
    transformer up to the query projection of its cross-attention — the work a duplicated forward does twice.  Per
    level-0 row (320 channels; transformer_in inner 512, 8 heads):"""
    c, i = 320, 512
    macs = 36 * c                                                   # conv_in (4 -> 320, 3x3)
    macs += c * i + 2 * (4 * i * i + 2 * frames * i) + 12 * i * i + i * c      # transformer_in: proj_in, 2 x (q|k|v|out + FxF core), GEGLU FF, proj_out
    macs += 2 * 9 * c * c                                           # ResnetBlock2D conv1 + conv2
    macs += 4 * 3 * c * c                                           # TemporalConvLayer, four (3,1,1) convolutions
    macs += c * c + 3 * c * c + 2 * hw * c + c * c + c * c          # spatial: proj_in, q|k|v, self-attention core, to_out, cross-attention q
    return 2.0 * macs * frames * hw / 1e12


def cpu_baseline_inputs(frames: int):
    """The bounded sample's inputs (fp16-rounded, so the oracle and the HIP forward see the same numbers): ONE latent for both
    CFG items — what `fsdp_chunked_coherent.py:133` builds — and two different text embeddings."""
    g = torch.Generator().manual_seed(0)
    lat = torch.randn(1, 4, frames, 72, 128, generator=g).half()
    e = torch.randn(2, 77, 1024, generator=g).half()
    return lat, e


def cpu_baseline(frames: int, threads: int, state_dict=None, hip_out=None):
    """Oracle (fp32 PyTorch-CPU restatement, kind "port") timed on this host on a bounded sample:
    one CFG UNet forward with XL-shaped weights at `frames` frames @ 576x1024; FLOPs are linear in
    the frame count, so the 24-frame rate is sample_time * 24/frames.
    `state_dict` (the table the HIP UNet of this process was loaded from) + `hip_out` (its output on
    `cpu_baseline_inputs(frames)`): the oracle then runs on the SAME weights and inputs and the line carries
    `rel_l2_vs_hip` — a parity number at the benchmark's own spatial extent from every driver run.
    The oracle is the checker here, never the thing measured as the product."""
    from oracle.unet3d_ref import UNet3DConditionModelRef, UNet3DConfig
    torch.set_num_threads(threads)
    with torch.device("meta"):
        m = UNet3DConditionModelRef(UNet3DConfig.zeroscope())
    m = m.to_empty(device="cpu").eval()
    with torch.no_grad():
        if state_dict is not None:
            m.load_state_dict({k: v.float().cpu() for k, v in state_dict.items()})
        else:
            for p in m.parameters():
                p.normal_(0.0, 0.02)
    lat, e = cpu_baseline_inputs(frames)
    x = torch.cat([lat, lat]).float()
    t0 = time.time()
    with torch.no_grad():
        y = m(x, torch.tensor(981), e.float()).sample
    dt = time.time() - t0
    out = {"value": round(1.0 / (dt * 24.0 / frames), 6), "unit": "steps/s", "cores": threads, "kind": "port",
           "sample": f"oracle fp32 torch-CPU UNet3D, XL widths, 1 CFG forward at {frames} of 24 frames @576x1024 "
                     f"({dt:.1f} s), scaled x{24 / frames:g} to the 24-frame step"}
    if hip_out is not None:
        out["rel_l2_vs_hip"] = round(float((hip_out.double() - y.double()).norm() / y.double().norm()), 6)
        out["rel_l2_vs_hip_note"] = (f"HIP fp16 forward (shared CFG prefix) vs this fp32 oracle forward: same seeded weights, same inputs, "
                                     f"(2,4,{frames},72,128); test bound 4e-3 (tests/test_full_extent_gpu.py)")
    return out


def box_probe(dev):
    """What THIS box gives a fixed instruction stream and a fixed copy, measured in this process before the warm-up and outside
    the timed region (~0.2 s): the boxes of the pool differ by several per cent as a whole (BENCH_r05 ran every kernel 7-9 %
    slower than the builder's box on unchanged sources), and a line without this cannot tell a slower part from a slower build.
      mfma_probe_tflops  `vdx_probe_mfma_f16`: dense v_mfma_f32_32x32x16_f16, two waves per SIMD, per-lane operands, every CU
                         (main() takes it again right behind the timed steps as mfma_probe_tflops_after_steps: reported, not used)
      hbm_probe_gbs      1 GiB device-to-device copy (read + write bytes / time, median of 3)
    `tools/perf_guard.py` and a reader normalise `ms_per_step` by mfma_probe_tflops (the step is MFMA-bound by arithmetic)."""
    from vdx import ops
    out = {"mfma_probe_tflops": round(ops.probe_mfma(dev), 1)}
    n = 1 << 30
    a = torch.zeros(n, dtype=torch.uint8, device=dev)
    b = torch.empty_like(a)
    b.copy_(a)
    ts_ = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        e1.synchronize()
        ts_.append(e0.elapsed_time(e1))
    out["hbm_probe_gbs"] = round(2 * n / (sorted(ts_)[1] * 1e-3) / 1e9, 1)
    del a, b
    torch.cuda.empty_cache()
    return out


class SclkSampler:
    """Mean shader clock over the timed steps, read from amdsmi by a thread every 20 ms (host only: no GPU work, no
    effect on the stream).  None with a reason where amdsmi cannot be used (not every box lets an ordinary user in)."""

    def __init__(self, index):
        self.samples, self.err, self._stop, self._thr = [], None, False, None
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self._smi, self._h = amdsmi, hs[index if index < len(hs) else 0]
            self._read()                    # fail here, not in the thread
        except Exception as e:              # noqa: BLE001 — a diagnostic: never fail the bench line for it
            self.err = f"{type(e).__name__}: {e}"[:160]

    def _read(self):
        d = self._smi.amdsmi_get_clock_info(self._h, self._smi.AmdSmiClkType.GFX)
        v = d.get("clk", d.get("cur_clk"))
        return float(v) if isinstance(v, (int, float)) else None

    def _run(self):
        while not self._stop:
            try:
                v = self._read()
                if v:
                    self.samples.append(v)
            except Exception as e:          # noqa: BLE001
                self.err = f"{type(e).__name__}: {e}"[:160]
                return
            time.sleep(0.02)

    def start(self):
        if self.err is None:
            import threading
            self._thr = threading.Thread(target=self._run, daemon=True)
            self._thr.start()

    def stop(self):
        self._stop = True
        if self._thr is not None:
            self._thr.join(timeout=1.0)
        if self.samples:
            return {"sclk_mhz_mean": round(sum(self.samples) / len(self.samples), 1), "sclk_mhz_min": min(self.samples),
                    "sclk_mhz_max": max(self.samples), "sclk_samples": len(self.samples)}
        return {"sclk_mhz_mean": None, "sclk_note": self.err or "no samples"}


def copy_activity_of_a_step(step, i, lats):
    """One more step under torch.profiler: do the parameter pulls of the peer transport show up as blit KERNELS
    (`__amd_rocclr_copyBuffer`: compute units) or only as memcpy activities (copy engines)?  vdx/shard.py claims the
    latter for remote pulls; on one GPU (local copies) the runtime uses blit kernels.  Answered here, not assumed."""
    try:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            step(i, lats)
            torch.cuda.synchronize()
        blit = memcpy = 0
        for ev in prof.events():
            nm = ev.name or ""
            if "copyBuffer" in nm or "rocclr_copy" in nm:
                blit += 1
            elif nm.startswith("Memcpy") or "hipMemcpyAsync" == nm:
                memcpy += 1
        return {"step_blit_copy_kernels": blit, "step_memcpy_activities": memcpy}
    except Exception as e:      # noqa: BLE001 — a diagnostic: never fail the bench line for it
        return {"step_copy_activity_error": f"{type(e).__name__}: {e}"[:200]}


def self_launch_command(argv, n, port):
    """The command a bare `python bench.py --gpus N` (N > 1, no launcher) re-runs itself under: one rank per GPU through
    torch.distributed.run on 127.0.0.1 — exactly what the driver's own multi-GPU command is."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(argv, n):
    """Parent of a launcher-less multi-GPU run.  Nothing here touches the GPU (no HIP call, no `torch.cuda.*`): the ranks
    are CHILD processes; rank 0's single JSON line and the launcher's exit code are relayed."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: what RCCL and the peer-mapped shard arenas need here
    env.setdefault("OMP_NUM_THREADS", "1")
    from vdx.shard import configure_rccl_env
    configure_rccl_env(env)
    r = subprocess.run(self_launch_command(argv, n, port), stdout=subprocess.PIPE, text=True, env=env)
    lines = [ln for ln in r.stdout.splitlines() if ln.lstrip().startswith("{") and '"metric"' in ln]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stderr.write(r.stdout[-4000:])
    raise SystemExit(r.returncode if (r.returncode or lines) else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--frames", type=int, default=0, help="override the total frame count of the configuration (dev aid)")
    ap.add_argument("--config", default=None, choices=["cfg2", "cfg3", "cfg4", "cfg5"],
                    help="BASELINE configuration to run (default: the one of the world size)")
    ap.add_argument("--as-world", type=int, default=0,
                    help="with --rehearse-dist on one GPU: plan as if the job had this many ranks ...")
    ap.add_argument("--as-rank", type=int, default=0, help="... and run the windows of this rank")
    ap.add_argument("--cpu-frames", type=int, default=2,
                    help="frames of the bounded CPU-baseline sample (0 = skip); 2 frames = ~13 s on 16 cores")
    ap.add_argument("--shapes", type=int, default=0, help="also list the top-N GEMM shapes by time (dev aid)")
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="N=1 only: run the multi-GPU code path (RCCL init, sharded weights, collectives) with world 1")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="rehearsal aid: gloo lets several ranks of a real multi-process job share ONE GPU (with --share-gpu); "
                         "the driver's runs use nccl (= RCCL)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal aid: every rank computes on cuda:0")
    ap.add_argument("--native-comm", action="store_true",
                    help="distributed / rehearsal runs: gather the parameter shards through the C-ABI RCCL entry point "
                         "(vdx_allgather_shard, vdx/comm.py) instead of torch.distributed's all_gather_into_tensor")
    ap.add_argument("--shard-transport", default=None, choices=["collective", "peer"],
                    help="parameter-gather transport of the shard store (default: collective = RCCL all-gather; peer = HIP-IPC mapped "
                         "arenas + copy-engine pulls, opt-in: vdx/shard.py)")
    ap.add_argument("--rehearsal", action="store_true",
                    help="required with --backend gloo / --share-gpu / --frames / --resident: states that the line is a "
                         "rehearsal, not a measurement of the BASELINE configuration (a driver run cannot take one by accident)")
    ap.add_argument("--peaked", action="store_true",
                    help="diagnostic: scale the spatial self-attention q|k projections x2 each (scores x4: row maxima tens of "
                         "nats above the mean, as trained checkpoints have) — the lazy softmax offset of the flash kernel then "
                         "has to move; the default synthetic weights give near-uniform attention, its best case")
    ap.add_argument("--no-profile", action="store_true", help="no per-kernel HIP events in the timed region")
    ap.add_argument("--no-shared-prefix", action="store_true",
                    help="diagnostic (needs --rehearsal): every block computes both CFG items, as a forward on torch.cat([lat]*2) does")
    ap.add_argument("--no-duplicate-leg", action="store_true",
                    help="skip the second timing of the same steps with the CFG-shared prefix off (`ms_per_step_full_duplicate`)")
    ap.add_argument("--rehearse-copies", action="store_true",
                    help="with --rehearse-dist --as-world N on one GPU: issue every parameter gather as N copies (the host call count of a node)")
    ap.add_argument("--hog", type=int, default=0,
                    help="with --rehearse-dist --as-world N: beside every parameter gather the side stream holds this many CUs "
                         "(workgroups of 256 threads with 64 KB of LDS that touch no memory) for the time a ring all-gather of the "
                         "group's remote bytes takes at --hog-gbs: what RCCL's channel kernels take from the step's persistent grids")
    ap.add_argument("--hog-gbs", type=float, default=100.0, help="modelled all-gather rate per GPU for --hog (GB/s)")
    ap.add_argument("--reserve-cus", type=int, default=-1,
                    help="rehearsal: CUs every persistent grid leaves free (vdx_set_reserved_cus); default = what the store chose")
    ap.add_argument("--profile-all", action="store_true",
                    help="HIP events around EVERY matrix kernel (the per-family table `gemm_kernels`): costs ~2.6 ms per step; the "
                         "default times only the two candidates for the dominant kernel (3x3-conv GEMM, spatial flash attention)")
    ap.add_argument("--ff-block-mb", type=int, default=0, help="diagnostic: feed-forward row-block size of the memory-lean mode")
    ap.add_argument("--no-lean", action="store_true",
                    help="diagnostic: sharded weights WITHOUT the memory-lean execution order (row-blocked feed-forward, split attention, ...)")
    ap.add_argument("--lean-parts", default=None,
                    help="diagnostic: which optional parts of the memory-lean order stay on besides the row-blocked feed-forward: "
                         "comma list of concat, attn (default: both)")
    ap.add_argument("--resident", action="store_true",
                    help="diagnostic: keep the UNet weights resident (no shard store) in a distributed / rehearsal run")
    args = ap.parse_args()
    if (args.backend != "nccl" or args.share_gpu or args.frames or args.resident or args.no_lean or args.lean_parts is not None
            or args.no_shared_prefix) and not args.rehearsal:
        raise SystemExit("--backend gloo, --share-gpu, --frames and --resident change what is measured: pass --rehearsal with them")

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` without a launcher (the driver's N = 1 command has none): become the launcher
        self_launch(sys.argv[1:], args.gpus)
    # stdout carries exactly ONE JSON line: libraries that print to fd 1 (RCCL's version banner)
    # are sent to stderr for the whole run.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", 0))
    local = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if args.share_gpu:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist_mode = world > 1 or args.rehearse_dist
    if dist_mode:
        os.environ.setdefault("NCCL_DEBUG", "WARN")      # keep RCCL's version banner off stdout (one JSON line)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.rehearse_dist:
            os.environ["VDX_SHARD_FORCE_COLLECTIVE"] = "1"
        if args.backend == "nccl":
            from vdx.shard import configure_rccl_env
            configure_rccl_env()          # the process group's stream on a hardware queue of its own, before the communicator exists
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import vdx  # noqa: F401
    from vdx import ops
    from vdx.pipeline import seeded_noise
    from vdx.planner import plan
    from vdx.scheduler import DDIMScheduler
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict

    H, W = 72, 128
    cfg = UNet3DConfig.zeroscope()
    unet = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
    if args.peaked:
        for k_, w_ in unet.W.items():
            if k_.endswith(".attn1.to_qkv.weight") and ".transformer_blocks." in k_ and "temp_attentions" not in k_ and "transformer_in" not in k_:
                w_[:2 * w_.shape[0] // 3].mul_(2.0)       # q and k rows of the spatial self-attention projections
    if dist_mode and not args.resident:
        comm = None
        if args.native_comm:
            from vdx.comm import Comm
            comm = Comm.from_torch(dev)
        unet.shard_(rank, world, comm=comm, transport=args.shard_transport)       # 1/N of every unit per GPU, gathered per unit on a side stream
        if args.rehearse_dist and args.rehearse_copies and args.as_world > 1 and world == 1:
            # a gather = as_world LOCAL copies: the host-side call count of a node.  (On one GPU every one of them is a
            # blit kernel on the GPU that is computing; on a node 7 of 8 are remote pulls.  An upper bound of the cost.)
            unet.W.rehearse_copies = args.as_world
        if args.hog:
            if not (args.rehearse_dist and args.as_world > 1 and world == 1) or args.resident:
                raise SystemExit("--hog is a one-GPU rehearsal aid of the SHARDED store: use it with --rehearse-dist --as-world N, without --resident")
            unet.W.rehearse_hog = (args.hog, 64 << 10, args.hog_gbs, args.as_world)
        if args.reserve_cus >= 0:
            if not args.rehearse_dist:
                raise SystemExit("--reserve-cus is a rehearsal aid (the store sets the reserve itself for world > 1)")
            ops.set_reserved_cus(args.reserve_cus)
    if args.no_shared_prefix:
        unet.share_cfg_prefix = False
    if args.ff_block_mb:
        unet.ff_block_bytes = args.ff_block_mb << 20
    if args.no_lean:
        unet.ff_block_bytes = None
    if args.lean_parts is not None:
        unet.lean_concat, unet.lean_attn = "concat" in args.lean_parts, "attn" in args.lean_parts
    sched = DDIMScheduler()
    sched.set_timesteps(50, device=dev)
    plan_world, plan_rank = (args.as_world, args.as_rank) if (args.rehearse_dist and args.as_world) else (world, rank)
    cname, T, mode = select_config(plan_world, args.config)
    if args.frames:
        T = args.frames
    if mode == "mono" and dist_mode:
        mode = "fsdp"
    cp = plan(T, plan_world, chunk_size=0, overlap=4, no_chunking=mode in ("mono", "fsdp"))
    ranges = cp.for_rank(plan_rank)
    desc = {"mono": "monolithic, weights resident, 1 GPU",
            "fsdp": f"--mode fsdp: every rank denoises the full clip, UNet parameters sharded 1/{world} per GPU",
            "hybrid": f"--mode hybrid (FSDP + chunked): windows {list(cp.ranges)} round-robin over {plan_world} ranks, parameters sharded 1/{world}",
            "hybrid_ctx": f"--mode hybrid_ctx (FSDP + chunked + global-context injection): windows {list(cp.ranges)} round-robin over {plan_world} ranks, parameters sharded 1/{world}"}[mode]
    workload = (f"BASELINE {cname}: Zeroscope_v2_XL UNet3D, {T} frames @576x1024 (latent {H}x{W}), CFG batch 2, {desc}"
                + (", per-unit RCCL all-gather prefetch on a side stream" if dist_mode else ""))
    base = seeded_noise((1, 4, T, H, W), sched.init_noise_sigma, dev)
    ctx = None
    if mode == "hybrid_ctx":
        ctx = base.mean(dim=2, keepdim=True).contiguous()
        if dist_mode:
            dist.broadcast(ctx, src=0)
    lats = [base[:, :, s:e].clone() for s, e in ranges]
    torch.manual_seed(1)
    emb = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
    ts = sched._host_timesteps

    def step(i, lats):
        t = ts[i % len(ts)]
        out = []
        for lat in lats:
            x = ops.cfg_input(lat, ctx, 0.35)
            noise = unet(x, t, encoder_hidden_states=emb).sample
            out.append(sched.step_cfg(noise, t, lat, 7.5))
        return out

    box = box_probe(dev) if rank == 0 else {}
    for i in range(args.warmup):
        lats = step(i, lats)
    torch.cuda.reset_peak_memory_stats()
    sclk = SclkSampler(local) if rank == 0 else None

    def fence():
        torch.cuda.synchronize()
        if dist_mode:
            dist.barrier()
        torch.cuda.synchronize()

    prof = None if args.no_profile else []
    ops.PROFILE = prof
    # the roofline object needs the dominant kernel's launch durations from inside the timed region; the two kernels that
    # can be it (30.5 / 30.8 ms per forward; the next family has 22) get events, the other ~500 launches per step do not
    ops.PROFILE_ONLY = None if args.profile_all else ("gemm_kernel<256, 320, 4, 2, 1, false, true, 0>", "flash_attn_kernel<2, false")
    store = unet.W if hasattr(unet.W, "transport") else None
    g0, h0 = (store.gathers, store.gather_host_s) if store is not None else (0, 0.0)
    fence()
    if sclk is not None:
        sclk.start()
    t0 = time.perf_counter()
    for i in range(args.steps):
        lats = step(args.warmup + i, lats)
    fence()
    dt = time.perf_counter() - t0
    if sclk is not None:
        box.update(sclk.stop())
    if rank == 0:
        # The probe once more right behind the timed steps, REPORTED ONLY: round 6 measured it on five boxes — it reads 1 453 to
        # 1 751 TFLOP/s (a sustained dense-MFMA stream runs into each part's power limit, 1.39-1.67 GHz) while the step differs by
        # 4.5 %, so it does not predict the step.  The reading BEFORE the warm-up (a short burst on a part that has just synthesised
        # and packed the weights) does: executed TFLOP/s over it is 0.640-0.647 on all five (profiles/r06_box_probe.md).
        from vdx import ops as _ops
        box["mfma_probe_tflops_after_steps"] = round(_ops.probe_mfma(dev), 1)
    g1, h1 = (store.gathers, store.gather_host_s) if store is not None else (0, 0.0)
    ops.PROFILE = None
    finite = all(bool(torch.isfinite(lat.float()).all()) for lat in lats)
    peak_gb = torch.cuda.max_memory_allocated() / 2 ** 30
    free_b, total_b = torch.cuda.mem_get_info(dev)            # device-wide (the reference's pynvml `used`, :41-45,262)
    # The same K steps once more with the CFG-shared prefix OFF (every block computes both batch items, as a forward on
    # `torch.cat([lat]*2)` does): the line carries both times, so a reader sees what the product's default saves and that
    # the headline is not a step with work dropped — only a duplicate (same output bits, tests/test_unet_gpu.py).
    shared_on = bool(getattr(unet, "last_forward_shared_prefix", False))
    dt_dup = 0.0
    if shared_on and not args.no_duplicate_leg:
        unet.share_cfg_prefix = False
        lats_d = step(args.warmup + args.steps, lats)           # warm the shapes of the duplicated form
        fence()
        t0 = time.perf_counter()
        for i in range(args.steps):
            lats_d = step(args.warmup + i, lats_d)
        fence()
        dt_dup = time.perf_counter() - t0
        unet.share_cfg_prefix = True
        del lats_d
    tt = torch.tensor([dt, peak_gb, (total_b - free_b) / 2 ** 30, dt_dup], device=dev, dtype=torch.float64)
    if dist_mode:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt, peak_gb, used_gb, dt_dup = float(tt[0]), float(tt[1]), float(tt[2]), float(tt[3])

    if rank == 0:
        my_frames = sum(e_ - s_ for s_, e_ in ranges)                        # frames THIS rank computes per step
        computed = sum(e_ - s_ for s_, e_ in cp.ranges)                     # all ranks (overlaps / replicas counted twice)
        useful = T                                                          # frames of the video
        if plan_world != world:                                             # one-GPU rehearsal of one rank's share
            computed, useful = my_frames, my_frames
        tf_step = TFLOP_PER_STEP_24F * my_frames / 24.0
        sps = (useful / 24.0) * args.steps / dt
        try:
            mono = json.load(open(MONO_PEAKS_JSON)).get(str(T))
        except (OSError, ValueError):
            mono = None
        out = {
            "metric": "denoising steps/sec, Zeroscope-XL 24f@1024x576 (CFG UNet3D forward + guidance + DDIM)",
            "value": round(sps, 5), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": cname, "mode": mode, "total_frames": T,
                       "windows": [list(r) for r in cp.ranges], "frames_rank0": my_frames, "scheduler": "DDIM-50",
                       "guidance_scale": 7.5},
            "lockstep_steps_per_s": round(args.steps / dt, 5),                 # plain steps of the job per second
            "frame_steps_per_s": round(useful * args.steps / dt, 3),          # USEFUL frames x steps / wall: what scales
            "computed_frame_steps_per_s": round(computed * args.steps / dt, 3),   # incl. overlap frames computed twice
            "peak_hbm_gb_per_gpu": round(peak_gb, 3), "device_used_gb_per_gpu": round(used_gb, 3),
            # per-device peak against the monolithic single-GPU peak at the SAME total frame count
            # (profiles/monolithic_peaks.json, measured by tools/mem_profile.py --frames T); north star: <= 0.15 at N = 8
            "peak_hbm_frac_of_monolithic": round(peak_gb / mono, 4) if mono else None,
            "output_finite": finite,
            "box": box,
        }
        # EXECUTED FLOPs: the shared prefix computes its blocks for one item, not two.  `path_tflops_per_gpu` / `path_mfma_frac`
        # — the fields tracked since round 1 — carry the EXECUTED work since round 6 (on the reference-equivalent
        # count a removed duplicate read as a utilisation gain); the reference-equivalent figures keep their own names.
        saved = sum(shared_prefix_tflop(e_ - s_, H * W) for s_, e_ in ranges) if shared_on else 0.0
        out["cfg_shared_prefix"] = shared_on
        out["tflop_per_step_reference_equivalent"] = round(tf_step, 2)
        out["tflop_per_step_executed"] = round(tf_step - saved, 2)
        out["path_tflops_per_gpu"] = out["path_tflops_executed_per_gpu"] = round((tf_step - saved) * args.steps / dt, 2)
        out["path_mfma_frac"] = out["path_mfma_frac_executed"] = round((tf_step - saved) * args.steps / dt / PEAK_MFMA_TFLOPS, 4)
        out["path_tflops_reference_equivalent_per_gpu"] = round(tf_step * args.steps / dt, 2)
        out["path_mfma_frac_reference_equivalent"] = round(tf_step * args.steps / dt / PEAK_MFMA_TFLOPS, 4)
        if box.get("mfma_probe_tflops"):
            # the step against what THIS box's matrix pipe sustains on the probe stream: comparable across boxes
            out["box"]["step_tflops_over_probe"] = round((tf_step - saved) * args.steps / dt / box["mfma_probe_tflops"], 4)
        if dt_dup > 0:
            out["ms_per_step_full_duplicate"] = round(1e3 * dt_dup / args.steps, 3)
            out["path_mfma_frac_full_duplicate"] = round(tf_step * args.steps / dt_dup / PEAK_MFMA_TFLOPS, 4)
        if store is not None:
            # how the parameter gathers travelled, whether the peer mapping reproduced the collective (world > 1 only), and
            # what they cost the HOST: gathers per step, copies per gather, host time spent enqueueing them per step
            out["shard_transport"] = "rccl-c-abi" if store.comm is not None else store.transport
            out["shard_peer_self_check"] = store.peer_self_check
            out["shard_gathers_per_step"] = round((g1 - g0) / args.steps, 1)
            out["shard_copies_per_gather"] = (store.rehearse_copies or store.world) if store.transport == "peer" else None
            out["shard_gather_host_ms_per_step"] = round(1e3 * (h1 - h0) / args.steps, 3)
            out["persistent_grid_reserved_cus"] = ops.reserved_cus()
            out["shard_prefetch_depth"] = store.prefetch_depth
            if store.rehearse_hog is not None:
                out["rehearse_hog"] = {"cus_held": store.rehearse_hog[0], "lds_bytes": store.rehearse_hog[1],
                                       "modelled_allgather_gbs": store.rehearse_hog[2], "as_world": store.rehearse_hog[3]}
            if args.profile_all:
                out.update(copy_activity_of_a_step(step, args.warmup + args.steps, lats))
        if args.rehearsal or args.rehearse_dist:
            out["rehearsal"] = True      # not the driver's measurement: one-GPU rehearsal of the distributed path / dev flags
        if plan_world != world and args.rehearse_dist and my_frames != 24:
            # The 1 -> N scaling target in one number, on ONE box and in ONE process (boxes of the pool differ by several
            # per cent): N ranks each denoise a `my_frames`-frame window with sharded weights while one GPU alone takes
            # t24 for the 24-frame clip; useful frames per step: N x 12 against 24.  The monolithic step
            # is timed here on a second, resident copy of the same weights, after the rehearsal's memory peak was read.
            ops.set_reserved_cus(0)                               # one GPU alone runs no collective: full grids
            unet24 = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
            lat24 = seeded_noise((1, 4, 24, H, W), sched.init_noise_sigma, dev)

            def step24(i, lat):
                t = ts[i % len(ts)]
                noise = unet24(ops.cfg_input(lat, None, 0.0), t, encoder_hidden_states=emb).sample
                return sched.step_cfg(noise, t, lat, 7.5)

            for i in range(max(args.warmup, 1)):
                lat24 = step24(i, lat24)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for i in range(args.steps):
                lat24 = step24(i, lat24)
            torch.cuda.synchronize()
            t24 = (time.perf_counter() - t1) / args.steps
            tw = dt / args.steps
            out[f"t{my_frames}_sharded_ms"] = round(1e3 * tw, 3)
            out["t24_resident_ms"] = round(1e3 * t24, 3)
            useful_per_rank = T / plan_world                      # frames of the video per rank and step
            out["scaling_1_to_N_projected"] = round(plan_world * (useful_per_rank / 24.0) * t24 / tw, 3)
            out["scaling_note"] = (f"{plan_world} ranks x {useful_per_rank:g} useful frames per step at t{my_frames}_sharded against 24 frames "
                                   f"at t24_resident: {plan_world} x ({useful_per_rank:g} / 24) x t24 / t{my_frames}s; same process, same box")
            del unet24
        if plan_world != world and hasattr(unet.W, "shards"):
            # one-GPU rehearsal of a rank of a bigger job: here the whole of every unit stays resident (world of one);
            # at the planned world size only 1/plan_world of the sharded bytes would.  Estimate, not a measurement.
            sharded = sum(t.numel() * t.element_size() for t in unet.W.shards.values()) / 2 ** 30
            est = peak_gb - sharded * (1.0 - 1.0 / plan_world)
            out["peak_hbm_gb_per_gpu_estimated_at_planned_world"] = round(est, 3)
            if mono:
                out["peak_hbm_frac_of_monolithic_estimated_at_planned_world"] = round(est / mono, 4)
        if prof:
            torch.cuda.synchronize()
            agg = {}
            shapes = {}
            for name, flops, e0, e1, mnk in prof:
                ms_ = e0.elapsed_time(e1)
                for d_, k_ in ((agg, name), (shapes, (name.split("_")[0][:5] + " " + (name.split("<")[1][:14] if "<" in name else name[:14]),) + mnk)):
                    a = d_.setdefault(k_, [0.0, 0.0, 0])
                    a[0] += flops
                    a[1] += ms_
                    a[2] += 1
            name, (fl, ms, n) = max(agg.items(), key=lambda kv: kv[1][1])
            ach = fl / (ms * 1e-3) / 1e12
            traffic, step_traffic, note = pmc_traffic(name) if cname == "cfg2" else (None, None, "PMC profile is of cfg2")
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": traffic, "kernel": name,
                               "launches": n, "avg_launch_ms": round(ms / n, 4),
                               "algorithmic_gflop_per_launch": round(fl / n / 1e9, 2)}
            if note:
                out["roofline"]["traffic_note"] = note
            out["hbm_traffic_bytes_per_step"] = step_traffic     # sum of PMC read+write over every kernel of one forward
            # (the matrix kernels with per-launch events: every GEMM family + flash attention; key kept from round 1)
            out["gemm_kernels"] = {k: {"launches": v[2], "ms": round(v[1], 2), "tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1)}
                                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
            out["gemm_kernels_complete"] = bool(args.profile_all)      # false: only the dominant-kernel candidates were timed
            if args.profile_all:
                out["gemm_ms_per_step"] = round(sum(v[1] for k, v in agg.items() if k.startswith("gemm")) / args.steps, 2)
                out["flash_ms_per_step"] = round(sum(v[1] for k, v in agg.items() if k.startswith("flash")) / args.steps, 2)
            if args.shapes:
                out["gemm_shapes_ms_per_step"] = [
                    [" ".join(str(x) for x in k), v[2] // args.steps, round(v[1] / args.steps, 2),
                     round(v[0] / (v[1] * 1e-3) / 1e12)]
                    for k, v in sorted(shapes.items(), key=lambda kv: -kv[1][1])[:args.shapes]]
        else:
            ach = tf_step * args.steps / dt
            out["roofline"] = {"bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / PEAK_MFMA_TFLOPS, 4), "traffic": None, "kernel": "whole step"}
        if world == 1 and args.cpu_frames > 0 and not args.rehearse_dist and cname == "cfg2":
            hip_y = sd_cpu = None
            if not args.peaked:
                # the HIP forward the oracle sample is compared with: this process's UNet on the sample's inputs
                lat_s, e_s = cpu_baseline_inputs(args.cpu_frames)
                unet.share_cfg_prefix = True
                hip_y = unet(ops.cfg_input(lat_s.to(dev), None, 0.0), 981, encoder_hidden_states=e_s.to(dev)).sample.float().cpu()
                sd_cpu = synthetic_state_dict(cfg, 1234, dev)      # the table `unet` was loaded from (same generator, same seed)
            del unet
            torch.cuda.empty_cache()
            ncpu = min(len(os.sched_getaffinity(0)), 16)      # the 1-GPU box's CPU share
            out["cpu_baseline"] = cpu_baseline(args.cpu_frames, ncpu, sd_cpu, hip_y)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist_mode:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
