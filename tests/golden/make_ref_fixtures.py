"""Runs the REFERENCE'S OWN orchestration code — unmodified, from where it lies under /root/reference — in the build
container and records what it computes, as fixtures under tests/golden/ref_exec_*.  BUILD CONTAINER ONLY: this script is
the only place the reference's path appears; nothing of the reference travels (the fixtures are inputs / outputs).

    python tests/golden/make_ref_fixtures.py            # everything (about two minutes on 8 cores)
    python tests/golden/make_ref_fixtures.py planner    # only the planner sweep

What is executed: `Distribution/strategies/{fsdp_chunked_coherent,fsdp_chunked,chunk_only,fsdp}.py`, each loaded with
`runpy.run_path(<file>, run_name="__main__")` so that its module body, `main()`, `DistributedVideoDiffuser.__init__`,
`_denoise` and `__call__` run as written (argparse -> planner -> shared noise -> [ctx] -> CFG/DDIM loop -> gather ->
ramp blend -> per-frame decode -> boundary metric -> CSV row).  World sizes 1, 2, 3, 4 and 8 as real processes over gloo.

What stands in for what the container lacks (stated once, here; DESIGN.md §2 repeats it):
  * `diffusers` is ABSENT and stays absent.  The script installs a module of that name whose
    `DiffusionPipeline.from_pretrained` returns a plain object carrying `.unet .text_encoder .vae .tokenizer .scheduler`:
    the build's fp32 oracle UNet behind fp16 tensors (or, for the planner sweep, a two-line stand-in), `DDIMSchedulerRef`,
    a fixed text-embedding table, and a recording VAE.  So rows a5 / a6 (the diffusers arithmetic) are NOT pinned by this —
    what is pinned is the code the reference itself owns: a1 planner + chunk -> rank, a2 shared noise + slicing, a3 global
    context, a4 the `_denoise` call sequence and its fp16 tensor arithmetic, a8 the gather order, a9 the ramp blend, and
    the CSV row / boundary metric of a10.
  * `pynvml`, `cv2`: no-op stand-ins (memory readings 0; Farneback flow = zero field, so `flow_err` is not a fixture).
  * `torch.cuda.{set_device,current_device,empty_cache,reset_peak_memory_stats,max_memory_allocated}`: no-ops (CPU host);
    `--device cpu` is passed on the command line the scripts already have.
  * `dist.init_process_group("nccl", ...)` is redirected to gloo over a file store.
  * `FullyShardedDataParallel(module, **kw)` returns `module` (numerically that is what FSDP inference is) and the keyword
    arguments the reference passes are recorded (row a7's wrap configuration).
Configurations on which the reference's planner loops forever (`i += sz - ov` with ov >= sz) are recorded as
"hang" from that arithmetic and NOT executed.
"""
import json
import os
import runpy
import subprocess
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/Distribution/strategies"
FILES = {"coherent": "fsdp_chunked_coherent.py", "fsdp_chunked": "fsdp_chunked.py", "chunk_only": "chunk_only.py", "fsdp": "fsdp.py"}
TINY = dict(ch=(64, 128, 128, 128), cross=128, in_heads=2)      # == tests/ref_exec_standins.TINY
WORLDS = (1, 2, 3, 4, 8)


# ------------------------------------------------------------------------------------------------------------------
# job lists
# ------------------------------------------------------------------------------------------------------------------
def planner_jobs(world):
    """(file, mode, T, chunk_size, overlap) — the sweep of (T, world, chunk, overlap, file) VERDICT r4 asks for, including
    every BASELINE configuration and SURVEY a1's known answers."""
    jobs = []
    Ts = (8, 12, 16, 17, 20, 24, 32, 48, 96)
    for T in Ts:
        if T < world:
            continue
        for chunk in (0, 6, 10, 16):
            for ov in (0, 2, 4, 5):
                jobs.append(("coherent", "hybrid", T, chunk, ov))
                if chunk in (0, 10):
                    jobs.append(("fsdp_chunked", "-", T, chunk, ov))
                    jobs.append(("chunk_only", "-", T, chunk, ov))
        jobs.append(("coherent", "fsdp", T, 0, 4))
        jobs.append(("coherent", "hybrid_ctx", T, 0, 4))
        jobs.append(("coherent", "chunk", T, 0, 4))
    return jobs


def planner_hangs(file, mode, T, world, chunk, ov_arg):
    """ov >= cs on the first `compute_chunks` call: `i += sz - ov` never advances (fsdp_chunked_coherent.py:160-165)."""
    if file == "coherent" and mode == "fsdp":
        return False
    if chunk <= 0:
        cs = min(min(16, T // world), max(max(4, T // (world * 2)), T // world))
    else:
        cs = chunk
    ov = (ov_arg if ov_arg > 0 else max(4, cs // 3)) if file == "coherent" else min(ov_arg, cs // 3)
    return ov >= cs


NUMERIC = {
    # name: (file, mode, world, T, latent h = w, steps, chunk, overlap, UNet stand-in)
    # "exact": the elementwise stand-in (tests/ref_exec_standins.ExactUNet) — every downstream tensor is compared BIT FOR BIT
    "exact_hybrid_ctx_w2": ("coherent", "hybrid_ctx", 2, 20, 16, 5, 0, 4, "exact"),
    "exact_hybrid_ctx_w3": ("coherent", "hybrid_ctx", 3, 32, 8, 50, 0, 4, "exact"),      # the reference's default job: 32 frames, 50 steps
    "exact_hybrid_w4": ("coherent", "hybrid", 4, 48, 8, 4, 0, 4, "exact"),                # BASELINE cfg4's plan
    "exact_hybrid_ctx_w8": ("coherent", "hybrid_ctx", 8, 96, 8, 3, 0, 4, "exact"),        # BASELINE cfg5's plan (last window 12 frames)
    "exact_chunk_w1": ("coherent", "chunk", 1, 12, 16, 4, 6, 2, "exact"),
    "exact_chunk_only_cfg1": ("chunk_only", "-", 1, 8, 32, 10, 0, 4, "exact"),            # BASELINE cfg1: 8 frames @256x256, 10 steps
    "exact_fsdp_chunked_w2": ("fsdp_chunked", "-", 2, 14, 16, 3, 0, 4, "exact"),
    "exact_fsdp_mode_w2": ("coherent", "fsdp", 2, 6, 16, 3, 0, 4, "exact"),               # BASELINE cfg3's mode: both ranks the whole clip
    # fsdp.py — the strategy file BASELINE cfg3 names (FSDPBenchmark: its own loop :141-153, literal 7.5, no scale_model_input,
    # UNSEEDED noise: the harness seeds torch's global generator before the script starts; decode through pipe.decode_latents)
    "exact_fsdp_file_w2": ("fsdp", "-", 2, 8, 16, 10, 0, 0, "exact"),
    # "oracle": the fp32 oracle UNet behind fp16 tensors — float arithmetic, compared within a stated tolerance
    "oracle_hybrid_ctx_w2": ("coherent", "hybrid_ctx", 2, 20, 16, 3, 0, 4, "oracle"),
    "oracle_chunk_only_cfg1": ("chunk_only", "-", 1, 8, 16, 10, 0, 4, "oracle"),
}


# ------------------------------------------------------------------------------------------------------------------
# worker: one rank
# ------------------------------------------------------------------------------------------------------------------
class Recorder:
    def __init__(self):
        self.reset()

    def reset(self):
        self.unet, self.gathered, self.z, self.fsdp_kwargs, self.ctx = None, None, [], [], None


REC = Recorder()


sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
from ref_exec_standins import ExactUNet, oracle_unet, standin_decode, text_table  # noqa: E402


class TextEncoder(torch.nn.Module):
    def __init__(self, cross):
        super().__init__()
        self.table = torch.nn.Parameter(text_table(cross), requires_grad=False)

    def forward(self, ids):
        assert tuple(ids.shape) == (2, 77)
        return (self.table.data.clone(),)


class Tokenizer:
    model_max_length = 77

    def __call__(self, texts, padding=None, max_length=None, truncation=None, return_tensors=None):
        assert len(texts) == 2 and texts[1] == "" and max_length == 77 and padding == "max_length" and return_tensors == "pt"
        return types.SimpleNamespace(input_ids=torch.zeros(2, 77, dtype=torch.long))


class RecordingVAE(torch.nn.Module):
    """`vae.decode(z / 0.18215).sample` (:223): records z — the blended latent frame as the reference hands it over — and
    returns a cheap deterministic image so that the frame mapping and the boundary L1 metric run on non-trivial data."""

    def __init__(self):
        super().__init__()
        self.decoder = torch.nn.Linear(1, 1)                            # a child with trainable parameters: :85-87 wraps it

    def decode(self, z):
        REC.z.append(z.detach().clone())
        return types.SimpleNamespace(sample=standin_decode(z))


KIND = ["exact"]


def install_stand_ins(tmp, seq):
    """Everything the module body and `__init__` reach for that this container lacks (module docstring)."""
    import torch.distributed as dist
    import torch.distributed.fsdp as fsdp_mod
    sys.path.insert(0, ROOT)
    from oracle.ddim_ref import DDIMSchedulerRef

    def from_pretrained(model_id, **kw):
        unet = oracle_unet() if KIND[0] == "oracle" else ExactUNet()
        REC.unet = unet
        cross = TINY["cross"]
        pipe = types.SimpleNamespace(unet=unet, text_encoder=TextEncoder(cross), vae=RecordingVAE(), tokenizer=Tokenizer(),
                                     scheduler=DDIMSchedulerRef())
        pipe.to = lambda dev: pipe

        def decode_latents(lat):                        # fsdp.py:172 — (1,4,1,h,w) -> (1,3,1,H,W)
            REC.z.append(lat[:, :, 0].detach().clone())
            return standin_decode(lat[:, :, 0]).unsqueeze(2)
        pipe.decode_latents = decode_latents
        pipe.from_pretrained_kwargs = {k: str(v) for k, v in kw.items()}
        REC.pipe = pipe
        return pipe

    import huggingface_hub
    if not hasattr(huggingface_hub, "HfFolder"):        # fsdp.py:26 imports a name newer huggingface_hub versions dropped; it never uses it
        huggingface_hub.HfFolder = type("HfFolder", (), {})
    d = types.ModuleType("diffusers")
    d.DiffusionPipeline = types.SimpleNamespace(from_pretrained=from_pretrained)
    sys.modules["diffusers"] = d
    nv = types.ModuleType("pynvml")
    nv.nvmlInit = lambda: None
    nv.nvmlDeviceGetHandleByIndex = lambda i: i
    nv.nvmlDeviceGetMemoryInfo = lambda h: types.SimpleNamespace(used=0)
    sys.modules["pynvml"] = nv
    cv = types.ModuleType("cv2")
    cv.COLOR_BGR2GRAY, cv.COLOR_RGB2BGR, cv.INTER_LINEAR = 6, 4, 1
    cv.cvtColor = lambda f, code: f.mean(axis=2).astype(np.uint8) if code == 6 else f[..., ::-1]
    cv.calcOpticalFlowFarneback = lambda a, b, *r: np.zeros(a.shape + (2,), np.float32)
    cv.remap = lambda f, mx, my, interp: f
    cv.VideoWriter_fourcc = lambda *a: 0
    cv.VideoWriter = lambda path, *a: types.SimpleNamespace(write=lambda f: None, release=lambda: open(path, "wb").close())
    sys.modules["cv2"] = cv

    torch.cuda.set_device = lambda *a, **k: None
    torch.cuda.current_device = lambda: 0
    torch.cuda.empty_cache = lambda: None
    torch.cuda.reset_peak_memory_stats = lambda *a: None
    torch.cuda.max_memory_allocated = lambda *a: 0

    real_init = dist.init_process_group
    if not getattr(dist, "_ref_exec_patched", False):
        def init_pg(backend=None, **kw):
            seq[0] += 1
            real_init("gloo", init_method=f"file://{tmp}/rdzv_{seq[0]}", rank=int(os.environ["RANK"]),
                      world_size=int(os.environ["WORLD_SIZE"]))
        dist.init_process_group = init_pg
        real_ago = dist.all_gather_object

        def ago(gathered, obj, *a, **k):
            real_ago(gathered, obj, *a, **k)
            REC.gathered = [[(s, e, t.clone()) for s, e, t in lst] for lst in gathered]
        dist.all_gather_object = ago
        real_bc = dist.broadcast

        def bc(t, src=0, *a, **k):
            real_bc(t, src, *a, **k)
            REC.ctx = t.detach().clone()
        dist.broadcast = bc

        def fsdp(module, **kw):
            REC.fsdp_kwargs.append((type(module).__name__, {k: (v.__name__ if callable(v) and hasattr(v, "__name__") else str(v))
                                                             for k, v in kw.items()}))
            pol = kw.get("auto_wrap_policy")
            if pol is not None:       # the reference's wrap_policy (:64-66) on three probes
                REC.wrap_policy = [bool(pol(torch.nn.ModuleList(), True, 20_000_000)), bool(pol(torch.nn.Linear(1, 1), True, 10_000_000)),
                                   bool(pol(torch.nn.Linear(1, 1), True, 9_999_999))]
            return module
        fsdp_mod.FullyShardedDataParallel = fsdp
        dist._ref_exec_patched = True


def run_reference(file, argv, cwd):
    """The reference file as `__main__`, unmodified."""
    old_argv, old_cwd = sys.argv, os.getcwd()
    sys.argv = [FILES[file]] + argv
    os.chdir(cwd)
    try:
        runpy.run_path(os.path.join(REF, FILES[file]), run_name="__main__")
    finally:
        sys.argv = old_argv
        os.chdir(old_cwd)


def argv_for(file, mode, T, chunk, ov, hw, steps, csv):
    if file == "fsdp":
        return ["--num_frames", str(T), "--steps", str(steps), "--height", str(hw * 8), "--width", str(hw * 8), "--device", "cpu",
                "--out_csv", csv, "--model_id", "stand-in"]
    a = ["--num_frames", str(T), "--steps", str(steps), "--chunk_size", str(chunk), "--overlap", str(ov), "--height", str(hw * 8),
         "--width", str(hw * 8), "--device", "cpu", "--out_csv", csv, "--model_id", "stand-in"]
    if file == "coherent":
        a += ["--mode", mode]
    return a


def ranges_from_gathered(g):
    per = len(g[0])
    return [[int(g[r][k][0]), int(g[r][k][1])] for k in range(per) for r in range(len(g))]


def worker(kind, outdir):
    import logging
    import resource
    resource.setrlimit(resource.RLIMIT_AS, (12 << 30, 12 << 30))       # a runaway planner loop must die, not the container
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.set_num_threads(max(1, 8 // world))
    tmp = os.environ["REF_EXEC_TMP"]
    seq = [0]
    logging.disable(logging.WARNING)
    os.environ["TQDM_DISABLE"] = "1"
    if kind == "planner":
        install_stand_ins(tmp, seq)
        rows = []
        for file, mode, T, chunk, ov in planner_jobs(world):
            row = {"file": file, "mode": mode, "T": T, "world": world, "chunk_size": chunk, "overlap": ov}
            if planner_hangs(file, mode, T, world, chunk, ov):
                row["hang"] = True
                rows.append(row)
                continue
            REC.reset()
            csv = os.path.join(tmp, f"p_{rank}.csv")
            if os.path.exists(csv):
                os.remove(csv)
            run_reference(file, argv_for(file, mode, T, chunk, ov, 2, 1, csv), tmp)
            row.update(ranges=ranges_from_gathered(REC.gathered))
            if rank == 0:
                hdr, vals = open(csv).read().strip().split("\n")
                d = dict(zip(hdr.split(","), vals.split(",")))
                row.update(cs=int(d["chunk_size"]), ov=int(d["overlap"]), network_bytes=int(d["network_bytes"]), csv_mode=d["mode"])
            rows.append(row)
        if rank == 0:
            json.dump(rows, open(os.path.join(outdir, f"planner_w{world}.json"), "w"))
        return
    # numeric jobs of this world size
    install_stand_ins(tmp, seq)
    for name, (file, mode, w, T, hw, steps, chunk, ov, kind_) in NUMERIC.items():
        if w != world:
            continue
        REC.reset()
        KIND[0] = kind_
        csv = os.path.join(tmp, f"n_{name}_{rank}.csv")
        if file == "fsdp":
            torch.manual_seed(4321)                     # fsdp.py draws its noise unseeded (:133): fix the generator it draws from
        run_reference(file, argv_for(file, mode, T, chunk, ov, hw, steps, csv), tmp)
        if rank != 0:
            continue
        if file == "fsdp":
            hdr, vals = open(csv).read().strip().split("\n")
            d = dict(zip(hdr.split(","), vals.split(",")))
            out = {"csv_header": np.array(hdr), "csv_mode": np.array(d["mode"]), "cs": int(d["chunk_size"]), "ov": int(d["overlap"]),
                   "network_bytes": int(d["network_bytes"]), "csv_temp_instab": np.array(d["temp_instab"]), "T": T, "hw": hw, "steps": steps,
                   "world": world, "ref_file": np.array(FILES[file]), "mode": np.array(mode), "unet_kind": np.array(kind_),
                   "z": torch.stack([z[0] for z in REC.z]).numpy(),                  # the final latent, frame by frame, as handed to decode_latents (fp32)
                   "x_first": REC.unet.calls[0][1].numpy(), "t_first": REC.unet.calls[0][0],
                   "timesteps": np.array([t for t, _ in REC.unet.calls[:steps]]), "unet_calls": len(REC.unet.calls),
                   "fsdp_kwargs": np.array(json.dumps(REC.fsdp_kwargs)), "seed": 4321}
            np.savez_compressed(os.path.join(outdir, f"ref_exec_{name}.npz"), **out)
            print(f"{name}: {out['unet_calls']} UNet calls, csv mode {d['mode']}", flush=True)
            continue
        hdr, vals = open(csv).read().strip().split("\n")
        d = dict(zip(hdr.split(","), vals.split(",")))
        out = {"csv_header": np.array(hdr), "csv_mode": np.array(d["mode"]), "cs": int(d["chunk_size"]), "ov": int(d["overlap"]),
               "network_bytes": int(d["network_bytes"]), "temp_instab": float(d["temp_instab"]) if d["temp_instab"] else np.nan,
               "ranges": np.array(ranges_from_gathered(REC.gathered)), "T": T, "hw": hw, "steps": steps, "world": world,
               "chunk_size_arg": chunk, "overlap_arg": ov, "ref_file": np.array(FILES[file]), "mode": np.array(mode),
               "z": torch.stack([z[0] for z in REC.z]).numpy(),                      # (T, C, h, w): lat[:, :, i] / 0.18215 as handed to the VAE
               "x_first": REC.unet.calls[0][1].numpy(), "t_first": REC.unet.calls[0][0], "unet_kind": np.array(kind_),
               "timesteps": np.array([t for t, _ in REC.unet.calls[:steps]]), "unet_calls": len(REC.unet.calls),
               "fsdp_kwargs": np.array(json.dumps(REC.fsdp_kwargs)), "wrap_policy": np.array(getattr(REC, "wrap_policy", [])),
               "from_pretrained_kwargs": np.array(json.dumps(REC.pipe.from_pretrained_kwargs))}
        if REC.ctx is not None:
            out["ctx"] = REC.ctx.numpy()
        for r, lst in enumerate(REC.gathered):
            for k, (s, e, t) in enumerate(lst):
                out[f"den_r{r}_k{k}"] = t.numpy()                                    # the denoised chunk (1, C, e - s, h, w) fp16
        np.savez_compressed(os.path.join(outdir, f"ref_exec_{name}.npz"), **out)
        print(f"{name}: cs {out['cs']} ov {out['ov']} ranges {out['ranges'].tolist()} temp_instab {out['temp_instab']:.4f}", flush=True)


# ------------------------------------------------------------------------------------------------------------------
def launch(kind, world, outdir):
    with tempfile.TemporaryDirectory() as tmp:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), REF_EXEC_TMP=tmp, TQDM_DISABLE="1")
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", kind, outdir], env=env))
        rcs = [p.wait() for p in procs]
        if any(rcs):
            raise SystemExit(f"{kind} world {world}: worker exit codes {rcs}")


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "planner"):
        with tempfile.TemporaryDirectory() as out:
            rows = []
            for w in WORLDS:
                launch("planner", w, out)
                rows += json.load(open(os.path.join(out, f"planner_w{w}.json")))
            json.dump(rows, open(os.path.join(HERE, "ref_exec_planner.json"), "w"), separators=(",", ":"))
            print(f"planner: {len(rows)} configurations, {sum(1 for r in rows if r.get('hang'))} of them hang in the reference")
    if what in ("all", "numeric"):
        for w in sorted({v[2] for v in NUMERIC.values()}):
            launch("numeric", w, HERE)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--worker":
        worker(sys.argv[2], sys.argv[3])
    else:
        main()
