"""Host tests (CPU) of the halo exchange + owned-frame blend that replaces the reference's
`all_gather_object` + full blend (`fsdp_chunked_coherent.py:190-217`): frame ownership, the transfer list, and
bit-identity of the owned frames with the oracle's `ramp_blend` of ALL chunks in the reference's rank-major order —
single-process simulation of every rank over a sweep of plans, then real point-to-point transfers with `gloo`
(world 2 and 4) through `DistributedVideoDiffuser.__call__`.

The HIP blend kernels cannot run here; the two blend ops are replaced by their torch-CPU statements (the oracle's
expressions, applied per frame segment).  The kernels themselves are checked bit-exact on the GPU
(tests/test_ops_gpu.py::test_blend_bit_exact, tests/test_halo_gpu.py)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import vdx  # noqa: E402,F401
from vdx import ops, pipeline  # noqa: E402
from vdx.pipeline import HaloPlan, blend_owned  # noqa: E402
from vdx.planner import PlannerError, plan  # noqa: E402
from oracle.pipeline_ref import ramp_blend  # noqa: E402


def cpu_blend_accumulate(full, weight, chunk, w, s, e):
    full[:, :, s:e] += chunk * w.view(1, 1, -1, 1, 1)          # fp16 accumulator += fp16 * fp32 (:214)
    weight[s:e] += w                                           # (:215)


def cpu_blend_finalize(full, weight):
    return full / weight.view(1, 1, -1, 1, 1).clamp(min=1e-6)  # (:217)


@pytest.fixture()
def cpu_blend(monkeypatch):
    monkeypatch.setattr(ops, "blend_accumulate", cpu_blend_accumulate)
    monkeypatch.setattr(ops, "blend_finalize", cpu_blend_finalize)


SWEEP = [(T, W, cs, ov, nc) for T in (8, 24, 31, 32, 48, 96) for W in (1, 2, 3, 4, 8)
         for cs, ov, nc in ((0, 4, False), (8, 2, False), (6, 4, False), (10, 7, False), (0, 0, True))]


@pytest.mark.parametrize("T,W,cs,ov,nc", SWEEP)
def test_ownership_partitions_the_video_and_transfers_are_the_halos(T, W, cs, ov, nc):
    try:
        cp = plan(T, W, cs, ov, no_chunking=nc)
    except PlannerError:
        pytest.skip("the reference hangs on this input")
    hp = HaloPlan(cp, T)
    owned = sorted(r for rs in hp.owned.values() for r in rs)
    assert owned[0][0] == 0 and owned[-1][1] == T
    assert all(a[1] == b[0] for a, b in zip(owned[:-1], owned[1:])), owned          # a partition of [0, T)
    for rank in range(W):
        segs = hp.segments[rank]
        assert sorted((g.s, g.e) for g in segs) == sorted((g.s, g.e) for g in segs)
        for g in segs:
            cover = {j for j, (s, e) in enumerate(cp.ranges) if s < g.e and e > g.s}
            assert set(g.chunks) == cover and all(cp.ranges[j][0] <= g.s and cp.ranges[j][1] >= g.e for j in cover)
            # the reference's order: rank-major, then the rank's own order (:208-209)
            assert list(g.chunks) == sorted(g.chunks, key=lambda j: (j % W, j // W))
        need = {(j, g.s, g.e) for g in segs for j in g.chunks if j % W != rank}
        have = {(t.chunk, a, b) for t in hp.transfers if t.dst == rank for (a, b) in
                [(g.s, g.e) for g in segs if t.chunk in g.chunks and t.s <= g.s and g.e <= t.e]}
        assert need == have
    assert all(t.src == t.chunk % W and t.src != t.dst for t in hp.transfers)
    if not nc and W > 1 and cp.per_rank == 1 and len(set(cp.ranges)) == len(cp.ranges) and 2 * cp.overlap <= cp.chunk:
        # one window per rank: a rank sends exactly its trailing overlap frames to its successor
        assert [(t.src, t.dst, t.e - t.s) for t in hp.transfers] == [(r, r + 1, cp.ranges[r][1] - cp.ranges[r + 1][0])
                                                                    for r in range(W - 1)]


@pytest.mark.parametrize("T,W,cs,ov,nc", SWEEP)
def test_owned_blend_equals_reference_blend_bitwise(cpu_blend, T, W, cs, ov, nc):
    """Every rank simulated in one process: the pieces a rank would receive are sliced from the sender's chunk."""
    try:
        cp = plan(T, W, cs, ov, no_chunking=nc)
    except PlannerError:
        pytest.skip("the reference hangs on this input")
    C, H, Wd = 2, 3, 4
    g = torch.Generator().manual_seed(T * 100 + W)
    chunks = {i: (torch.randn(1, C, e - s, H, Wd, generator=g) * 3).half() for i, (s, e) in enumerate(cp.ranges)}
    like = torch.zeros(1, C, T, H, Wd, dtype=torch.float16)
    gathered = [(cp.ranges[i][0], cp.ranges[i][1], chunks[i]) for r in range(W) for i in range(len(cp.ranges)) if i % W == r]
    want = ramp_blend(gathered, T, cp.overlap, like)
    hp = HaloPlan(cp, T)
    got_full = torch.full_like(want, float("nan"))
    for rank in range(W):
        mine = [chunks[i] for i in range(len(cp.ranges)) if i % W == rank]
        recv = {(t.chunk, t.s, t.e): chunks[t.chunk][:, :, t.s - cp.ranges[t.chunk][0]:t.e - cp.ranges[t.chunk][0]].clone()
                for t in hp.transfers if t.dst == rank}
        for s, e, lat in blend_owned(mine, hp, recv, None, like, rank):
            assert lat.dtype == torch.float32
            got_full[:, :, s:e] = lat
    assert torch.equal(got_full, want)
    if cp.overlap > 0:
        assert got_full[:, :, 0].abs().max() == 0 and got_full[:, :, -1].abs().max() == 0      # reference quirk (a9)


def test_halo_traffic_is_the_overlap_only():
    """BASELINE cfg5 at XL size: 4 overlap frames = 294 912 B to one neighbour, against 7 x 1 179 648 B received per
    rank by the all-gather."""
    cp = plan(96, 8, 0, 4)
    hp = HaloPlan(cp, 96)
    frame = 4 * 72 * 128 * 2
    assert [hp.bytes_sent(r, frame) for r in range(8)] == [294912] * 7 + [0]


GLOO_SCRIPT = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
sys.path.insert(0, os.path.join({root!r}, "tests"))
import vdx
from vdx import ops
from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser
from vdx.scheduler import DDIMScheduler
from oracle.pipeline_ref import base_noise, plan_chunks, ramp_blend
from test_halo_host import cpu_blend_accumulate, cpu_blend_finalize
ops.blend_accumulate, ops.blend_finalize = cpu_blend_accumulate, cpu_blend_finalize
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()

def stub(lat):          # stands in for 50 UNet + DDIM steps: any deterministic function of the chunk's content
    x = lat.float()
    return (0.5 * x + 0.25 * torch.roll(x, 1, dims=2) - 0.1 * x.mean(dim=2, keepdim=True)).half()

class Unet:             # only .config.in_channels is used outside denoise()
    class config: in_channels = 4
    W = None

for T, cs, ov, mode in ((24, 0, 4, "hybrid_ctx"), (24, 8, 2, "hybrid"), (31, 10, 7, "hybrid_ctx"), (12, 0, 4, "fsdp")):
    cfg = DiffuserConfig(num_frames=T, steps=2, chunk_size=cs, overlap=ov, height=32, width=48, mode=mode,
                         device="cpu", noise_device="cpu")
    d = DistributedVideoDiffuser(cfg, Unet(), DDIMScheduler(), None, None)
    d.denoise = lambda lat: stub(lat)
    # what the N-rank reference job computes (oracle, serial): rank-major gathered list, full blend
    c, o, ranges = plan_chunks(T, world, cs, ov, mode == "fsdp")
    base = base_noise(T, 4, 4, 6)
    gathered = [(s, e, stub(base[:, :, s:e].clone())) for r in range(world) for i, (s, e) in enumerate(ranges) if i % world == r]
    want = ramp_blend(gathered, T, o, base)
    if mode == "hybrid_ctx":
        assert torch.equal(d.ctx, base.mean(dim=2, keepdim=True))
    full, info = d(exchange="allgather")
    assert (info["chunk_size"], info["overlap"], [tuple(r) for r in info["ranges"]]) == (c, o, [tuple(r) for r in ranges])
    assert torch.equal(full, want), ("allgather", T, cs, ov, mode)
    owned, info = d(exchange="halo")
    for s, e, lat in owned:
        assert torch.equal(lat, want[:, :, s:e]), ("halo", T, cs, ov, mode, s, e)
    # every frame is owned by exactly one rank
    counts = torch.zeros(T, dtype=torch.int64)
    for s, e, _ in owned:
        counts[s:e] += 1
    dist.all_reduce(counts)
    assert bool((counts == 1).all()), counts
    nb = torch.tensor([info["network_bytes"]])
    dist.all_reduce(nb)
    if rank == 0:
        print("case", T, cs, ov, mode, "halo bytes", int(nb), "owned", info["owned"])
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok")
"""


@pytest.mark.parametrize("world,port", [(2, 29641), (4, 29643)])
def test_whole_pipeline_gloo(tmp_path, world, port):
    """plan -> shared noise -> ctx broadcast -> denoise (stub) -> exchange -> blend with `world` processes:
    all-gather path and halo path both equal the oracle's serial statement of the N-rank job, bit for bit."""
    script = tmp_path / "pipe.py"
    script.write_text(GLOO_SCRIPT.format(root=ROOT))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(script)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world
