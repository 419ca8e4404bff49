"""Parameter shard store (vdx/shard.py): the build's counterpart of the reference's FSDP wrap
(fsdp_chunked_coherent.py:63-88).  CPU: 2-rank gloo run checks that every rank sees bit-identical
full tensors while holding half the bytes, in schedule order and out of order.  GPU: the tiny UNet
gives identical outputs with and without the store (world 1, side-stream prefetch exercised)."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

GLOO_SCRIPT = r"""
import sys, torch, torch.distributed as dist
sys.path.insert(0, {root!r})
import vdx
import vdx.shard
from vdx.shard import ShardedStore
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
from vdx.weights import synthetic_state_dict
vdx.shard.ARENA_BYTES = {arena_bytes}
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
cfg = UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)
m = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, seed=5), device="cpu")
full = dict(m.W)
total = sum(t.numel() for t in full.values()) * 2
m.shard_(rank, world, prefetch_depth={depth})
st = m.W
assert isinstance(st, ShardedStore) and set(st.keys()) == set(full) and len(st._bufs) == {depth} + 1
# schedule order (what forward does), twice (second pass uses the wrap-around prefetch)
by_unit = {{}}
for k in full:
    by_unit.setdefault(m.unit_of(k), []).append(k)
for _ in range(2):
    for u in m.unit_schedule():
        for k in by_unit[u]:
            assert torch.equal(st[k], full[k]), k
for k in by_unit[None]:
    assert st[k] is full[k]
n_sched = len(m.unit_schedule())
assert st.gathers <= 2 * n_sched + 2, (st.gathers, n_sched)
# out-of-order access falls back to an on-demand gather and is still exact
for u in list(reversed(m.unit_schedule()))[:7]:
    for k in by_unit[u]:
        assert torch.equal(st[k], full[k]), k
shard_bytes = sum(s.numel() for s in st.shards.values()) * 2
sharded_total = sum(full[k].numel() for u in m.unit_schedule() for k in by_unit[u]) * 2
assert shard_bytes <= sharded_total / world * 1.02 + 4096 * n_sched, (shard_bytes, sharded_total)
# the arenas: none above the cap (unless one unit alone is), every unit inside exactly one, shards are views of them
es = 2
assert len(st._arenas) >= {min_arenas}, len(st._arenas)
for a in st._arenas:
    assert a.numel() * es <= max(vdx.shard.ARENA_BYTES, max(st._padded.values()) // world * es)
for u, (ai, ao) in st._arena_off.items():
    n = st._padded[u] // world
    assert ao + n <= st._arenas[ai].numel() and st.shards[u].data_ptr() == st._arenas[ai].data_ptr() + ao * es
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok", st.gathers)
"""


@pytest.mark.parametrize("arena_bytes,min_arenas,depth,world", [(256 << 20, 1, 2, 2), (1 << 20, 4, 1, 2), (1 << 20, 4, 3, 2), (1 << 20, 3, 2, 3)],
                         ids=["one-arena-depth2", "many-arenas-depth1", "many-arenas-depth3", "three-ranks-depth2"])
def test_shard_store_two_ranks_gloo(tmp_path, arena_bytes, min_arenas, depth, world):
    """`many-arenas`: the cap lowered to 1 MB so that the tiny model's shards spread over several arenas — the layout the XL
    model has at its 256 MB cap (an exported allocation must stay under 1 GiB: vdx/shard.py)."""
    script = tmp_path / "shard.py"
    script.write_text(GLOO_SCRIPT.format(root=ROOT, arena_bytes=arena_bytes, min_arenas=min_arenas, depth=depth))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", "29631", str(script)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world


def test_unit_map_covers_every_packed_tensor():
    import vdx  # noqa: F401
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import state_dict_spec
    cfg = UNet3DConfig.zeroscope()
    m = UNet3DConditionModel(cfg)
    sd = {k: torch.empty(s, device="meta") for k, s in state_dict_spec(cfg).items()}
    m.load_diffusers_state_dict(sd, device="meta")
    sched = m.unit_schedule()
    assert len(sched) == len(set(sched)) == 83           # leaf units in forward order
    units, stem = {}, 0
    for k, v in m.W.items():
        u = m.unit_of(k)
        if u is None:
            stem += v.numel()
        else:
            assert u in sched, (k, u)
            units[u] = units.get(u, 0) + v.numel()
    assert set(units) == set(sched)
    assert stem < 30e6                                   # replicated stem: conv_in/out, time embedding
    assert max(units.values()) * 2 < 100 * 2 ** 20       # largest unit < 100 MiB fp16 (93.8 MiB)


@pytest.mark.gpu
def test_unet_through_shard_store_matches_unsharded(gpu):
    import vdx  # noqa: F401
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict
    cfg = UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)
    sd = synthetic_state_dict(cfg, seed=9, device=gpu)
    g = torch.Generator().manual_seed(3)
    x = torch.randn(2, 4, 4, 16, 32, generator=g).half().to(gpu)
    e = torch.randn(2, 77, 128, generator=g).half().to(gpu)
    a = UNet3DConditionModel(cfg).load_diffusers_state_dict(sd, device=gpu)
    want = a(x, 501, encoder_hidden_states=e).sample
    b = UNet3DConditionModel(cfg).load_diffusers_state_dict(sd, device=gpu).shard_(0, 1)
    for _ in range(3):                                   # repeated steps reuse the wrap-around prefetch
        got = b(x, 501, encoder_hidden_states=e).sample
        assert torch.equal(got, want)
    assert b.W.gathers <= 3 * len(b.unit_schedule()) + 2


@pytest.mark.gpu
def test_peer_transport_world_of_one(gpu):
    """The peer transport of the shard store with one rank: the arena is exported (vdx_ipc_export succeeds on the
    caching allocator's block), every unit is gathered through vdx_peer_gather on the side stream, the UNet output keeps
    its bits.  (Mapping another process's arena: tests/test_dist_gpu.py.)"""
    import vdx  # noqa: F401
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict
    cfg = UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)
    m = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, seed=5, device=gpu), device=gpu)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 3, 16, 16, generator=g).half().to(gpu)
    e = torch.randn(2, 77, 128, generator=g).half().to(gpu)
    want = m(x, 501, encoder_hidden_states=e).sample
    m.shard_(0, 1, transport="peer")
    assert m.W.transport == "peer"
    for _ in range(2):
        got = m(x, 501, encoder_hidden_states=e).sample
        assert torch.equal(got, want)
    assert m.W.gathers > 10


@pytest.mark.gpu
def test_native_rccl_entry_points_world_of_one(gpu):
    """`vdx_comm_init / vdx_allgather_shard / vdx_halo_exchange` (include/vdx.h; RCCL resolved at run time) with a world
    of ONE rank — the round trip a single-GPU box allows: the gather reproduces the shard, the shard store gathers its
    units through the C-ABI on its side stream and the UNet output keeps its bits, an empty halo exchange is a no-op."""
    import vdx  # noqa: F401
    from vdx.comm import Comm
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict
    comm = Comm.create(Comm.unique_id(), 0, 1)
    side = torch.cuda.Stream(device=gpu)
    shard = torch.arange(4096, dtype=torch.float16, device=gpu)
    full = torch.zeros_like(shard)
    side.wait_stream(torch.cuda.current_stream())
    comm.allgather(shard, full, side)
    comm.halo(None, -1, None, -1, side)
    side.synchronize()
    assert torch.equal(full, shard)
    cfg = UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)
    m = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, seed=5, device=gpu), device=gpu)
    g = torch.Generator().manual_seed(2)
    x = torch.randn(2, 4, 3, 16, 16, generator=g).half().to(gpu)
    e = torch.randn(2, 77, 128, generator=g).half().to(gpu)
    want = m(x, 501, encoder_hidden_states=e).sample
    m.shard_(0, 1, comm=comm)
    got = m(x, 501, encoder_hidden_states=e).sample
    assert m.W.gathers > 10 and torch.equal(got, want)
    torch.cuda.synchronize()
    comm.destroy()


@pytest.mark.gpu
def test_side_stream_runs_beside_the_compute_stream(gpu):
    """HIP maps streams onto a few hardware queues and two streams on one queue take turns; every 4th normal-priority stream
    torch hands out shares the default (compute) stream's queue — with a NCCL process group alive, the first one (round 6,
    profiles/r06_rccl_contention.md).  The store's side stream is a HIGH-priority stream, which has a queue of its own:
    a 2 ms occupancy hog of one workgroup on it and 2 ms of GEMMs on the compute stream must take ~2 ms together, not 4 —
    whichever of torch's pool streams the process has handed out before."""
    import time
    import vdx  # noqa: F401
    from vdx import ops
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict
    burn = [torch.cuda.Stream(device=gpu) for _ in range(5)]          # move torch's round-robin pool on: the store must not care
    cfg = UNet3DConfig(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)
    m = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, seed=9, device=gpu), device=gpu).shard_(0, 1)
    side = m.W._side
    assert side.priority < 0, "the shard store's side stream must be a high-priority stream"
    g = torch.Generator(device=gpu).manual_seed(0)
    a = torch.randn(27648, 1280, device=gpu, dtype=torch.float16, generator=g)
    w = torch.randn(1280, 1280, device=gpu, dtype=torch.float16, generator=g) * 0.03

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) * 1e3

    def compute():
        for _ in range(20):
            ops.gemm(a, w, M=27648)
    compute()
    tc = min(timed(compute) for _ in range(3))

    def both(stream):
        def run():
            ops.occupancy_hog(1, 0, int(tc * 1e3), stream)
            compute()
            stream.synchronize()
        return min(timed(run) for _ in range(3))
    t_side = both(side)
    print(f"compute {tc:.2f} ms; with a {tc:.2f} ms hog on the store's side stream {t_side:.2f} ms")
    assert t_side < 1.5 * tc, f"the side stream takes turns with the compute stream ({t_side:.2f} ms against {tc:.2f})"
    del burn
