"""Worker of tests/test_dist_gpu.py (not a test): one rank of a real multi-process run of the chunked denoising flow on
the HIP kernels.  All ranks compute on cuda:0 (a GPU box has one card) and talk over gloo; everything else is the
product path: tiny-width UNet with its parameters sharded 1/world per rank and gathered per unit, shared noise, ctx
broadcast, CFG + DDIM per window, all-gather + full blend and halo exchange + owned-frame blend.

    torchrun --nproc-per-node W tests/dist_pipeline_worker.py OUT.pt MODE T CHUNK OVERLAP STEPS"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import vdx  # noqa: E402,F401
from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser  # noqa: E402
from vdx.scheduler import DDIMScheduler  # noqa: E402
from vdx.unet3d import UNet3DConditionModel, UNet3DConfig  # noqa: E402
from vdx.weights import synthetic_state_dict  # noqa: E402

TINY = dict(block_out_channels=(64, 128, 128, 128), cross_attention_dim=128, transformer_in_heads=2)


def build(dev, rank, world):
    cfg = UNet3DConfig(**TINY)
    m = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
    if world > 1:
        m.shard_(rank, world)
    emb = torch.randn(2, 77, TINY["cross_attention_dim"], generator=torch.Generator().manual_seed(5)).half().to(dev)
    return m, emb


def main():
    out, mode, T, chunk, ov, steps = sys.argv[1], sys.argv[2], *(int(a) for a in sys.argv[3:7])
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    m, emb = build(dev, rank, world)
    want_transport = os.environ.get("VDX_SHARD_TRANSPORT", "collective")
    if world > 1:
        assert m.W.transport == want_transport, (m.W.transport, want_transport)
    cfg = DiffuserConfig(num_frames=T, steps=steps, chunk_size=chunk, overlap=ov, height=256, width=256, mode=mode,
                         device="cuda", noise_device="cpu")
    d = DistributedVideoDiffuser(cfg, m, DDIMScheduler(), emb[1:], emb[:1])
    full, info = d(exchange="allgather")
    owned, info2 = d(exchange="halo")
    for s, e, lat in owned:
        assert torch.equal(lat, full[:, :, s:e]), ("halo != all-gather blend", rank, s, e)
    counts = torch.zeros(T, dtype=torch.int64)
    for s, e, _ in owned:
        counts[s:e] += 1
    dist.all_reduce(counts)
    assert bool((counts == 1).all()), counts
    second = None
    if world > 1 and want_transport == "peer":
        # a second store in the same process re-uses the peers' mappings (HIP refuses to open an open handle twice), and
        # close() lets go of them: the last reference closes the mapping
        from vdx import shard as shard_mod
        assert m.W.peer_self_check == "passed"
        m2, _ = build(dev, rank, world)
        second = m2.W.transport
        assert second == "peer", second
        refs = lambda: sum(ent[1] for ent in shard_mod._IPC_OPEN.values())    # noqa: E731
        assert refs() == 2 * (world - 1), shard_mod._IPC_OPEN      # (one mapping per peer segment, shared by both stores if the arenas share it)
        dist.barrier()
        m.W.close()
        assert m.W.transport == "collective" and refs() == world - 1
        m2.W.close()
        m2.W.close()                                              # idempotent
        assert len(shard_mod._IPC_OPEN) == 0
        dist.barrier()
    if rank == 0:
        torch.save({"lat": full.cpu(), "ranges": [tuple(r) for r in info["ranges"]], "overlap": info["overlap"],
                    "gathers": getattr(m.W, "gathers", None), "halo_bytes": info2["network_bytes"],
                    "transport": want_transport if world > 1 else getattr(m.W, "transport", None), "second_store": second}, out)
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok", flush=True)


if __name__ == "__main__":
    main()
