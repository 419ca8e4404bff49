"""-m gpu: owned-frame blend on the HIP kernels.  One process plays every rank of an N-rank job (the pieces a
rank would receive over RCCL are sliced from the sender's chunk on the same device); the owned frames must carry the
bits of (a) the oracle's `ramp_blend` of all chunks (`fsdp_chunked_coherent.py:204-217`) and (b) the product's own
full blend.  The transfers themselves are exercised with real processes in tests/test_halo_host.py (gloo)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("T,W,cs,ov,nc", [(96, 8, 0, 4, False), (48, 4, 0, 4, False), (24, 2, 0, 4, False),
                                          (32, 3, 0, 4, False), (31, 4, 10, 7, False), (24, 2, 0, 0, True),
                                          (32, 1, 0, 4, False)])
def test_owned_blend_bits_on_hip_kernels(gpu, T, W, cs, ov, nc):
    _owned_blend_case(gpu, T, W, cs, ov, nc, 4, 9, 16)


def test_owned_blend_bits_at_cfg5_full_size(gpu):
    """BASELINE config 5 at its real extent: 96 frames of a (4, 72, 128) latent, 8 ranks, 16-frame windows."""
    _owned_blend_case(gpu, 96, 8, 0, 4, False, 4, 72, 128)


def _owned_blend_case(gpu, T, W, cs, ov, nc, C, H, Wd):
    import vdx  # noqa: F401
    from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser, HaloPlan, blend_owned
    from vdx.planner import plan
    from oracle.pipeline_ref import ramp_blend
    cp = plan(T, W, cs, ov, no_chunking=nc)
    g = torch.Generator().manual_seed(T + W)
    chunks = {i: (torch.randn(1, C, e - s, H, Wd, generator=g) * 2).half() for i, (s, e) in enumerate(cp.ranges)}
    like = torch.zeros(1, C, T, H, Wd, dtype=torch.float16)
    order = [i for r in range(W) for i in range(len(cp.ranges)) if i % W == r]
    want = ramp_blend([(cp.ranges[i][0], cp.ranges[i][1], chunks[i]) for i in order], T, cp.overlap, like)
    dev = {i: t.to(gpu) for i, t in chunks.items()}
    # (b) the product's full blend in the reference's order
    d = object.__new__(DistributedVideoDiffuser)
    d.cfg, d.rank, d.world = DiffuserConfig(device="cuda"), 0, 1
    full = d.blend([(cp.ranges[i][0], cp.ranges[i][1], dev[i]) for i in order], like.to(gpu), cp.overlap)
    assert torch.equal(full.cpu(), want)
    hp = HaloPlan(cp, T)
    got_full = torch.full_like(want, float("nan"))
    for rank in range(W):
        mine = [dev[i] for i in range(len(cp.ranges)) if i % W == rank]
        recv = {(t.chunk, t.s, t.e): dev[t.chunk][:, :, t.s - cp.ranges[t.chunk][0]:t.e - cp.ranges[t.chunk][0]].contiguous()
                for t in hp.transfers if t.dst == rank}
        for s, e, lat in blend_owned(mine, hp, recv, None, like.to(gpu), rank):
            got_full[:, :, s:e] = lat.cpu()
    assert torch.equal(got_full, want)
