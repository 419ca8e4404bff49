"""-m gpu: the whole flow with REAL processes on the HIP kernels.  `world` processes (one rank each, all computing on this
box's one GPU, collectives over gloo — RCCL refuses two ranks on one device) run plan -> shared noise -> ctx broadcast ->
CFG/DDIM per window on a UNet whose parameters are sharded 1/world per rank -> all-gather + blend, and halo exchange +
owned-frame blend.  The result must carry the bits of the SAME job computed serially in this process (every window
denoised with resident weights, blended in the reference's rank-major order, fsdp_chunked_coherent.py:184-217)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,mode,T,chunk,ov,transport", [(2, "hybrid_ctx", 20, 0, 4, "peer"), (3, "hybrid", 17, 6, 2, "peer"),
                                                             (2, "fsdp", 6, 0, 4, "peer"), (2, "hybrid_ctx", 20, 0, 4, "collective")])
def test_multi_process_job_equals_serial_job_bitwise(gpu, tmp_path, world, mode, T, chunk, ov, transport):
    """transport "peer": every rank maps the other ranks' shard arenas (HIP IPC between the processes) and gathers a unit
    as device-to-device copies on its side stream — no collective, no gather kernel (vdx/shard.py, vdx_peer_gather);
    "collective": torch.distributed all-gather (gloo here)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import vdx  # noqa: F401
    from dist_pipeline_worker import build
    from vdx.pipeline import DiffuserConfig, DistributedVideoDiffuser, seeded_noise
    from vdx.planner import plan
    from vdx.scheduler import DDIMScheduler
    steps = 2
    out = tmp_path / "rank0.pt"
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(29660 + world + T),
                        os.path.join(ROOT, "tests", "dist_pipeline_worker.py"), str(out), mode, str(T), str(chunk), str(ov),
                        str(steps)], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, VDX_SHARD_TRANSPORT=transport))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert r.stdout.count("ok") == world
    got = torch.load(out, weights_only=True)

    # the same N-rank job, serially: one process, resident weights, every window in turn
    m, emb = build(gpu, 0, 1)
    cfg = DiffuserConfig(num_frames=T, steps=steps, chunk_size=chunk, overlap=ov, height=256, width=256, mode=mode,
                         device="cuda", noise_device="cpu")
    d = DistributedVideoDiffuser(cfg, m, DDIMScheduler(), emb[1:], emb[:1])
    cp = plan(T, world, chunk, ov, no_chunking=mode == "fsdp")
    assert [tuple(x) for x in got["ranges"]] == [tuple(x) for x in cp.ranges] and got["overlap"] == cp.overlap
    base = seeded_noise((1, 4, T, 32, 32), d.scheduler.init_noise_sigma, "cuda", "cpu")
    order = [i for rk in range(world) for i in range(len(cp.ranges)) if i % world == rk]
    den = {i: d.denoise(base[:, :, cp.ranges[i][0]:cp.ranges[i][1]].clone()) for i in order}
    want = d.blend([(cp.ranges[i][0], cp.ranges[i][1], den[i]) for i in order], base, cp.overlap)
    assert torch.equal(got["lat"], want.cpu())
    if world > 1 and got["gathers"] is not None:
        assert got["gathers"] > 0                     # the parameters really went through the per-unit gather
        assert got["transport"] == transport
