"""ORACLE — TEST INFRASTRUCTURE ONLY.

CPU restatement (plain PyTorch fp32 / fp16-on-CPU) of the reference's hybrid
FSDP + frame-chunked denoising path
(`Distribution/strategies/fsdp_chunked_coherent.py`) and of the diffusers
operators that path calls (`UNet3DConditionModel`, `DDIMScheduler`).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg
may import anything from this package, and only as the checker / the reported
CPU baseline.  Nothing under the product package imports it; the product path
raises when the HIP library is missing.

(`vae_ref.py` restates `AutoencoderKL.decode` the same way.  The CLIP text tower needs no restatement: its
dependency, `transformers`, is installed, and the tests compare against `transformers.CLIPTextModel` directly.)

PARITY UNPINNED at the diffusers boundary: `diffusers` is not installed in the
build container, is not vendored under /root/reference, and the reference
holds no test, fixture or golden vector for any UNet / scheduler / blended
latent result.  What *is* pinned:
  * planner / blend / noise semantics — hand-executed known-answer tables from
    the reference source (tests/test_oracle_host.py, tests/test_host.py,
    tests/test_halo_host.py) plus the two committed notebook known answers:
    the chunk split (`Distribution/legacy/Latent Chunking/latent_chunking.ipynb:173-176`)
    and the shared-vs-independent overlap-noise statistics 0.0000 / 1.9990
    (`.../shared_overlap_noise/chunking_benchmark copy.ipynb:589-619`,
    restated in tests/test_oracle_host.py);
  * the UNet structure — total parameter count 1 411 233 860, per-block totals
    and the diffusers state-dict key/shape table (tests/test_oracle_host.py,
    tests/test_host.py).
"""
