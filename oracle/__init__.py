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

PARITY STATUS.
PINNED to the reference itself (round 5): everything the reference OWNS on the path.
`tests/golden/make_ref_fixtures.py` executed `Distribution/strategies/{fsdp_chunked_coherent,
fsdp_chunked,chunk_only,fsdp}.py` UNMODIFIED (runpy, worlds 1, 2, 3, 4, 8 as real processes over
gloo, stand-ins only for what the container lacks) and recorded what they compute;
`tests/test_ref_exec_host.py` requires `pipeline_ref.py` to reproduce it with `==` / `torch.equal`:
1 575 planner configurations (80 of which the reference loops on forever: both raise), the seeded
noise and its per-chunk slices, the broadcast global context, one UNet call per (chunk, step) with
the reference's timestep sequence, every denoised chunk of eight "exact"-UNet jobs, the gathered
lists in blend order, every blended frame as it reached `vae.decode`, the CSV header and row.
Also used: the two notebook known answers the reference holds (the chunk split,
`Distribution/legacy/Latent Chunking/latent_chunking.ipynb:173-176`; the shared-vs-independent
overlap-noise statistics 0.0000 / 1.9990, `.../shared_overlap_noise/chunking_benchmark copy.ipynb:589-619`).

UNPINNED at the diffusers boundary: `diffusers` is not installed in the build container, is not
vendored under /root/reference, and the reference holds no test, fixture or golden vector for any
UNet / scheduler / VAE result.  `unet3d_ref.py`, `ddim_ref.py` and `vae_ref.py` restate
the published semantics; what pins them is structural — total parameter count 1 411 233 860,
per-block totals and the diffusers state-dict key/shape table (tests/test_oracle_host.py,
tests/test_host.py); the scheduler constants are recalled.
"""
