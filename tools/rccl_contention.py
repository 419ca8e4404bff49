#!/usr/bin/env python3
"""One-GPU pricing of RCCL-channel contention against the exact-fit persistent grids.

On a node the parameter gathers of the shard store run as RCCL all-gathers on a side stream; each RCCL channel is a
workgroup that holds a compute unit while the collective runs.  The weights-stationary GEMMs and K5 / K7 / K8 launch ONE
workgroup per CU with up to 160 KB of LDS: a launch that overlaps a gather finds R CUs taken and runs its R displaced
workgroups as a second round (2x for that launch).  A world-1 rehearsal cannot see this (its "all-gather" is a local copy).

Here: the cfg5 rank-0 window (16 frames, ctx injection, sharded store, world-1 collective path) runs with an OCCUPANCY HOG
on the side stream behind every gather — R workgroups x 256 threads x 64 KB LDS that touch no memory, for the time a ring
all-gather of the group's remote bytes (7/8 at world 8) takes at the modelled rate — for R in {0, 4, 8, 16, 32}, with the
persistent grids at full size and with `vdx_set_reserved_cus(r)`.  All configurations in ONE process on one set of weights,
interleaved round-robin (boxes and thermal state differ more than the effect).  Prints a markdown table + one JSON line.

    python tools/rccl_contention.py [--steps 4] [--rounds 3] [--gbs 100]"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--gbs", type=float, default=100.0, help="modelled all-gather rate per GPU (GB/s) at >= 8 channels")
    ap.add_argument("--as-world", type=int, default=8)
    ap.add_argument("--only", default=None, help="run only the configurations whose label contains this text (for a rocprofv3 kernel trace)")
    args = ap.parse_args()
    os.environ["VDX_SHARD_FORCE_COLLECTIVE"] = "1"
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29544")
    os.environ.setdefault("NCCL_DEBUG", "WARN")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    from vdx.shard import configure_rccl_env
    configure_rccl_env()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    import vdx  # noqa: F401
    from vdx import ops
    from vdx.pipeline import seeded_noise
    from vdx.planner import plan
    from vdx.scheduler import DDIMScheduler
    from vdx.unet3d import UNet3DConditionModel, UNet3DConfig
    from vdx.weights import synthetic_state_dict

    cfg = UNet3DConfig.zeroscope()
    unet = UNet3DConditionModel(cfg).load_diffusers_state_dict(synthetic_state_dict(cfg, 1234, dev), device=dev)
    unet.shard_(0, 1)
    store = unet.W
    sched = DDIMScheduler()
    sched.set_timesteps(50, device=dev)
    cp = plan(96, args.as_world, chunk_size=0, overlap=4)
    s0, e0 = cp.for_rank(0)[0]
    base = seeded_noise((1, 4, 96, 72, 128), sched.init_noise_sigma, dev)
    ctx = base.mean(dim=2, keepdim=True).contiguous()
    lat0 = base[:, :, s0:e0].clone()
    torch.manual_seed(1)
    emb = torch.randn(2, 77, 1024, device=dev, dtype=torch.float16)
    ts = sched._host_timesteps

    def step(i, lat):
        t = ts[i % len(ts)]
        noise = unet(ops.cfg_input(lat, ctx, 0.35), t, encoder_hidden_states=emb).sample
        return sched.step_cfg(noise, t, lat, 7.5)

    # (label, CUs held by the hog, modelled GB/s, reserved CUs).  Rate model: `--gbs` for 8+ channels; half of it for 4.
    g = args.gbs
    configs = [("no collective (world-1 rehearsal as before)", 0, g, 0),
               ("no collective, grids reserve 8", 0, g, 8),
               ("no collective, grids reserve 16", 0, g, 16),
               ("R=1 held (the gather's DURATION alone)", 1, g, 0), ("R=1 held, reserve 8", 1, g, 8),
               ("R=4 held", 4, g / 2, 0), ("R=4 held, reserve 8", 4, g / 2, 8),
               ("R=8 held", 8, g, 0), ("R=8 held, reserve 8", 8, g, 8),
               ("R=16 held", 16, g, 0), ("R=16 held, reserve 16", 16, g, 16),
               ("R=32 held", 32, g, 0), ("R=32 held, reserve 32", 32, g, 32),
               ("R=16 held, reserve 8", 16, g, 8), ("R=32 held, reserve 16", 32, g, 16), ("R=8 held, reserve 16", 8, g, 16)]
    if args.only:
        configs = [c for c in configs if c[1] == 0 and c[3] == 0] + [c for c in configs if args.only in c[0]]
    times = {c[0]: [] for c in configs}
    lat = lat0
    for i in range(2):
        lat = step(i, lat)
    torch.cuda.synchronize()
    for rnd in range(args.rounds):
        for label, R, gbs, res in configs:
            store.rehearse_hog = (R, 64 << 10, gbs, args.as_world) if R else None
            ops.set_reserved_cus(res)
            lat = step(0, lat0)                      # one untimed step in the new configuration
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(args.steps):
                lat = step(i, lat)
            torch.cuda.synchronize()
            times[label].append((time.perf_counter() - t0) / args.steps * 1e3)
    ops.set_reserved_cus(0)
    store.rehearse_hog = None
    es = 2
    # what the store's schedule predicts: group k computes for c_k while group k+1 is gathered for g_(k+1) (prefetch depth 1):
    # step = sum_k max(c_k, g_(k+1)).  c_k from events at the entry of every group on the compute stream, no hog.
    store.trace = []
    lat = step(0, lat0)
    end = torch.cuda.Event(enable_timing=True)
    end.record()
    torch.cuda.synchronize()
    tr, store.trace = store.trace, None
    names = [u for u, _ in tr]
    c = [tr[i][1].elapsed_time(tr[i + 1][1] if i + 1 < len(tr) else end) for i in range(len(tr))]
    gb = [store._padded[u] * es * (args.as_world - 1) / args.as_world / 1e9 for u in names]
    print("| group (first unit) | compute ms | remote MB gathered for it | gather ms at %g GB/s |" % g)
    print("|---|---|---|---|")
    for u, ck, b in zip(names, c, gb):
        print(f"| {u} | {ck:.3f} | {b * 1e3:.1f} | {b / g * 1e3:.3f} |")
    pred = sum(max(c[k], gb[(k + 1) % len(c)] / g * 1e3) for k in range(len(c)))
    print(f"\nsum of compute {sum(c):.2f} ms; sum of gathers {sum(gb) / g * 1e3:.2f} ms; predicted step with prefetch depth 1 = sum_k max(c_k, g_(k+1)) = {pred:.2f} ms\n")
    remote_gb = sum(store._padded[u] for u in store.schedule) * es * (args.as_world - 1) / args.as_world / 1e9
    base_ms = min(times[configs[0][0]])
    print(f"cfg5 rank-0 window ({e0 - s0} frames), sharded store, {len(store.schedule)} gathers per step, {remote_gb:.2f} GB of remote "
          f"parameters per step at world {args.as_world}; best of {args.rounds} rounds x {args.steps} steps, one process\n")
    print("| configuration | CUs held beside a gather | modelled rate GB/s | hog time per step ms | persistent grids | ms per step (best) | vs. no collective |")
    print("|---|---|---|---|---|---|---|")
    rows = []
    for label, R, gbs, res in configs:
        best = min(times[label])
        hog_ms = remote_gb / gbs * 1e3 if R else 0.0
        grid = ops.set_reserved_cus(res)
        print(f"| {label} | {R} | {gbs:g} | {hog_ms:.1f} | {grid} CUs | {best:.2f} | {100 * (best / base_ms - 1):+.1f} % |")
        rows.append({"label": label, "cus_held": R, "gbs": gbs, "reserved": res, "grid_cus": grid, "ms": round(best, 3), "all_ms": [round(t, 3) for t in times[label]]})
    ops.set_reserved_cus(0)
    print()
    print(json.dumps({"tool": "rccl_contention", "window_frames": e0 - s0, "remote_gb_per_step": round(remote_gb, 3), "rows": rows}))
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
