"""The reference's result row (fsdp_chunked_coherent.py:227-276, 313-333): boundary-quality metric, peak-memory
reduction and the CSV contract that `Distribution/plot_helpers/plot.py:7-13` reads.

`temp_instab` (mean absolute difference of the two frames either side of every chunk boundary) is restated
exactly (:227-247, host numpy on the decoded uint8 frames, as the reference computes it).  `flow_err` (:236-245) is
the same loop with OpenCV's Farneback flow + remap; `cv2` is taken from the environment when it is installed and
from `vdx.compat.cv2_shim` (an own implementation of the published algorithm) otherwise — with the shim the number
is not pinned against OpenCV's ("parity unpinned", see the shim's docstring).  `write_video` is :250-253.
"""
from __future__ import annotations

import csv
import datetime
import os
import socket
from typing import Dict, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

CSV_HEADER = ["timestamp", "host", "mode", "world_size", "num_frames", "chunk_size", "overlap", "latency_s",
              "throughput_fps", "peak_vram_mb", "end_vram_mb", "network_bytes", "net_gather_s", "net_reduce_s",
              "temp_instab", "flow_err"]                                     # :323-329 == plot.py:7-13


def boundary_l1(frames: Sequence[np.ndarray], ranges: Sequence[Tuple[int, int]]) -> Optional[float]:
    """:229-247 — for every chunk end e (all but the last chunk in start order) with 0 < e < len(frames):
    mean |frames[e] - frames[e-1]| in float32; the mean of those, or None when there is no boundary."""
    if len(frames) <= 1:
        return None
    ends = [e for (_s, e) in sorted(ranges, key=lambda r: r[0])[:-1]]
    diffs = [np.mean(np.abs(frames[e].astype(np.float32) - frames[e - 1].astype(np.float32)))
             for e in ends if 0 < e < len(frames)]
    return float(np.mean(diffs)) if diffs else None


def _cv2():
    try:
        import cv2
        return cv2
    except ImportError:
        from .compat import cv2_shim
        return cv2_shim


def flow_warp_error(frames: Sequence[np.ndarray], ranges: Sequence[Tuple[int, int]]) -> Optional[float]:
    """:229-246 — at every chunk boundary: Farneback flow prev -> next (pyr 0.5, 3 levels, window 15, 3 iterations,
    poly 5 / 1.2), warp prev by it (remap, bilinear) and take the mean absolute difference to next."""
    cv2 = _cv2()
    if len(frames) <= 1:
        return None
    ends = [e for (_s, e) in sorted(ranges, key=lambda r: r[0])[:-1]]
    diffs = []
    for e in ends:
        if not 0 < e < len(frames):
            continue
        f_prev, f_next = frames[e - 1], frames[e]
        prev_gray, next_gray = cv2.cvtColor(f_prev, cv2.COLOR_BGR2GRAY), cv2.cvtColor(f_next, cv2.COLOR_BGR2GRAY)
        flow = cv2.calcOpticalFlowFarneback(prev_gray, next_gray, None, 0.5, 3, 15, 3, 5, 1.2, 0)
        h, w = prev_gray.shape
        flow_x = (np.arange(w)[None, :] + flow[:, :, 0]).astype(np.float32)
        flow_y = (np.arange(h)[:, None] + flow[:, :, 1]).astype(np.float32)
        warp_prev = cv2.remap(f_prev, flow_x, flow_y, cv2.INTER_LINEAR)
        diffs.append(np.mean(np.abs(warp_prev.astype(np.float32) - f_next.astype(np.float32))))
    return float(np.mean(diffs)) if diffs else None


def write_video(frames: Sequence[np.ndarray], path: str, fps: float) -> None:
    """:250-253 — `cv2.VideoWriter(path, fourcc("mp4v"), fps, (W, H))`, frames RGB -> BGR, release."""
    cv2 = _cv2()
    h, w = frames[0].shape[:2]
    vw = cv2.VideoWriter(path, cv2.VideoWriter_fourcc(*"mp4v"), fps, (w, h))
    for f in frames:
        vw.write(cv2.cvtColor(f, cv2.COLOR_RGB2BGR))
    vw.release()


def peak_vram_mb(device) -> Tuple[int, float]:
    """:255-261 — this rank's torch peak (MiB) MAX-reduced over ranks, and the seconds the reduction took."""
    import time
    peak = torch.tensor(torch.cuda.max_memory_allocated(device) // 1024 ** 2, device=device)
    t0 = time.time()
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(peak, op=dist.ReduceOp.MAX)
    return int(peak.item()), time.time() - t0


def result_row(res: Dict, *, mode: str, num_frames: int, elapsed_s: float) -> Dict:
    """:313-321 — the rank-0 additions to the dict `DistributedVideoDiffuser.__call__` returns."""
    row = dict(res)
    row.update({"timestamp": datetime.datetime.utcnow().isoformat(timespec="seconds"), "host": socket.gethostname(),
                "mode": mode, "latency_s": elapsed_s, "throughput_fps": round(num_frames / elapsed_s, 3)})
    return row


def append_csv(path: str, row: Dict) -> None:
    """:322-332 — append one row under the fixed header (written only when the file is new); missing -> ''. """
    write_hdr = not os.path.exists(path)
    with open(path, "a", newline="") as f:
        w = csv.DictWriter(f, fieldnames=CSV_HEADER)
        if write_hdr:
            w.writeheader()
        w.writerow({k: ("" if row.get(k) is None else row.get(k, "")) for k in CSV_HEADER})
