"""CPU suite: the reference's result row — boundary metric and CSV contract (fsdp_chunked_coherent.py:227-247,
313-333; header = Distribution/plot_helpers/plot.py:7-13)."""
import csv

import numpy as np

import vdx  # noqa: F401
from vdx import metrics


def test_csv_header_is_the_plot_helpers_contract():
    assert metrics.CSV_HEADER == ["timestamp", "host", "mode", "world_size", "num_frames", "chunk_size", "overlap",
                                  "latency_s", "throughput_fps", "peak_vram_mb", "end_vram_mb", "network_bytes",
                                  "net_gather_s", "net_reduce_s", "temp_instab", "flow_err"]


def test_boundary_l1_hand_computed():
    # 6 frames of 2x2x3, frame i filled with 10*i except frame 4 = 55; chunks (0,3),(2,5),(4,6) -> boundary ends 3, 5
    frames = [np.full((2, 2, 3), 10 * i, np.uint8) for i in range(6)]
    frames[4][:] = 55
    got = metrics.boundary_l1(frames, [(2, 5), (0, 3), (4, 6)])        # unsorted on purpose: :229 sorts by start
    # e=3: |30-20| = 10 ; e=5: |50-55| = 5  -> mean 7.5 ; the last chunk's end is not a boundary
    assert got == 7.5
    assert metrics.boundary_l1(frames, [(0, 6)]) is None               # one chunk: no boundary
    assert metrics.boundary_l1(frames[:1], [(0, 1)]) is None
    assert metrics.boundary_l1(frames, [(0, 6), (0, 6)]) is None       # end == len(frames) is skipped (:234)
    # uint8 wrap-around must not happen (the reference casts to float32 first)
    a, b = np.zeros((1, 1, 3), np.uint8), np.full((1, 1, 3), 255, np.uint8)
    assert metrics.boundary_l1([b, a], [(0, 1), (1, 2)]) == 255.0


def test_csv_rows_append_under_one_header(tmp_path):
    path = str(tmp_path / "results.csv")
    res = {"world_size": 2, "chunk_size": 16, "overlap": 4, "num_frames": 24, "peak_vram_mb": 8123, "end_vram_mb": 900,
           "network_bytes": 2359296, "net_gather_s": 0.01, "net_reduce_s": 0.001, "temp_instab": 3.25, "flow_err": None}
    row = metrics.result_row(res, mode="hybrid_ctx", num_frames=24, elapsed_s=12.0)
    assert row["throughput_fps"] == 2.0 and row["mode"] == "hybrid_ctx"
    metrics.append_csv(path, row)
    metrics.append_csv(path, row)
    rows = list(csv.reader(open(path)))
    assert rows[0] == metrics.CSV_HEADER and len(rows) == 3
    rec = dict(zip(rows[0], rows[1]))
    assert rec["flow_err"] == "" and rec["temp_instab"] == "3.25" and rec["world_size"] == "2"
