"""Frame-chunk planner and chunk->rank assignment (host logic, integers only).

Behaviour follows `Distribution/strategies/fsdp_chunked_coherent.py:149-177,184` and the overlap
rule of its siblings (`fsdp_chunked.py:143`, `chunk_only.py:86`).  Where the reference's window
walk would never terminate (overlap >= chunk) this raises instead.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple

Range = Tuple[int, int]


class PlannerError(ValueError):
    pass


@dataclass(frozen=True)
class ChunkPlan:
    chunk: int
    overlap: int
    ranges: Tuple[Range, ...]
    world: int

    def for_rank(self, rank: int) -> List[Range]:
        """Round-robin: window i belongs to rank i % world (reference :184)."""
        return [r for i, r in enumerate(self.ranges) if i % self.world == rank]

    @property
    def per_rank(self) -> int:
        return len(self.ranges) // self.world


def _windows(total: int, size: int, overlap: int) -> List[Range]:
    step = size - overlap
    if step <= 0:
        raise PlannerError(
            f"overlap {overlap} >= chunk {size}: windows would never advance "
            "(the reference loops forever on this input)")
    return [(s, min(s + size, total)) for s in range(0, total, step)]


def auto_chunk(total: int, world: int) -> int:
    lo = max(4, total // (2 * world))
    hi = min(16, total // world)
    return min(hi, max(lo, total // world))


def plan(total: int, world: int, chunk_size: int = 0, overlap: int = 4, no_chunking: bool = False,
         overlap_rule: str = "coherent") -> ChunkPlan:
    if total <= 0 or world <= 0:
        raise PlannerError("num_frames and world size must be positive")
    if no_chunking:
        size, ov = total, 0
    else:
        size = chunk_size if chunk_size > 0 else auto_chunk(total, world)
        if overlap_rule == "coherent":
            ov = overlap if overlap > 0 else max(4, size // 3)
        elif overlap_rule == "third":
            ov = min(overlap, size // 3)
        else:
            raise PlannerError(f"unknown overlap rule {overlap_rule!r}")
    wins = _windows(total, size, ov)
    if len(wins) % world:
        # grow the chunk until the window count divides evenly, if any size below 2x does
        for grown in range(size + 1, 2 * size):
            cand = _windows(total, grown, ov)
            if len(cand) % world == 0:
                size, wins = grown, cand
                break
    if len(wins) % world:
        wins = wins + [wins[-1]] * (world - len(wins) % world)      # repeat the tail window
    return ChunkPlan(size, ov, tuple(wins), world)
