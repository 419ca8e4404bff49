"""Per-unit parameter sharding with prefetching all-gather — the MI355X-native counterpart of the
reference's `FSDP(pipe.unet, FULL_SHARD, ...)` wrap (`fsdp_chunked_coherent.py:63-88`).

Every *unit* (one resnet, one temporal-conv stack, one spatial or temporal transformer, one
resampler) owns a flat fp16 buffer; each GPU keeps 1/world of it.  During a forward the units are
used strictly in order (`UNet3DConditionModel.unit_schedule`), so while unit k computes, units k+1
and k+2 are gathered on a side HIP stream (prefetch depth 2, three gather buffers; round 6); a buffer is
only overwritten after the compute stream has passed the unit that used it.  Inference only: no reduce-scatter.
Sharding changes memory, never results.

Transports of the gather (`transport=`, env VDX_SHARD_TRANSPORT):
  "collective"  (DEFAULT, round 5) `torch.distributed.all_gather_into_tensor` on the side stream — backend "nccl" = RCCL
                over xGMI, what the north star names — or `all_gather` over gloo (CPU tensors, tests);
  "peer"        (opt-in) the ranks' shard arenas are mapped into each other once (HIP IPC, `vdx_ipc_*`); a gather is
                `world` device-to-device copies on the side stream (`vdx_peer_gather`): copy engines over xGMI, no compute
                unit taken from the GEMMs that run meanwhile, no collective — parameters are read-only after load, so a
                rank pulls what it needs when it needs it.  It was the default through round 4.  Round 5
                ran it for the first time between two processes at FULL size (ONE 1.3 GB arena per rank, both ranks on the
                one GPU of a box): a failed self-check and a `hipIpcOpenMemHandle` that never returned.  Round 6 found
                the cause with a staged probe under host-side deadlines (tools/peer_check.py, profiles/r06_peer_transport.md):
                a handle exports the whole allocation around the pointer — here the caching allocator's segment — and a
                mapping of an allocation of 2 GiB or more does not open on this driver (segments up to 2000 MB: open, read
                back bit for bit, serialised or concurrent opens alike; 2048 MB and 2600 MB: never return).  Round 5's arena
                had been carved out of a cached 2.6 GB block.  Nothing to do with concurrent opens or with the gloo gather
                the self-check compares against.  The arenas are now allocations of their own of at most 256 MB
                (`ARENA_BYTES`), `vdx_ipc_export` refuses an allocation of 1 GiB or more (the store then falls back to the
                collective on every rank), and the full-size two-process run passes its self-check.  The cross-DEVICE
                case still has never run (no multi-GPU node in any round), so RCCL stays the default;
  comm=         the C-ABI RCCL entry point `vdx_allgather_shard` (vdx/comm.py).

The store is a read-only mapping (name -> tensor view) and is what `UNet3DConditionModel.W`
becomes after `shard_()`.  Works on CPU tensors with the `gloo` backend (used by the tests).
"""
from __future__ import annotations

import ctypes as C
import os
import time
import warnings
from typing import Callable, Dict, List, Optional

import torch
import torch.distributed as dist

ALIGN = 64          # elements: keeps every view 128-byte aligned

# RCCL's all-gather runs as `nchannels` workgroups, each holding a compute unit for the duration of the collective.  Round 6
# PRICED that on one GPU (tools/rccl_contention.py, tools/trace_hog_diff.py, profiles/r06_rccl_contention.md: an occupancy hog of
# R workgroups behind every gather for the time a ring all-gather of the group's remote bytes takes):
#   * holding 1, 8, 16 or 32 CUs costs the SAME, +6 % of a 16-frame window at 100 GB/s: ~2 ms of exposed gathers + ~4-5 ms of
#     SECOND ROUNDS in the tiled GEMM, whose planner covers a level-1/2 product with whole rounds of one tile per CU — ONE held CU
#     turns a round into two (the persistent exact-fit grids sit at levels 0-1, where the gathers are short);
#   * `vdx_set_reserved_cus(r)` — persistent grids AND those main launches fill only the unreserved CUs — takes it to +3.4 % when
#     r >= the CUs held and r <= 16, costs +0.5 % when nothing is held, and buys nothing when r is smaller than what is held;
#   * halving the gather RATE (4 channels at 50 GB/s) costs +17 %.
# The right r is RCCL's channel count on a node, which no run of this build has seen (32+ by default: no reserve helps, and the
# gathers are shorter than modelled).  Hence the DEFAULTS: no channel cap, reserve 0 — both environment knobs for the first node
# run (NCCL_MAX_NCHANNELS=16 VDX_RESERVED_CUS=16 is the pair to try) — and the two things that mattered unconditionally: the side
# stream on a hardware queue of its own (`_side` below: -18 ms) and prefetch depth 2 (`prefetch_depth`).


def configure_rccl_env(env=None):
    """Environment of a multi-GPU run, to be in place BEFORE `init_process_group` / the first collective (bench.py and its
    launcher call this).  torch.distributed runs collectives on the process group's OWN stream: take it from the
    high-priority pool, whose hardware queues the default (compute) stream never shares (ShardedStore.__init__ on `_side`)."""
    env = os.environ if env is None else env
    env.setdefault("TORCH_NCCL_HIGH_PRIORITY", "1")
    return env


def reserved_cus_for(world: int, transport: str) -> int:
    """CUs the persistent grids and the tiled GEMM's exact-fit main launches leave to a collective's channel kernels:
    VDX_RESERVED_CUS for a world > 1 on the collective transport, else 0 (default 0: see above)."""
    if world <= 1 or transport != "collective":
        return 0
    return int(os.environ.get("VDX_RESERVED_CUS", "0"))


ARENA_BYTES = 256 << 20      # largest shard arena (= largest allocation ever exported through HIP IPC); see ShardedStore.__init__


def _FORCE_COLLECTIVE():
    """Rehearsal switch: go through the gather transport even for a world of 1 (bench.py --rehearse-dist)."""
    return os.environ.get("VDX_SHARD_FORCE_COLLECTIVE") == "1" and dist.is_initialized()


def _round_up(x, m):
    return (x + m - 1) // m * m


# Peer allocations mapped into this process: (peer pid, IPC handle bytes) -> [base pointer, reference count].  HIP refuses
# to open a handle that is already open, and two stores of one process can export from the same allocator segment, so
# a mapping is opened once per process and closed when its last store lets go of it.
_IPC_OPEN: Dict = {}


def _ipc_acquire(lib, pid, handle, offset):
    ent = _IPC_OPEN.get((pid, handle))
    if ent is None:
        p = C.c_void_p()
        if lib.vdx_ipc_open(handle, 0, C.byref(p)) != 0:
            return None
        ent = _IPC_OPEN[(pid, handle)] = [p.value, 0]
    ent[1] += 1
    return ent[0] + offset


def _ipc_release(lib, pid, handle):
    ent = _IPC_OPEN.get((pid, handle))
    if ent is None:
        return
    ent[1] -= 1
    if ent[1] <= 0:
        del _IPC_OPEN[(pid, handle)]
        lib.vdx_ipc_close(ent[0], 0)


class ShardedStore:
    def __init__(self, tensors: Dict[str, torch.Tensor], unit_of: Callable[[str], Optional[str]],
                 schedule: List[str], rank: int, world: int, group=None, comm=None, transport: Optional[str] = None,
                 merge_bytes: int = 64 << 20, prefetch_depth: int = 2):
        """`unit_of(name)` -> unit id, or None for tensors kept replicated (small stem tensors).
        `comm`: a `vdx.comm.Comm` — the gathers then go through the C-ABI (`vdx_allgather_shard`, RCCL) instead of
        `torch.distributed`.  `transport`: "collective" | "peer" (module docstring); default from VDX_SHARD_TRANSPORT,
        else "collective".  `merge_bytes`: neighbours of the schedule are gathered TOGETHER while their sum stays within
        this many bytes (fewer, larger transfers: the level-0 / level-1 units of the XL UNet are 2-18 MB each and a gather
        is `world` copies or one collective whatever its size); the gather buffers are sized by the largest unit
        anyway (95 MB), so merging below that costs no memory.  0 = one gather per unit."""
        self.rank, self.world, self.group, self.comm = rank, world, group, comm
        if merge_bytes > 0:
            schedule, unit_of = _merge_units(tensors, unit_of, list(schedule), merge_bytes)
        self.schedule = list(schedule)
        self._pos = {u: i for i, u in enumerate(self.schedule)}
        self.replicated: Dict[str, torch.Tensor] = {}
        self._layout: Dict[str, List] = {u: [] for u in self.schedule}     # unit -> [(name, off, shape)]
        self._unit_of: Dict[str, str] = {}
        sizes = {u: 0 for u in self.schedule}
        any_t = next(iter(tensors.values()))
        self.device, self.dtype = any_t.device, any_t.dtype
        for name, t in tensors.items():
            u = unit_of(name)
            if u is None:
                self.replicated[name] = t
                continue
            if u not in sizes:
                raise KeyError(f"tensor {name!r} maps to unit {u!r} which is not in the schedule")
            self._layout[u].append((name, sizes[u], tuple(t.shape)))
            self._unit_of[name] = u
            sizes[u] += _round_up(t.numel(), ALIGN)
        self._padded = {u: _round_up(max(n, 1), ALIGN * world) for u, n in sizes.items()}
        # local shards: a few ARENAS (the units' 1/world slices back to back, the same layout on every rank), so that a
        # handful of IPC exports per rank make all of them reachable.  An arena is at most ARENA_BYTES and an allocation of
        # its own: a HIP IPC handle exports the whole allocation AROUND a pointer (the caching allocator's segment), and a
        # mapping of an allocation of 2 GiB or more does not open on this driver (`hipIpcOpenMemHandle` never returns;
        # measured round 6, profiles/r06_peer_transport.md: arenas in segments of up to 2000 MB open and read back, in
        # segments of 2048 / 2600 MB they hang — round 5's single 1.3 GB arena had been carved out of a cached 2.6 GB
        # block).  `vdx_ipc_export` refuses allocations of 1 GiB or more.  A unit never straddles two arenas.
        es = any_t.element_size()
        self._arena_off: Dict[str, tuple] = {}       # unit -> (arena index, element offset inside it)
        sizes_a: List[int] = [0]
        for u in self.schedule:
            n = self._padded[u] // world
            if sizes_a[-1] and (sizes_a[-1] + n) * es > ARENA_BYTES:
                sizes_a.append(0)
            self._arena_off[u] = (len(sizes_a) - 1, sizes_a[-1])
            sizes_a[-1] += n
        if self.device.type == "cuda":
            torch.cuda.empty_cache()     # an arena must be its OWN allocation (what gets exported is the allocation around it)
        self._arenas = [torch.zeros(max(n, 1), dtype=self.dtype, device=self.device) for n in sizes_a]
        self.shards: Dict[str, torch.Tensor] = {}
        for u in self.schedule:
            flat = torch.zeros(self._padded[u], dtype=self.dtype, device=self.device)
            for name, o, shape in self._layout[u]:
                flat[o:o + tensors[name].numel()] = tensors[name].reshape(-1)
            n = self._padded[u] // world
            ai, ao = self._arena_off[u]
            self.shards[u] = self._arenas[ai][ao:ao + n]
            self.shards[u].copy_(flat[rank * n:(rank + 1) * n])
            del flat
        cap = max(self._padded.values())
        # prefetch depth d: while group k computes, groups k+1 .. k+d are gathered (or already there); d + 1 gather buffers.
        # Depth 1 hides a gather only behind the ONE group before it, and at the deep levels (1280 channels: 60-95 MB of
        # weights per group, 0.4-0.6 ms of compute at 16 frames) a gather outlasts that group: sum_k max(c_k, g_(k+1)).
        # Depth 2 lets the gathers of the deep levels run on while the 1.5-ms groups around them compute
        # (profiles/r06_rccl_contention.md); one more buffer of the largest group (95 MB).
        self.prefetch_depth = max(1, int(prefetch_depth))
        nb = self.prefetch_depth + 1
        self._bufs = [torch.empty(cap, dtype=self.dtype, device=self.device) for _ in range(nb)]
        self._resident = [None] * nb           # unit held by each buffer
        self._ready = [None] * nb              # event: gather into buffer finished
        self._released = [None] * nb           # event: compute stream is past the unit in that buffer
        self._view_cache: Dict = {}            # (slot, unit) -> {name: view}: built once, not per switch
        self._views: Dict[str, torch.Tensor] = {}
        self._current: Optional[str] = None
        self._cuda = self.device.type == "cuda"
        # HIGH priority, and not for the priority's sake: HIP maps streams onto a few hardware queues (4 by default), two
        # streams on one queue run in submission order, and every 4th normal-priority stream torch hands out — with a NCCL
        # process group alive, the very FIRST — shares the queue of the default (compute) stream: the "side" stream then
        # takes turns with the step instead of running beside it (measured round 6, tools/overlap_streams.py,
        # profiles/r06_rccl_contention.md: a gather of modelled duration g made the step g longer whatever it held).
        # High-priority streams live on queues of their own.
        self._side = torch.cuda.Stream(device=self.device, priority=-1) if self._cuda else None
        self.gathers = 0
        self.transport = "collective"
        self.peer_self_check = "not run"       # "passed" | "failed" once a world > 1 store has compared peer vs collective
        self.rehearse_copies = 0               # > 1 (world of 1 only): issue a gather as that many copies (see _peer_gather)
        self.gather_host_s = 0.0               # host time spent ENQUEUEING gathers (perf_counter around the transport call)
        self._peer_ptrs = None
        self._opened: List = []                # (peer pid, handle) of every mapping this store holds a reference to
        want = transport or os.environ.get("VDX_SHARD_TRANSPORT") or "collective"
        if want not in ("peer", "collective"):
            raise ValueError(f"unknown shard transport {want!r}")
        if want == "peer" and self._cuda and comm is None and (world > 1 or _FORCE_COLLECTIVE() or transport == "peer"):
            self._setup_peer()
        self.trace = None                      # list to collect (group, event at entry on the compute stream): tools/rccl_contention.py
        self.rehearse_hog = None               # (blocks, lds bytes, modelled GB/s, as_world): one-GPU rehearsal of the CUs RCCL holds
        self.reserved_cus = 0
        if self._cuda:
            from . import ops
            self.reserved_cus = reserved_cus_for(world, self.transport)
            ops.set_reserved_cus(self.reserved_cus)

    # ---- peer transport -----------------------------------------------------------------------
    def _setup_peer(self):
        """Map every other rank's arena into this process (collective over the group: handles travel by
        all_gather_object).  Falls back to the collective transport, on every rank alike, if any rank fails."""
        from . import _lib
        lib = _lib.load()
        torch.cuda.synchronize(self.device)          # the arenas are complete before anybody may read them
        exports = []                                 # per arena: (handle bytes | None, byte offset inside its allocation)
        for a in self._arenas:
            handle, off = C.create_string_buffer(64), C.c_size_t(0)
            rc = lib.vdx_ipc_export(a.data_ptr(), handle, C.byref(off))      # refuses allocations of 1 GiB or more
            exports.append((handle.raw if rc == 0 else None, off.value))
        mine = (self.rank, os.getpid(), exports)
        if self.world > 1:
            infos = [None] * self.world
            dist.all_gather_object(infos, mine, group=self.group)
        else:
            infos = [mine]
        ptrs, opened = [None] * self.world, []       # ptrs[rank][arena] = address of that rank's arena in THIS process
        ok = all(h is not None for i in infos for h, _ in i[2])
        if ok:
            for r, pid, exps in infos:
                if r == self.rank:
                    ptrs[r] = [a.data_ptr() for a in self._arenas]
                    continue
                ptrs[r] = []
                for h, o in exps:
                    p = _ipc_acquire(lib, pid, h, o)
                    if p is None:
                        ok = False
                        break
                    ptrs[r].append(p)
                    opened.append((pid, h))
                if not ok:
                    break
        if self.world > 1:       # everybody or nobody
            flags = [None] * self.world
            dist.all_gather_object(flags, ok, group=self.group)
            ok = all(flags)
        if not ok:
            msg = lib.vdx_last_error()
            for pid, h in opened:
                _ipc_release(lib, pid, h)
            warnings.warn(f"ShardedStore: peer mapping of the shard arenas failed ({msg.decode() if msg else 'export failed'}); "
                          "using the collective all-gather")
            return
        self._peer_ptrs, self._opened, self.transport = ptrs, opened, "peer"
        if self.world > 1:
            # self-check, once: the first unit pulled through the mapped arenas must equal the same unit gathered by the
            # collective (a mapping that opens but reads something else must not go unnoticed); everybody or nobody
            unit = self.schedule[0]
            n = self._padded[unit]
            a, b = self._bufs[0][:n], self._bufs[1][:n]
            with torch.cuda.stream(self._side):
                self._peer_gather(a, unit)
            self._side.synchronize()
            if dist.get_backend(self.group) != "gloo":
                dist.all_gather_into_tensor(b, self.shards[unit], group=self.group)
            else:
                dist.all_gather(list(b.chunk(self.world)), self.shards[unit], group=self.group)
            torch.cuda.synchronize(self.device)
            # BITWISE: the packed blobs of the fused kernels carry fp32 biases inside fp16-typed tensors, and some of those
            # bit patterns read as fp16 NaNs — a float comparison calls identical bytes different (this, not the mapping,
            # failed round 5's self-check: 96 + 326 such elements in the first unit, on the rank's OWN slice too)
            ai_, bi_ = a.view(torch.int16), b.view(torch.int16)
            same = bool(torch.equal(ai_, bi_))
            detail = ""
            if not same:       # which rank's slice differs, and how much of it: the first thing anybody debugging this needs
                per = n // self.world
                bad = [(r, int((ai_[r * per:(r + 1) * per] != bi_[r * per:(r + 1) * per]).sum())) for r in range(self.world)]
                detail = (f" [rank {self.rank}: unit {unit!r}, {per} elements per slice, differing elements per owner rank "
                          f"{bad}; arena {self._arena_off[unit]}, {len(self._arenas)} arenas]")
            flags = [None] * self.world
            dist.all_gather_object(flags, same, group=self.group)
            if not all(flags):
                warnings.warn("ShardedStore: the peer-mapped gather does not reproduce the collective one; using the collective all-gather" + detail)
                self._close_peer()
            self.peer_self_check = "passed" if self.transport == "peer" else "failed"

    def _close_peer(self):
        """Drop this store's references to the peers' mappings (the last reference closes the mapping) and fall back to
        the collective transport.  Outstanding copies are waited for first."""
        if not getattr(self, "_opened", None):      # nothing mapped (world of 1, collective transport, failed setup)
            return
        from . import _lib
        lib = _lib.load()
        if self._side is not None:
            self._side.synchronize()
        for pid, h in self._opened:
            _ipc_release(lib, pid, h)
        self._opened, self._peer_ptrs, self.transport = [], None, "collective"

    def close(self):
        """Release the peer mappings (idempotent).  A COLLECTIVE call for a world > 1: every rank of the group must call it
        at the same point of the program — afterwards the gathers go through the collective, and a rank that switched alone
        would wait in its next all-gather for peers that still pull through their mappings.  Garbage
        collection of a store (`__del__`) on one rank only is safe only when that rank gathers no more."""
        self._close_peer()

    def __del__(self):
        try:
            self._close_peer()
        except Exception:       # interpreter teardown: the process's mappings go with it
            pass

    def _peer_gather(self, out: torch.Tensor, unit: str):
        from . import _lib
        es = out.element_size()
        n = self._padded[unit] // self.world
        k = self.rehearse_copies
        ai, ao = self._arena_off[unit]
        if k > 1 and self.world == 1 and (n * es) % (16 * k) == 0:
            # one-GPU rehearsal of a bigger world (bench.py --as-world): the same bytes as `k` copies, so that the host
            # issues what it would issue on a node (one ctypes call, k hipMemcpyAsync per unit)
            base = self._peer_ptrs[0][ai] + ao * es
            srcs = (C.c_void_p * k)(*[base + i * (n * es // k) for i in range(k)])
            rc = _lib.load().vdx_peer_gather(out.data_ptr(), srcs, k, n * es // k, self._side.cuda_stream)
        else:
            srcs = (C.c_void_p * self.world)(*[p[ai] + ao * es for p in self._peer_ptrs])
            rc = _lib.load().vdx_peer_gather(out.data_ptr(), srcs, self.world, n * es, self._side.cuda_stream)
        _lib.check(rc, "vdx_peer_gather")

    # ---- mapping protocol -------------------------------------------------------------------
    def __contains__(self, name):
        return name in self._unit_of or name in self.replicated

    def __getitem__(self, name):
        t = self.replicated.get(name)
        if t is not None:
            return t
        u = self._unit_of[name]
        if u != self._current:
            self._switch_to(u)
        return self._views[name]

    def keys(self):
        return list(self.replicated) + list(self._unit_of)

    def local_bytes(self) -> int:
        es = torch.empty(0, dtype=self.dtype).element_size()
        return (sum(s.numel() for s in self.shards.values()) + sum(t.numel() for t in self.replicated.values())
                + sum(b.numel() for b in self._bufs)) * es

    # ---- gather machinery -------------------------------------------------------------------
    def _gather_into(self, slot: int, unit: str):
        """Enqueue the all-gather of `unit` into buffer `slot` (side stream on GPU)."""
        n = self._padded[unit]
        out = self._bufs[slot][:n]
        shard = self.shards[unit]

        def run():
            if self.transport == "peer":
                self._peer_gather(out, unit)
            elif self.world == 1 and not _FORCE_COLLECTIVE() and self.comm is None:
                out.copy_(shard)
            elif self._cuda and self.comm is not None:
                self.comm.allgather(shard, out, self._side)
            elif self._cuda and dist.get_backend(self.group) != "gloo":
                dist.all_gather_into_tensor(out, shard, group=self.group)
            else:
                dist.all_gather(list(out.chunk(self.world)), shard, group=self.group)

        t0 = time.perf_counter()
        if self._cuda:
            with torch.cuda.stream(self._side):
                if self._released[slot] is not None:
                    self._side.wait_event(self._released[slot])     # old contents no longer needed
                    self._released[slot] = None                     # (the event belonged to the old contents)
                run()
                if self.rehearse_hog is not None:
                    # one-GPU rehearsal: hold `blocks` CUs for as long as a ring all-gather of this group's REMOTE bytes
                    # would take at the modelled rate — what the step's persistent grids see beside a real collective
                    from . import ops
                    blocks, lds, gbs, as_world = self.rehearse_hog
                    remote = n * out.element_size() * (as_world - 1) / as_world
                    ops.occupancy_hog(blocks, lds, max(1, int(remote / (gbs * 1e3))), self._side)
                ev = torch.cuda.Event()
                ev.record(self._side)
            self._ready[slot] = ev
        else:
            run()
        self.gather_host_s += time.perf_counter() - t0
        self._resident[slot] = unit
        self.gathers += 1

    def _pick_slot(self, pos: int, protect) -> int:
        """Buffer to overwrite: never one in `protect`; an empty one if any; else the one whose group is needed LATEST
        counting forward from schedule position `pos` (the group just left is L - 1 steps away: the first to go)."""
        best, best_d = None, -1
        L = len(self.schedule)
        for s_, u in enumerate(self._resident):
            if s_ in protect:
                continue
            if u is None:
                return s_
            d = (self._pos[u] - pos) % L
            if d > best_d:
                best, best_d = s_, d
        return best

    def _switch_to(self, unit: str):
        if self.trace is not None and self._cuda:      # diagnostic: when does the compute stream ENTER each gather group
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self.trace.append((unit, ev))
        # release the buffer of the unit we are leaving
        if self._current is not None and self._cuda and self._current in self._resident:
            old = self._resident.index(self._current)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._released[old] = ev
        pos = self._pos[unit]
        if unit in self._resident:
            slot = self._resident.index(unit)
        else:                                                   # not prefetched: gather on demand
            slot = self._pick_slot(pos, ())
            self._gather_into(slot, unit)
        if self._cuda and self._ready[slot] is not None:
            torch.cuda.current_stream(self.device).wait_event(self._ready[slot])
        views = self._view_cache.get((slot, unit))
        if views is None:
            buf = self._bufs[slot]
            views = {name: buf[off:off + _numel(shape)].view(shape) for name, off, shape in self._layout[unit]}
            self._view_cache[(slot, unit)] = views
        self._views = views
        self._current = unit
        # prefetch the next `prefetch_depth` units of the schedule into the buffers that hold nothing still needed
        L = len(self.schedule)
        ahead = []
        for d in range(1, self.prefetch_depth + 1):
            nxt = self.schedule[(pos + d) % L]
            if nxt == unit or nxt in ahead:
                break
            ahead.append(nxt)
        for nxt in ahead:
            if nxt in self._resident:
                continue
            protect = {slot} | {self._resident.index(u) for u in ahead if u in self._resident}
            tgt = self._pick_slot(pos, protect)
            if tgt is None:
                break
            self._gather_into(tgt, nxt)


def _merge_units(tensors, unit_of, schedule, merge_bytes):
    """Greedy merge of consecutive schedule units into gather groups of at most `merge_bytes` (a unit larger than that
    stays alone).  Returns the group schedule and the tensor-name -> group function."""
    es = next(iter(tensors.values())).element_size()
    size = {u: 0 for u in schedule}
    for name, t in tensors.items():
        u = unit_of(name)
        if u is not None:
            if u not in size:
                raise KeyError(f"tensor {name!r} maps to unit {u!r} which is not in the schedule")
            size[u] += _round_up(t.numel(), ALIGN) * es
    # (never coarser than 1/24 of the model: a small model keeps a schedule worth prefetching through)
    merge_bytes = min(merge_bytes, sum(size.values()) // 24)
    group_of, groups, cur, cur_bytes = {}, [], None, 0
    for u in schedule:
        if cur is None or cur_bytes + size[u] > merge_bytes:
            cur, cur_bytes = u, 0                      # a group is named after its first unit
            groups.append(cur)
        group_of[u] = cur
        cur_bytes += size[u]

    def unit_of_group(name):
        u = unit_of(name)
        return None if u is None else group_of[u]
    return groups, unit_of_group


def _numel(shape):
    n = 1
    for s in shape:
        n *= s
    return n
