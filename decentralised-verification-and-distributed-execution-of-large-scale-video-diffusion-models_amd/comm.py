"""RCCL communicator behind the C-ABI (`vdx_comm_*`, `vdx_allgather_shard`, `vdx_halo_exchange`; include/vdx.h): the
native form of the path's two exchange steps — the per-unit parameter all-gather (the reference's FSDP wrap,
`fsdp_chunked_coherent.py:63-88`) and the post-loop overlap-frame exchange (`:190-202`).

`Comm.from_torch()` bootstraps it from an initialised `torch.distributed` process group (the 128-byte id travels by
broadcast).  `ShardedStore(comm=...)` and `exchange_halos(comm=...)` then issue their collectives through the library on
their side streams instead of through `torch.distributed`; the default stays `torch.distributed` (backend "nccl" = the
same RCCL), which is what the multi-process tests exercise.  Status: world-1 round trips are tested on the GPU box;
a multi-GPU node was not available to this build.
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import VdxError


class Comm:
    def __init__(self, handle, rank: int, world: int):
        self._h, self.rank, self.world = handle, rank, world

    @classmethod
    def create(cls, unique_id: bytes, rank: int, world: int) -> "Comm":
        lib = _lib.load()
        if len(unique_id) != 128:
            raise VdxError("Comm: the RCCL unique id is 128 bytes")
        h = C.c_void_p()
        _lib.check(lib.vdx_comm_init(C.c_char_p(unique_id), rank, world, C.byref(h)), "vdx_comm_init")
        return cls(h, rank, world)

    @staticmethod
    def unique_id() -> bytes:
        lib = _lib.load()
        buf = C.create_string_buffer(128)
        _lib.check(lib.vdx_comm_unique_id(buf), "vdx_comm_unique_id")
        return buf.raw

    @classmethod
    def from_torch(cls, device) -> "Comm":
        """Collective over the default process group: rank 0 makes the id, everybody joins."""
        rank, world = dist.get_rank(), dist.get_world_size()
        idt = torch.zeros(128, dtype=torch.uint8, device=device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(cls.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, src=0)
        return cls.create(bytes(idt.cpu().tolist()), rank, world)

    def allgather(self, shard: torch.Tensor, full: torch.Tensor, stream: torch.cuda.Stream) -> None:
        n = shard.numel() * shard.element_size()
        if not (shard.is_cuda and full.is_cuda and shard.is_contiguous() and full.is_contiguous()):
            raise VdxError("Comm.allgather: contiguous GPU tensors expected")
        if full.numel() * full.element_size() != n * self.world:
            raise VdxError("Comm.allgather: full must hold world x shard bytes")
        _lib.check(_lib.load().vdx_allgather_shard(self._h, shard.data_ptr(), full.data_ptr(), n, stream.cuda_stream),
                   "vdx_allgather_shard")

    def halo(self, send, send_to: int, recv, recv_from: int, stream: torch.cuda.Stream) -> None:
        sb = send.numel() * send.element_size() if send is not None else 0
        rb = recv.numel() * recv.element_size() if recv is not None else 0
        for t in (send, recv):
            if t is not None and not (t.is_cuda and t.is_contiguous()):
                raise VdxError("Comm.halo: contiguous GPU tensors expected")
        _lib.check(_lib.load().vdx_halo_exchange(self._h, send.data_ptr() if sb else None, sb, send_to,
                                                 recv.data_ptr() if rb else None, rb, recv_from, stream.cuda_stream),
                   "vdx_halo_exchange")

    def destroy(self) -> None:
        if self._h:
            _lib.check(_lib.load().vdx_comm_destroy(self._h), "vdx_comm_destroy")
            self._h = None
