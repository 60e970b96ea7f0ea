"""Gather of the row bands to rank 0 with RCCL called directly (grouped ncclSend / ncclRecv over xGMI), no per-frame Python
collective machinery: one frame costs the host a handful of C calls instead of the ~35-60 us of a torch.distributed.gather.

Rank r renders rows [r*H/G, (r+1)*H/G) (bands.band_rows).  Rank 0 renders its own band straight into its rows of the frame and
receives every peer's band straight into that peer's rows (no staging, no assembly copy); rank r > 0 sends its band buffer.
Every call is enqueued on the stream the band was rendered on, so "render k -> gather k" is plain stream order, and frames that
rotate over several streams overlap (RCCL orders the operations of one communicator among themselves).

The library is the librccl.so torch already loaded (one RCCL per process); the communicator is this module's own, created from a
unique id that rank 0 broadcasts through the existing torch.distributed process group.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

from . import bands

NCCL_UINT8 = 1  # rccl.h: ncclUint8


class _UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]  # rccl.h: NCCL_UNIQUE_ID_BYTES


def _load_rccl() -> C.CDLL:
    import torch
    path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: the RCCL that torch.distributed uses was not found")
    lib = C.CDLL(path, mode=C.RTLD_GLOBAL)
    vp = C.c_void_p
    lib.ncclGetErrorString.restype = C.c_char_p
    lib.ncclGetErrorString.argtypes = [C.c_int]
    lib.ncclGetUniqueId.argtypes = [C.POINTER(_UniqueId)]
    lib.ncclCommInitRank.argtypes = [C.POINTER(vp), C.c_int, _UniqueId, C.c_int]
    lib.ncclCommDestroy.argtypes = [vp]
    lib.ncclSend.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclRecv.argtypes = [vp, C.c_size_t, C.c_int, C.c_int, vp, vp]
    lib.ncclGroupStart.argtypes = []
    lib.ncclGroupEnd.argtypes = []
    for f in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd"):
        getattr(lib, f).restype = C.c_int
    return lib


class RcclBandGather:
    """One per process (= per GPU).  `world`/`rank` are those of the torch.distributed group used to hand out the unique id
    (None: a one-rank communicator without any process group, for single-GPU rehearsals)."""

    def __init__(self, world: int, rank: int, device_index: int, group=None):
        import torch
        self.world, self.rank = world, rank
        self._lib = _load_rccl()
        self._comm = C.c_void_p()
        torch.cuda.set_device(device_index)
        uid = _UniqueId()
        if rank == 0:
            self._check(self._lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
        if world > 1:
            import torch.distributed as dist
            box = [bytes(uid.internal) if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            C.memmove(C.byref(uid), box[0], 128)
        self._check(self._lib.ncclCommInitRank(C.byref(self._comm), world, uid, rank), "ncclCommInitRank")

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise RuntimeError(f"{what} failed: {self._lib.ncclGetErrorString(rc).decode()} ({rc})")

    def plan(self, h: int, w: int, bytes_per_pixel: int = 4) -> List[Tuple[int, int, int]]:
        """(first row, rows, bytes) of every rank's band."""
        out = []
        for r in range(self.world):
            y0, y1 = bands.band_rows(h, self.world, r)
            out.append((y0, y1 - y0, (y1 - y0) * w * bytes_per_pixel))
        return out

    def gather(self, h: int, w: int, stream: int, band_ptr: int = 0, frame_ptr: int = 0, bytes_per_pixel: int = 4, self_send: bool = False):
        """Enqueue the gather of one frame on `stream` (hipStream_t as int).  Rank 0: `frame_ptr` = device address of the
        [h, w, bytes_per_pixel] frame whose own rows it has already rendered in place; other ranks: `band_ptr` = device address of
        their band.  `self_send` (one-rank rehearsal / test only): rank 0 also sends `band_ptr` to itself and receives it into its
        own rows, which exercises exactly the calls a peer and the root make."""
        lib, comm, st = self._lib, self._comm, C.c_void_p(stream)
        plan = self.plan(h, w, bytes_per_pixel)
        row_bytes = w * bytes_per_pixel
        if self.world == 1 and not self_send:
            return
        self._check(lib.ncclGroupStart(), "ncclGroupStart")
        try:
            if self.rank == 0:
                if not frame_ptr:
                    raise ValueError("rank 0 needs the frame address")
                for r, (y0, _rows, nbytes) in enumerate(plan):
                    if r == 0 and not self_send:
                        continue
                    if nbytes:
                        self._check(lib.ncclRecv(C.c_void_p(frame_ptr + y0 * row_bytes), nbytes, NCCL_UINT8, r, comm, st), "ncclRecv")
            if self.rank != 0 or self_send:
                if not band_ptr:
                    raise ValueError("a sending rank needs its band address")
                nbytes = plan[self.rank][2]
                if nbytes:
                    self._check(lib.ncclSend(C.c_void_p(band_ptr), nbytes, NCCL_UINT8, 0, comm, st), "ncclSend")
        finally:
            self._check(lib.ncclGroupEnd(), "ncclGroupEnd")

    def loopback(self, src_ptr: int, dst_ptr: int, nbytes: int, stream: int):
        """One ncclSend + ncclRecv of rank 0 with itself in one group (rehearsal of the per-frame traffic on a single GPU)."""
        lib, comm, st = self._lib, self._comm, C.c_void_p(stream)
        self._check(lib.ncclGroupStart(), "ncclGroupStart")
        try:
            self._check(lib.ncclRecv(C.c_void_p(dst_ptr), nbytes, NCCL_UINT8, self.rank, comm, st), "ncclRecv")
            self._check(lib.ncclSend(C.c_void_p(src_ptr), nbytes, NCCL_UINT8, self.rank, comm, st), "ncclSend")
        finally:
            self._check(lib.ncclGroupEnd(), "ncclGroupEnd")

    def close(self):
        if self._comm:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()
