"""Gather of the row bands to rank 0 with RCCL called directly (grouped ncclSend / ncclRecv over xGMI), no per-frame Python
collective machinery: one frame costs the host a handful of C calls instead of the ~35-60 us of a torch.distributed.gather.

Rank r renders rows [r*H/G, (r+1)*H/G) (bands.band_rows).  Rank 0 renders its own band straight into its rows of the frame and
receives every peer's band straight into that peer's rows (no staging, no assembly copy); rank r > 0 sends its band buffer.
Every call is enqueued on the stream the band was rendered on, so "render k -> gather k" is plain stream order, and frames that
rotate over several streams overlap (RCCL orders the operations of one communicator among themselves).

The library is the librccl.so torch already loaded (one RCCL per process); the communicator is this module's own, created from a
unique id that rank 0 broadcasts through the existing torch.distributed process group.

Set-up is collective-safe (create_agreed): every phase that can fail on ONE rank only (loading the library, making the unique ids,
the self-check's verdict) is followed by an all-reduce of an ok flag BEFORE the next blocking collective, so a rank that failed
never leaves its peers waiting in a broadcast, a communicator init or a send/recv: either every rank gets its communicators or
every rank gets None and the caller falls back to torch.distributed.gather.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Callable, List, Optional, Tuple

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


def _check(lib, rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib.ncclGetErrorString(rc).decode()} ({rc})")


def make_unique_id(lib) -> bytes:
    uid = _UniqueId()
    _check(lib, lib.ncclGetUniqueId(C.byref(uid)), "ncclGetUniqueId")
    return C.string_at(C.byref(uid), 128)  # all 128 bytes (`.internal` of a c_char array would stop at the first NUL)


class RcclBandGather:
    """One communicator of one process (= one GPU).  `uid` = the 128 unique-id bytes every rank of the communicator passes
    (made by rank 0 with make_unique_id and handed out by the caller; create_agreed does that collective-safely).  Without `uid`
    the constructor hands the id out itself through the torch.distributed group (a one-rank communicator needs no group).
    `lib` = the loaded RCCL (tests inject a recording fake); `device_index` is made current first unless it is None."""

    def __init__(self, world: int, rank: int, device_index: Optional[int], group=None, lib=None, uid: Optional[bytes] = None):
        self.world, self.rank = world, rank
        self._lib = lib if lib is not None else _load_rccl()
        self._comm = C.c_void_p()
        if device_index is not None:
            import torch
            torch.cuda.set_device(device_index)
        if uid is None:
            uid = make_unique_id(self._lib) if rank == 0 else None
            if world > 1:
                import torch.distributed as dist
                box = [uid]
                dist.broadcast_object_list(box, src=0, group=group)
                uid = box[0]
        if not isinstance(uid, (bytes, bytearray)) or len(uid) != 128:
            raise ValueError("the RCCL unique id is 128 bytes")
        u = _UniqueId()
        C.memmove(C.byref(u), bytes(uid), 128)
        self._check(self._lib.ncclCommInitRank(C.byref(self._comm), world, u, rank), "ncclCommInitRank")

    def _check(self, rc: int, what: str):
        _check(self._lib, rc, what)

    def plan(self, h: int, w: int, bytes_per_pixel: int = 4) -> List[Tuple[int, int, int]]:
        """(first row, rows, bytes) of every rank's band."""
        out = []
        for r in range(self.world):
            y0, y1 = bands.band_rows(h, self.world, r)
            out.append((y0, y1 - y0, (y1 - y0) * w * bytes_per_pixel))
        return out

    def _group(self, body: Callable[[], None]):
        """ncclGroupStart / body / ncclGroupEnd; the group is always closed, and an error of ncclGroupEnd never hides the exception
        that is already on its way out of `body`."""
        self._check(self._lib.ncclGroupStart(), "ncclGroupStart")
        try:
            body()
        except BaseException:
            self._lib.ncclGroupEnd()
            raise
        self._check(self._lib.ncclGroupEnd(), "ncclGroupEnd")

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
        if self.rank == 0 and not frame_ptr:
            raise ValueError("rank 0 needs the frame address")
        sends = self.rank != 0 or self_send
        if sends and not band_ptr and plan[self.rank][2]:
            raise ValueError("a sending rank needs its band address")
        if self.rank != 0 and plan[self.rank][2] == 0:
            return  # an empty band (more ranks than rows): nothing to send, and rank 0 posts no receive for it

        def body():
            if self.rank == 0:
                for r, (y0, _rows, nbytes) in enumerate(plan):
                    if r == 0 and not self_send:
                        continue
                    if nbytes:
                        self._check(lib.ncclRecv(C.c_void_p(frame_ptr + y0 * row_bytes), nbytes, NCCL_UINT8, r, comm, st), "ncclRecv")
            if sends:
                nbytes = plan[self.rank][2]
                if nbytes:
                    self._check(lib.ncclSend(C.c_void_p(band_ptr), nbytes, NCCL_UINT8, 0, comm, st), "ncclSend")

        self._group(body)

    def loopback(self, src_ptr: int, dst_ptr: int, nbytes: int, stream: int):
        """One ncclSend + ncclRecv of rank 0 with itself in one group (rehearsal of the per-frame traffic on a single GPU)."""
        lib, comm, st = self._lib, self._comm, C.c_void_p(stream)

        def body():
            self._check(lib.ncclRecv(C.c_void_p(dst_ptr), nbytes, NCCL_UINT8, self.rank, comm, st), "ncclRecv")
            self._check(lib.ncclSend(C.c_void_p(src_ptr), nbytes, NCCL_UINT8, self.rank, comm, st), "ncclSend")

        self._group(body)

    def close(self):
        if self._comm:
            self._lib.ncclCommDestroy(self._comm)
            self._comm = C.c_void_p()


def _all_ok(ok: bool, world: int, group=None, device=None) -> bool:
    """True iff `ok` on every rank (all-reduce MIN of a flag through the torch.distributed group)."""
    if world == 1:
        return bool(ok)
    import torch
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return int(t[0]) == 1


def create_agreed(world: int, rank: int, device_index: Optional[int], n_comms: int, group=None, device=None,
                  lib_loader: Callable[[], object] = _load_rccl,
                  self_check: Optional[Callable[[int, "RcclBandGather"], bool]] = None) -> Optional[List[RcclBandGather]]:
    """`n_comms` communicators on every rank, or None on every rank — never a mix, never a rank left waiting.

    Phases, each ending in an all-reduce of an ok flag before the next blocking collective may start:
      1. local only: load the library; rank 0 makes the unique ids;
      2. the ids are broadcast (a torch.distributed collective: every rank takes part, failed or not, because phase 1 was agreed);
         ncclCommInitRank on every rank — the one RCCL-side rendezvous; it only runs when every rank holds a loaded library;
      3. self_check(ci, comm) per communicator (enqueue a gather of known bytes and inspect it; it must not raise on one rank while
         its peers block in RCCL: the callback catches its own local errors and returns False) — its verdict is agreed too.
    `device` = where the flag tensors live (cuda:N with the nccl backend, None/cpu with gloo)."""
    lib, uids, comms = None, None, []
    why = ""
    try:
        lib = lib_loader()
        if rank == 0:
            uids = [make_unique_id(lib) for _ in range(n_comms)]
    except Exception as e:  # noqa: BLE001 - whatever went wrong locally, the ranks must agree on it
        why = "phase 1 (load / unique ids): %s" % e
        lib = None
    if not _all_ok(lib is not None, world, group, device):
        _say(rank, why or "another rank could not load RCCL")
        return None
    if world > 1:
        import torch.distributed as dist
        box = [uids]
        dist.broadcast_object_list(box, src=0, group=group)
        uids = box[0]
    try:
        for ci in range(n_comms):
            comms.append(RcclBandGather(world, rank, device_index, lib=lib, uid=uids[ci]))
    except Exception as e:  # noqa: BLE001
        why = "phase 2 (ncclCommInitRank): %s" % e
    if not _all_ok(len(comms) == n_comms, world, group, device):
        _say(rank, why or "another rank could not create its communicators")
        _close_all(comms)
        return None
    ok = True
    if self_check is not None:
        for ci, comm in enumerate(comms):
            try:
                ok = bool(self_check(ci, comm)) and ok
            except Exception as e:  # noqa: BLE001
                why = "phase 3 (self-check): %s" % e
                ok = False
    if not _all_ok(ok, world, group, device):
        _say(rank, why or "the gather self-check failed (here or on another rank)")
        _close_all(comms)
        return None
    return comms


def _close_all(comms):
    for c in comms:
        try:
            c.close()
        except Exception:  # noqa: BLE001
            pass


def _say(rank: int, msg: str):
    print("rank %d: direct RCCL gather unavailable: %s" % (rank, msg), file=sys.stderr)
