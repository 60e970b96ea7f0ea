"""CPU: raytracing_amd.rccl_gather against a RECORDING FAKE of librccl (no GPU, no RCCL): which rank sends, which receives, the
addresses (frame_ptr + y0 * w * 4), the byte counts, one ncclGroupStart / ncclGroupEnd pair per frame, nothing issued for an empty
band — for world sizes 2, 3, 4, 8 and even, odd and tiny heights.  The fake also "moves" the bytes (a dict of numpy buffers keyed by
address), so the assembled frame is checked too.  And create_agreed (collective-safe set-up) with two gloo ranks where one rank fails
in each phase: both ranks must come back with None — no hang, no split decision."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import bands, rccl_gather

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _val(x):
    return x.value if hasattr(x, "value") else x


class FakeRccl:
    """Records every call; ncclSend/ncclRecv of all ranks share `wire` so a test can replay the transfers."""

    def __init__(self, log=None, fail=None):
        self.log = log if log is not None else []
        self.fail = fail or {}   # function name -> return code
        self.depth = 0
        self.next_comm = 0x1000

    def _rc(self, name):
        return self.fail.get(name, 0)

    def ncclGetErrorString(self, rc):
        return b"fake error %d" % rc

    def ncclGetUniqueId(self, uid_ref):
        uid = uid_ref._obj
        C.memmove(C.byref(uid), bytes([0, 7] + [0] * 125 + [9]), 128)  # NUL bytes inside: must survive the hand-out intact
        self.log.append(("uid",))
        return self._rc("ncclGetUniqueId")

    def ncclCommInitRank(self, comm_ref, world, uid, rank):
        self.log.append(("init", world, C.string_at(C.byref(uid), 128), rank))
        rc = self._rc("ncclCommInitRank")
        if rc == 0:
            self.next_comm += 0x10
            comm_ref._obj.value = self.next_comm
        return rc

    def ncclCommDestroy(self, comm):
        self.log.append(("destroy", _val(comm)))
        return 0

    def ncclGroupStart(self):
        self.depth += 1
        self.log.append(("start",))
        return self._rc("ncclGroupStart")

    def ncclGroupEnd(self):
        self.depth -= 1
        self.log.append(("end",))
        return self._rc("ncclGroupEnd")

    def ncclSend(self, ptr, count, dtype, peer, comm, stream):
        assert self.depth == 1, "send outside a group"
        self.log.append(("send", _val(ptr), count, dtype, peer, _val(comm), _val(stream)))
        return self._rc("ncclSend")

    def ncclRecv(self, ptr, count, dtype, peer, comm, stream):
        assert self.depth == 1, "recv outside a group"
        self.log.append(("recv", _val(ptr), count, dtype, peer, _val(comm), _val(stream)))
        return self._rc("ncclRecv")


UID = bytes(range(128))


@pytest.mark.parametrize("world", [2, 3, 4, 8])
@pytest.mark.parametrize("h", [2160, 4320, 61, 5])
def test_every_rank_issues_exactly_its_part_of_the_gather(world, h):
    w, bpp = 48, 4
    row_bytes = w * bpp
    frame_ptr, stream = 0x7F0000000000, 0xABC0
    band_ptr = [0x100000000 * (r + 1) for r in range(world)]
    logs = []
    for rank in range(world):
        lib = FakeRccl()
        g = rccl_gather.RcclBandGather(world, rank, None, lib=lib, uid=UID)
        assert lib.log == [("init", world, UID, rank)]
        del lib.log[:]
        assert g.plan(h, w) == [(bands.band_rows(h, world, r)[0], bands.band_rows(h, world, r)[1] - bands.band_rows(h, world, r)[0],
                                 (bands.band_rows(h, world, r)[1] - bands.band_rows(h, world, r)[0]) * row_bytes) for r in range(world)]
        g.gather(h, w, stream, band_ptr=0 if rank == 0 else band_ptr[rank], frame_ptr=frame_ptr if rank == 0 else 0)
        logs.append(list(lib.log))
        g.close()
    # rank 0: one group, one receive per NON-EMPTY peer band, straight into that peer's rows of the frame; no send
    ops0 = logs[0]
    assert ops0[0] == ("start",) and ops0[-1] == ("end",) and sum(1 for o in ops0 if o[0] in ("start", "end")) == 2
    recvs = [o for o in ops0 if o[0] == "recv"]
    assert not [o for o in ops0 if o[0] == "send"]
    expect = []
    for r in range(1, world):
        y0, y1 = bands.band_rows(h, world, r)
        if y1 > y0:
            expect.append(("recv", frame_ptr + y0 * row_bytes, (y1 - y0) * row_bytes, rccl_gather.NCCL_UINT8, r, recvs[0][5] if recvs else 0, stream))
    assert recvs == expect
    # peers: one group with exactly one send of the whole band to rank 0 — or nothing at all for an empty band
    total = (bands.band_rows(h, world, 0)[1]) * row_bytes
    for rank in range(1, world):
        y0, y1 = bands.band_rows(h, world, rank)
        ops = logs[rank]
        if y1 == y0:
            assert ops == [], "an empty band issues nothing (not even a group)"
            continue
        assert [o[0] for o in ops] == ["start", "send", "end"]
        assert ops[1][1:5] == (band_ptr[rank], (y1 - y0) * row_bytes, rccl_gather.NCCL_UINT8, 0) and ops[1][6] == stream
        total += ops[1][2]
    assert total == h * row_bytes, "the bands tile the frame"
    # every receive has its matching send (peer, byte count)
    sends = {rank: logs[rank][1][2] for rank in range(1, world) if logs[rank]}
    assert {o[4]: o[2] for o in recvs} == sends


def test_replayed_transfers_assemble_the_frame():
    """Move the bytes the way the recorded calls say: the frame on rank 0 equals the concatenation of the bands."""
    world, h, w = 3, 61, 10
    row_bytes = w * 4
    rng = np.random.default_rng(0)
    full = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    frame = np.zeros_like(full)
    y0, y1 = bands.band_rows(h, world, 0)
    frame[y0:y1] = full[y0:y1]  # rank 0 renders in place
    frame_ptr = 1 << 40
    sent = {}
    recvs = []
    for rank in range(world):
        lib = FakeRccl()
        g = rccl_gather.RcclBandGather(world, rank, None, lib=lib, uid=UID)
        a, b = bands.band_rows(h, world, rank)
        g.gather(h, w, 5, band_ptr=0 if rank == 0 else 1000 + rank, frame_ptr=frame_ptr if rank == 0 else 0)
        for o in lib.log:
            if o[0] == "send":
                sent[rank] = full[a:b].tobytes()[: o[2]]
            elif o[0] == "recv":
                recvs.append(o)
    flat = frame.reshape(-1)
    for o in recvs:
        off = o[1] - frame_ptr
        flat[off:off + o[2]] = np.frombuffer(sent[o[4]], dtype=np.uint8)
    assert np.array_equal(frame, full)


def test_self_send_rehearsal_and_argument_errors():
    lib = FakeRccl()
    g = rccl_gather.RcclBandGather(1, 0, None, lib=lib, uid=UID)
    del lib.log[:]
    g.gather(100, 8, 3, frame_ptr=4096)  # one rank, no self-send: nothing to do
    assert lib.log == []
    g.gather(100, 8, 3, band_ptr=64, frame_ptr=4096, self_send=True)
    assert [o[0] for o in lib.log] == ["start", "recv", "send", "end"]
    assert lib.log[1][1:5] == (4096, 100 * 8 * 4, rccl_gather.NCCL_UINT8, 0) and lib.log[2][1:5] == (64, 100 * 8 * 4, rccl_gather.NCCL_UINT8, 0)
    del lib.log[:]
    g.loopback(64, 4096 + 32, 77, 3)
    assert [o[0] for o in lib.log] == ["start", "recv", "send", "end"] and lib.log[1][2] == 77
    with pytest.raises(ValueError):
        rccl_gather.RcclBandGather(2, 0, None, lib=FakeRccl(), uid=UID).gather(10, 4, 1)  # rank 0 without a frame
    with pytest.raises(ValueError):
        rccl_gather.RcclBandGather(2, 1, None, lib=FakeRccl(), uid=UID).gather(10, 4, 1)  # a peer without a band
    with pytest.raises(ValueError):
        rccl_gather.RcclBandGather(2, 1, None, lib=FakeRccl(), uid=b"short")


def test_group_is_closed_and_the_first_error_survives():
    """An error inside the group closes the group (ncclGroupEnd is still called) and is the exception that comes out, even when
    ncclGroupEnd fails too."""
    lib = FakeRccl(fail={"ncclSend": 5, "ncclGroupEnd": 3})
    g = rccl_gather.RcclBandGather(2, 1, None, lib=lib, uid=UID)
    del lib.log[:]
    with pytest.raises(RuntimeError, match="ncclSend failed"):
        g.gather(10, 4, 1, band_ptr=64)
    assert [o[0] for o in lib.log] == ["start", "send", "end"] and lib.depth == 0
    lib = FakeRccl(fail={"ncclGroupEnd": 3})
    g = rccl_gather.RcclBandGather(2, 1, None, lib=lib, uid=UID)
    with pytest.raises(RuntimeError, match="ncclGroupEnd failed"):
        g.gather(10, 4, 1, band_ptr=64)


def test_unique_id_keeps_all_128_bytes():
    assert rccl_gather.make_unique_id(FakeRccl()) == bytes([0, 7] + [0] * 125 + [9])


def test_create_agreed_single_rank():
    comms = rccl_gather.create_agreed(1, 0, None, 3, lib_loader=FakeRccl)
    assert len(comms) == 3 and all(c.world == 1 for c in comms)
    assert rccl_gather.create_agreed(1, 0, None, 2, lib_loader=lambda: (_ for _ in ()).throw(FileNotFoundError("no librccl"))) is None
    assert rccl_gather.create_agreed(1, 0, None, 2, lib_loader=FakeRccl, self_check=lambda ci, c: ci == 0) is None
    assert len(rccl_gather.create_agreed(1, 0, None, 2, lib_loader=FakeRccl, self_check=lambda ci, c: True)) == 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _agree_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def broken_loader():
        raise FileNotFoundError("librccl.so not found on this rank")

    def bad_init():
        return FakeRccl(fail={"ncclCommInitRank": 2})

    def check_fails_on_rank1(ci, comm):
        if rank == 1:
            raise RuntimeError("wrong byte in the gathered frame")
        return True

    results = []
    # every rank fine
    c = rccl_gather.create_agreed(world, rank, None, 2, lib_loader=FakeRccl, self_check=lambda ci, comm: True)
    results.append(-1 if c is None else len(c))
    ids = sorted({o[2] for cc in (c or []) for o in cc._lib.log if o[0] == "init"})
    results.append(len(ids))  # two communicators -> two ids (the fake hands out the same bytes twice: one distinct value)
    # phase 1 fails on rank 1 only, phase 2 fails on rank 0 only, phase 3 fails on rank 1 only: always None on BOTH ranks, no hang
    c = rccl_gather.create_agreed(world, rank, None, 2, lib_loader=broken_loader if rank == 1 else FakeRccl)
    results.append(-1 if c is None else len(c))
    c = rccl_gather.create_agreed(world, rank, None, 2, lib_loader=bad_init if rank == 0 else FakeRccl)
    results.append(-1 if c is None else len(c))
    c = rccl_gather.create_agreed(world, rank, None, 2, lib_loader=FakeRccl, self_check=check_fails_on_rank1)
    results.append(-1 if c is None else len(c))
    np.save(os.path.join(outdir, f"agree{rank}.npy"), np.array(results))
    dist.barrier()
    dist.destroy_process_group()


def test_create_agreed_two_ranks_fail_together(tmp_path):
    mp.spawn(_agree_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "agree0.npy"), np.load(tmp_path / "agree1.npy")
    assert a.tolist() == b.tolist() == [2, 1, -1, -1, -1]
