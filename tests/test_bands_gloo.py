"""CPU, world_size 2 (gloo): the row-band split + gather of raytracing_amd.bands reassembles exactly the single-process
frame.  The band renderer here is the ORACLE (tests may use it); on the GPU the same host code is fed by the HIP kernel."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from raytracing_amd import bands, scene, scenegen

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_band_rows_partition():
    for h in (1, 7, 100, 2160, 4320):
        for world in (1, 2, 3, 4, 8):
            rows = [bands.band_rows(h, world, r) for r in range(world)]
            assert rows[0][0] == 0 and rows[-1][1] == h
            for (a0, a1), (b0, b1) in zip(rows[:-1], rows[1:]):
                assert a1 == b0 and a0 <= a1
            assert bands.max_band_rows(h, world) == max(b - a for a, b in rows)
    assert bands.band_rows(2160, 8, 3) == (810, 1080)
    with pytest.raises(ValueError):
        bands.band_rows(10, 2, 2)


def test_cost_balanced_cuts():
    """bands.balanced_cuts / fit_cost_model (used by tools/band_balance.py): contiguous, aligned, complete, and they equalise the cost."""
    h = 2160
    cover = np.zeros(h)
    cover[700:1500] = 0.5
    for world in (1, 2, 4, 8):
        assert bands.equal_cuts(h, world) == [bands.band_rows(h, world, r)[0] for r in range(world)] + [h]
        cuts = bands.balanced_cuts(1.0 + 0.5 * cover, world)
        assert cuts[0] == 0 and cuts[-1] == h and len(cuts) == world + 1
        assert all(a <= b for a, b in zip(cuts[:-1], cuts[1:])) and all(c % 16 == 0 for c in cuts[1:-1])
        cost = [float((1.0 + 0.5 * cover[a:b]).sum()) for a, b in zip(cuts[:-1], cuts[1:])]
        assert max(cost) / (sum(cost) / world) < 1.06  # 16-row alignment: +-8 rows of ~270
        assert bands.cut_band(cuts, world - 1)[1] == h
    # the fit recovers kappa from exact band times of the equal split
    cuts = bands.equal_cuts(h, 8)
    times = [3.0 * ((b - a) + 0.2 * cover[a:b].sum()) for a, b in zip(cuts[:-1], cuts[1:])]
    assert abs(bands.fit_cost_model(cuts, times, cover) - 0.2) < 1e-9
    assert bands.fit_cost_model(cuts, [1.0] * 8, np.zeros(h)) == 0.0  # degenerate: no coverage anywhere
    with pytest.raises(ValueError):
        bands.balanced_cuts(np.zeros(10), 2)


def test_frame_gather_list_views():
    frame = torch.zeros((8, 3, 4), dtype=torch.uint8)
    views = bands.frame_gather_list(frame, 8, 4)
    assert len(views) == 4 and all(v.is_contiguous() and v.shape == (2, 3, 4) for v in views)
    views[2].fill_(7)
    assert frame[4:6].eq(7).all() and frame[:4].eq(0).all() and frame[6:].eq(0).all()
    assert bands.frame_gather_list(frame, 8, 3) is None  # unequal bands: no zero-copy layout


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, w, h, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_py
    orc = oracle_py.load()
    objs, lights = scenegen.generate(w, h, 40, 3, seed=5)
    m = orc.default_camera(w, h)
    rows_max = bands.max_band_rows(h, world)

    def render_band(y0, y1):
        band = torch.zeros((rows_max, w, 4), dtype=torch.uint8)
        band[: y1 - y0] = torch.from_numpy(orc.render(objs, lights, m, w, h, y0=y0, y1=y1, nthreads=1))
        return band

    frame = bands.render_frame_distributed(render_band, h, world, rank)
    if rank == 0:
        np.save(os.path.join(outdir, "frame.npy"), frame.numpy())
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("h", [60, 61])
def test_two_rank_gather_matches_single_process(tmp_path, oracle, h):
    w, world = 80, 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, w, h, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "frame.npy")
    objs, lights = scenegen.generate(w, h, 40, 3, seed=5)
    ref = oracle.render(objs, lights, oracle.default_camera(w, h), w, h)
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def test_single_rank_needs_no_process_group(oracle):
    w, h = 32, 20
    elements, lights = scene.default_scene()
    objs, lts = scene.flatten_elements(elements), scene.flatten_lights(lights)
    m = oracle.default_camera(w, h)
    frame = bands.render_frame_distributed(lambda y0, y1: torch.from_numpy(oracle.render(objs, lts, m, w, h, y0=y0, y1=y1)), h, 1, 0)
    assert np.array_equal(frame.numpy(), oracle.render(objs, lts, m, w, h))


def _pipe_worker(rank, world, port, w, h, nframes, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tests import oracle_py
    orc = oracle_py.load()
    rows_max = bands.max_band_rows(h, world)
    y0, y1 = bands.band_rows(h, world, rank)
    pipe = bands.GatherPipeline(world, rank, slots=2)
    band_bufs = [torch.zeros((rows_max, w, 4), dtype=torch.uint8) for _ in range(2)]
    glists = [[torch.empty_like(band_bufs[0]) for _ in range(world)] for _ in range(2)] if rank == 0 else [None, None]
    m = orc.default_camera(w, h)
    frames = []
    pending = []
    for k in range(nframes):
        objs, lights = scenegen.generate(w, h, 20, 2, seed=100 + k)  # a different scene per frame
        s = pipe.begin(k)
        if rank == 0 and len(pending) == 2:  # slot s's previous gather has completed: consume it before it is overwritten
            frames.append(bands.assemble(glists[pending.pop(0)], h, world).clone())
        band_bufs[s][: y1 - y0] = torch.from_numpy(orc.render(objs, lights, m, w, h, y0=y0, y1=y1, nthreads=1))
        pipe.submit(k, band_bufs[s], glists[s])
        pending.append(s)
    pipe.drain()
    if rank == 0:
        while pending:
            frames.append(bands.assemble(glists[pending.pop(0)], h, world).clone())
        np.save(os.path.join(outdir, "frames.npy"), torch.stack(frames).numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_gather_pipeline_two_ranks(tmp_path, oracle):
    """The double-buffered gather pipeline used by bench.py delivers every frame intact and in order."""
    w, h, world, nframes = 48, 37, 2, 5
    mp.spawn(_pipe_worker, args=(world, _free_port(), w, h, nframes, str(tmp_path)), nprocs=world, join=True)
    got = np.load(tmp_path / "frames.npy")
    assert got.shape == (nframes, h, w, 4)
    m = oracle.default_camera(w, h)
    for k in range(nframes):
        objs, lights = scenegen.generate(w, h, 20, 2, seed=100 + k)
        assert np.array_equal(got[k], oracle.render(objs, lights, m, w, h)), f"frame {k}"


def test_gather_pipeline_single_rank_is_noop():
    pipe = bands.GatherPipeline(1, 0)
    for k in range(4):
        assert pipe.begin(k) == k % 2
        pipe.submit(k, torch.zeros(1))
    pipe.drain()


def _slots_worker(rank, world, port, outdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    got = [bands.agree_slots(4, 4 if rank == 0 else 3, world),   # one rank found fewer concurrent streams
           bands.agree_slots(4, 4, world),
           bands.agree_slots(4, 1 if rank == 1 else 4, world),   # never below two frames in flight
           bands.agree_slots(1, 4, world)]                       # unless one slot was asked for
    np.save(os.path.join(outdir, f"slots{rank}.npy"), np.array(got))
    dist.destroy_process_group()


def test_ranks_agree_on_the_number_of_frame_slots(tmp_path):
    """bench.py rotates frames over as many slots as concurrent streams exist; every rank must end up with the same number."""
    assert bands.agree_slots(4, 3, 1) == 3 and bands.agree_slots(3, 16, 1) == 3 and bands.agree_slots(4, 0, 1) == 2
    with pytest.raises(ValueError):
        bands.agree_slots(0, 4, 1)
    mp.spawn(_slots_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    a, b = np.load(tmp_path / "slots0.npy"), np.load(tmp_path / "slots1.npy")
    assert a.tolist() == b.tolist() == [3, 4, 2, 1]
