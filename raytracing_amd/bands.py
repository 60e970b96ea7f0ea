"""Row-band split of one frame across ranks and the gather of the bands to rank 0.

The reference parallelises over image rows with rayon (display.rs:90-93); across GPUs the same axis is cut into
contiguous bands: rank r renders rows [r*H/G, (r+1)*H/G) with GLOBAL row indices in the ray formula (display.rs:99),
no exchange during rendering, one gather (RCCL over xGMI through torch.distributed's "nccl" backend; "gloo" on CPU in
the tests) of the finished bands to rank 0.  torch is plumbing here: device memory, streams, the process group.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple


def band_rows(h: int, world: int, rank: int) -> Tuple[int, int]:
    """Rows [y0, y1) of rank `rank` when an h-row frame is cut into `world` contiguous bands."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad world/rank")
    return (rank * h) // world, ((rank + 1) * h) // world


def max_band_rows(h: int, world: int) -> int:
    return max(band_rows(h, world, r)[1] - band_rows(h, world, r)[0] for r in range(world))


# ---- cost-balanced cuts: measured by tools/band_balance.py, NOT used by bench.py (profiles/r02_tuning.md: 0-2 % at best, and worse where a
# band crosses a tile-round boundary of the primary kernel) ----

def equal_cuts(h: int, world: int):
    """Cut rows of the equal split: cuts[r] .. cuts[r + 1] is band r (the same rows as band_rows)."""
    return [(r * h) // world for r in range(world)] + [h]


def cut_band(cuts, rank: int) -> Tuple[int, int]:
    return int(cuts[rank]), int(cuts[rank + 1])


def fit_cost_model(cuts, times, row_cover):
    """Least-squares fit of  time(band) = a * rows + b * (hit pixels per row, summed over the band's rows)  to measured per-band
    times: the filter loop costs the same for every pixel, the exact tail and the shading only where rays hit, so a dense band of
    equal height takes longer (C3 at 8 bands: 79 against 71 us).  Returns the relative extra cost of a fully covered row,
    kappa = b / a >= 0 (0 when the fit is degenerate)."""
    import numpy as np
    cover = np.asarray(row_cover, dtype=np.float64)
    rows = np.array([cuts[r + 1] - cuts[r] for r in range(len(cuts) - 1)], dtype=np.float64)
    hits = np.array([cover[cuts[r]:cuts[r + 1]].sum() for r in range(len(cuts) - 1)], dtype=np.float64)
    t = np.asarray(times, dtype=np.float64)
    ok = (rows > 0) & np.isfinite(t) & (t > 0)
    if ok.sum() < 2:
        return 0.0
    A = np.stack([rows[ok], hits[ok]], axis=1)
    if np.linalg.matrix_rank(A) < 2:
        return 0.0
    (a, b), *_ = np.linalg.lstsq(A, t[ok], rcond=None)
    if not (a > 0) or not np.isfinite(b):
        return 0.0
    return float(min(max(b / a, 0.0), 4.0))


def balanced_cuts(row_cost, world: int, align: int = 16):
    """Cut rows of contiguous bands with (nearly) equal summed cost: cuts[0] = 0 < ... <= cuts[world] = h, interior cuts rounded to
    multiples of `align` rows (16 = the efficient tile height of the primary kernel).  Deterministic: every rank computes the same
    cuts from the same costs.  A rank may end up with an empty band only if there are fewer aligned rows than ranks."""
    import numpy as np
    cost = np.asarray(row_cost, dtype=np.float64)
    h = int(cost.size)
    if world < 1 or h < 1 or not np.isfinite(cost).all() or (cost < 0).any() or cost.sum() <= 0:
        raise ValueError("need a non-negative finite cost per row with a positive sum")
    cum = np.concatenate([[0.0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        y = int(np.searchsorted(cum, cum[-1] * r / world, side="left"))
        if y > 0 and abs(cum[y - 1] - cum[-1] * r / world) < abs(cum[min(y, h)] - cum[-1] * r / world):
            y -= 1
        y = int(round(y / align)) * align
        cuts.append(min(max(y, cuts[-1]), h))
    cuts.append(h)
    return cuts


def agree_slots(wanted: int, found: int, world: int, group=None, device=None, floor: int = 2) -> int:
    """Number of frames every rank keeps in flight.  All ranks must rotate over the SAME number of slots (the slot of a frame picks
    its stream, its buffers and its communicator), but each rank may have found a different number of mutually concurrent streams:
    take the smallest `found` of any rank, at most `wanted`, and never fewer than min(floor, wanted) (floor 2: two frames in flight
    still overlap the gather with the next render, on shared queues if need be).  `wanted` and `floor` must be the same on every rank.
    One rank: no collective."""
    if wanted < 1:
        raise ValueError("at least one slot")
    lowest = found
    if world > 1:
        import torch
        import torch.distributed as dist
        t = torch.tensor([found], dtype=torch.int32, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        lowest = int(t[0])
    return max(min(floor, wanted), min(wanted, lowest))


def frame_gather_list(frame, h: int, world: int):
    """Rank 0: the gather's receive list as row views INTO `frame` ([h, w, ch], contiguous), so the collective itself
    assembles the frame and no copy follows.  Possible when every band has the same height (h % world == 0: the views are
    disjoint and each is contiguous); returns None otherwise (use separate buffers + assemble())."""
    if h % world != 0 or frame.shape[0] != h or not frame.is_contiguous():
        return None
    rows = h // world
    return [frame[r * rows:(r + 1) * rows] for r in range(world)]


def gather_bands(local_band, h: int, world: int, rank: int, group=None, gather_list=None):
    """Gather per-rank bands (torch uint8 tensors [rows_r, w, ch], padded to the largest band) to rank 0.

    Returns the assembled [h, w, ch] frame on rank 0, None elsewhere.  `local_band` must already have
    max_band_rows(h, world) rows (only the first rows_r are meaningful) so every rank contributes an equal-size buffer
    to the collective.  With world == 1 no collective is issued.
    """
    import torch
    import torch.distributed as dist

    rows_max = max_band_rows(h, world)
    if local_band.shape[0] != rows_max:
        raise ValueError(f"band buffer must have {rows_max} rows, got {local_band.shape[0]}")
    if world == 1:
        return local_band[:h]
    if rank == 0:
        if gather_list is None and h % world == 0:  # equal bands: receive straight into the frame
            frame = torch.empty((h,) + tuple(local_band.shape[1:]), dtype=local_band.dtype, device=local_band.device)
            dist.gather(local_band, gather_list=frame_gather_list(frame, h, world), dst=0, group=group)
            return frame
        if gather_list is None:
            gather_list = [torch.empty_like(local_band) for _ in range(world)]
        dist.gather(local_band, gather_list=gather_list, dst=0, group=group)
        parts = []
        for r in range(world):
            y0, y1 = band_rows(h, world, r)
            parts.append(gather_list[r][: y1 - y0])
        return torch.cat(parts, dim=0)
    dist.gather(local_band, gather_list=None, dst=0, group=group)
    return None


class GatherPipeline:
    """Double-buffered render -> gather pipeline: the gather of frame k (async collective) overlaps the render of frame k+1.

    Usage per frame k:   s = pipe.begin(k)          # makes the current stream wait until slot s's previous gather is done
                         <render the band into band_buffers[s]>
                         pipe.submit(k, band_buffers[s], gather_lists[s] on rank 0 else None)
    and pipe.drain() at the end.  With world == 1 every call is a no-op (no collective exists).  Works with "nccl"
    (RCCL; Work.wait() is a stream-level wait, the host never blocks) and with "gloo" (CPU tests; wait() blocks).
    """

    def __init__(self, world: int, rank: int, slots: int = 2, group=None, always_collective: bool = False):
        if slots < 1:
            raise ValueError("need at least one slot")
        self.world, self.rank, self.slots, self.group = world, rank, slots, group
        self.always_collective = always_collective  # issue the (trivial) collective even with one rank: rehearsal of the N > 1 path
        self._work = [None] * slots

    def begin(self, k: int) -> int:
        s = k % self.slots
        if self._work[s] is not None:
            self._work[s].wait()
            self._work[s] = None
        return s

    def submit(self, k: int, band, gather_list=None):
        if self.world == 1 and not self.always_collective:
            return
        import torch.distributed as dist
        s = k % self.slots
        if self._work[s] is not None:
            raise RuntimeError("slot reused without begin()")
        if self.rank == 0 and gather_list is None:
            raise ValueError("rank 0 needs a gather_list")
        self._work[s] = dist.gather(band, gather_list=gather_list if self.rank == 0 else None, dst=0, group=self.group, async_op=True)

    def drain(self):
        for s in range(self.slots):
            if self._work[s] is not None:
                self._work[s].wait()
                self._work[s] = None


def assemble(gather_list, h: int, world: int):
    """Rank 0: cut the padding off the gathered bands and stack them into the [h, w, ch] frame."""
    import torch
    parts = []
    for r in range(world):
        y0, y1 = band_rows(h, world, r)
        parts.append(gather_list[r][: y1 - y0])
    return torch.cat(parts, dim=0)


def render_frame_distributed(render_band: Callable[[int, int], "object"], h: int, world: int, rank: int, group=None):
    """render_band(y0, y1) -> padded band tensor; returns the full frame on rank 0."""
    y0, y1 = band_rows(h, world, rank)
    band = render_band(y0, y1)
    return gather_bands(band, h, world, rank, group)
