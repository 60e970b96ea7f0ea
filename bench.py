#!/usr/bin/env python3
"""bench.py — headline benchmark: primary rays/s on config C3 (3840x2160, 1024 spheres, 4 point lights, depth 1).

A step = one pass of the hot path over one frame: every rank renders its row band of the SAME frame with the HIP kernel
(C ABI rt_render_rows_device, scene already resident in HBM) and the bands are gathered to rank 0 (RCCL through
torch.distributed "nccl", double-buffered so the gather of frame k overlaps the render of frame k+1); with --gpus 1 no
collective is issued.  Total work per step is fixed (one frame) => strong scaling.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config C3|C4|C5] [--depth D] [--shadows] [--cull] [--no-cpu-baseline]
  N > 1:  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Rank 0 prints ONE JSON line (metric/value/roofline/cpu_baseline ...).  The oracle is used only for the cpu_baseline leg.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP32_VALU_TFLOPS = 157.3  # MI355X_MICROARCH.md: 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E ~8 TB/s
FLOPS_PER_TEST = 6             # the hot loop executes three v_fma_f32 per (pixel, sphere) pair (render_kernels.h)
FLOPS_PER_SHADOW_TEST = 6      # shadow_mask_kernel: the same 3-FMA filter per (shadow ray, sphere) with per-light constants
FLOPS_PER_BOUNCE_TEST = 14     # trace_bundle's filter: 7 FMAs per (bounce ray, sphere) it tests (round 2 walked the whole list with 10 operations, 16 flops)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C3", choices=["C3", "C4", "C5"])
    ap.add_argument("--depth", type=int, default=1, help="renderer.rs:32 depth; > 1 runs the bounce path (general-ray kernels)")
    ap.add_argument("--shadows", action="store_true", help="RT_SHADOWS extension (general-ray kernels)")
    ap.add_argument("--cull", action="store_true", help="RT_CULL (opt-in, same bytes out): primary rays through the cone hierarchy instead of the walk over the whole list")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    if args.depth < 1:
        raise SystemExit("--depth must be >= 1")

    # Keep stdout for the ONE JSON line: native libraries (RCCL prints a version banner at communicator creation) write to
    # fd 1 directly, so fd 1 is pointed at stderr for the run and the JSON goes to a saved duplicate of the real stdout.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    # the pool's host driver supports dmabuf IPC only: RCCL / cross-process device memory needs this before HIP starts (the GPU boxes
    # export it already; set here so that a hand-started multi-rank run works too)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    from raytracing_amd import bands, capi, scenegen

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the render path has no CPU fallback")
    # BENCH_BACKEND=gloo: REHEARSAL of the multi-rank control flow on a box with fewer GPUs than ranks (ranks share the GPUs, the
    # bands travel through host memory over gloo).  Everything a rank does except the link traffic runs for real — every rank's set-up
    # and agreement collectives, the direct-RCCL attempt and the agreed fall-back, the band offsets of rank > 0 — and rank 0 checks the
    # assembled frame against a single launch.  The JSON line is then NOT a benchmark result.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend not in ("nccl", "gloo"):
        raise SystemExit("BENCH_BACKEND must be nccl or gloo")
    dev_index = local_rank if backend == "nccl" else local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = dev if backend == "nccl" else torch.device("cpu")  # where the tensors of the control collectives live
    # BENCH_REHEARSE_DIST=1 runs the process-group / RCCL gather code path even with one rank (a 1-GPU box can then exercise it)
    use_dist = world > 1 or os.environ.get("BENCH_REHEARSE_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        # the band / frame buffers live for the whole run, so the process group need not pin them to its stream per call
        os.environ.setdefault("TORCH_NCCL_AVOID_RECORD_STREAMS", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    w, h, objs, lights = scenegen.config(args.config)
    m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
    ctx = capi.Context(dev_index)
    ctx.scene_upload(objs, lights)

    y0, y1 = bands.band_rows(h, world, rank)
    rows_max = bands.max_band_rows(h, world)
    # BENCH_REHEARSE_BAND=G (only with BENCH_REHEARSE_DIST=1, one rank): render just the band one of G ranks would, to measure the
    # host-side issue rate of render + collective at band size on a 1-GPU box.  The JSON line is then NOT a benchmark result.
    rehearse_band = int(os.environ.get("BENCH_REHEARSE_BAND", "0")) if (use_dist and world == 1) else 0
    if rehearse_band > 1:
        y0, y1 = bands.band_rows(h, rehearse_band, rehearse_band // 2)
        rows_max = bands.max_band_rows(h, rehearse_band)
    # Non-default streams: their handles are what the C ABI launches on, and the torch events below are recorded on them.
    # One rank, one stream: launches run back to back and the per-launch HIP-event durations are exclusive (clean roofline).
    # Several ranks: every rank renders only a band (270 rows at 8 GPUs), a launch too small to fill the chip on its own, so
    # frames rotate over up to four slots = four streams: the render of frame k+1 overlaps the drain of frame k and the RCCL gather
    # of frame k (measured on one GPU with 270-row bands: 103 -> 72 us per frame, tools/two_stream.py), and a gather may take up to
    # three frame times before it holds up its slot (its kernel only gets workgroup slots as render workgroups leave).  The streams
    # are probed for real concurrency (pick_concurrent_streams): as many slots as mutually concurrent streams, at most four.
    general = args.depth > 1 or args.shadows
    flags = (capi.RT_SHADOWS if args.shadows else 0) | (capi.RT_CULL if args.cull else 0)
    slots = int(os.environ.get("BENCH_SLOTS", "4")) if use_dist else 1
    if general:
        slots = min(slots, 2)  # general launches of one context are ordered by the library (shared hit records): two slots overlap the gather only
    want_torch = use_dist and os.environ.get("BENCH_GATHER") == "torch"
    found = 1
    try:
        streams = pick_concurrent_streams(torch, dev, slots) if slots > 1 else [torch.cuda.Stream(dev)]
        found = len(streams)
    except Exception as e:  # noqa: BLE001 - the probe is an optimisation; plain streams always work
        print("rank %d: stream probe failed (%s), using plain streams" % (rank, e), file=sys.stderr)
        streams = [torch.cuda.Stream(dev) for _ in range(min(slots, 3))]
        found = len(streams)
    torch.cuda.set_stream(streams[0])
    band_bufs = [torch.zeros((rows_max, w, 4), dtype=torch.uint8, device=dev) for _ in range(max(slots, 1))]
    # BENCH_GATHER=rccl (default): grouped ncclSend/ncclRecv called directly (raytracing_amd.rccl_gather), rank 0 renders its band in
    # place and receives the peers' bands straight into their rows of a [h, w, 4] frame (one frame per slot); everything of a frame
    # is enqueued on that frame's stream, so the host spends a few C calls per step.  BENCH_GATHER=torch: torch.distributed.gather
    # through bands.GatherPipeline (the same traffic; ~35-60 us more host time per step on rank 0, measured with tools/step_cost_probe.py).
    gather_mode = os.environ.get("BENCH_GATHER", "rccl") if use_dist else "none"
    if gather_mode not in ("none", "rccl", "torch"):
        raise SystemExit("BENCH_GATHER must be rccl or torch")
    frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device=dev) for _ in range(slots)] if (use_dist and rank == 0) else []
    rccl = None   # one communicator per slot: RCCL orders the operations of ONE communicator among themselves, so with a single
    # communicator the gathers of consecutive frames would run strictly one after the other (each pays its launch / hand-shake latency
    # serially); with a communicator per slot the gather of frame k+1 can start while frame k is still on the links.  All ranks issue
    # the frames, hence the communicators, in the same order.  BENCH_COMMS=1 restores the single communicator.
    n_comms = max(1, min(slots, int(os.environ.get("BENCH_COMMS", str(slots)))))
    pipe = None
    gather_lists = [None] * slots
    if gather_mode == "rccl":
        # Collective-safe set-up (rccl_gather.create_agreed): library + unique ids locally, agreement, communicators, agreement, then a
        # self-check before anything is timed — every rank fills its band with (rank + 1 + ci), the bands are gathered, rank 0 looks at
        # its frame — and a last agreement.  ALL ranks get their communicators or ALL fall back to torch.distributed.gather.
        from raytracing_amd import rccl_gather

        def self_check(ci, comm):
            if world == 1:
                return True
            band_bufs[0].fill_(rank + 1 + ci)
            if rank == 0:
                frames[0].zero_()
                frames[0][y0:y1].fill_(1 + ci)
            torch.cuda.synchronize(dev)
            comm.gather(h, w, streams[0].cuda_stream, band_ptr=0 if rank == 0 else band_bufs[0].data_ptr(), frame_ptr=frames[0].data_ptr() if rank == 0 else 0)
            streams[0].synchronize()
            good = True
            if rank == 0:
                for r in range(world):
                    a_, b_ = bands.band_rows(h, world, r)
                    good = good and bool((frames[0][a_:b_] == r + 1 + ci).all())
            return good

        rccl = rccl_gather.create_agreed(world, rank, dev_index, n_comms, device=cdev, self_check=self_check)
        band_bufs[0].zero_()
        if rccl is None:
            gather_mode = "torch"
    if use_dist:
        # every rank must rotate over the same number of slots (the slot picks the communicator): agree on the smallest number of
        # mutually concurrent streams any rank found, but keep at least two frames in flight
        wanted, usable, floor = slots, len(streams), 2
        if gather_mode == "torch" and "BENCH_SLOTS" not in os.environ:
            # torch.distributed.gather runs on the process group's OWN stream, tied to the frame streams by events in both directions
            # (issue: pg stream waits for the frame stream; Work.wait(): frame stream waits for the pg stream).  That stream is one more
            # client of the four hardware queues: keep frame streams + 1 within the number of streams the probe found to run
            # concurrently, so that no frame stream shares a queue with it ("the hang on record").
            wanted, usable, floor = min(slots, 3), found - 1, 1
        slots = bands.agree_slots(wanted, usable, world, device=cdev, floor=floor)
        while len(streams) < slots:
            streams.append(torch.cuda.Stream(dev))
        streams = streams[:slots]
        frames = frames[:slots] if rank == 0 else frames
        n_comms = min(n_comms, slots)
    if gather_mode == "torch":
        if rank == 0:
            for s_ in range(slots):
                views = bands.frame_gather_list(frames[s_], h, world) if rows_max * world == h else None  # (a band rehearsal has a short band)
                gather_lists[s_] = views or [torch.empty_like(band_bufs[0]) for _ in range(world)]
        pipe = bands.GatherPipeline(world, rank, slots=slots, always_collective=True)
        if backend == "gloo":
            pipe = HostStagedGather(torch, dist, world, rank, h, frames)
    frame_no = [0]
    row_bytes = w * 4
    band_bytes = (y1 - y0) * row_bytes

    def step(ev=None):
        k = frame_no[0]
        frame_no[0] += 1
        s = k % slots
        st = streams[s]
        if gather_mode == "torch":
            with torch.cuda.stream(st):
                pipe.begin(k)  # stream-level wait (on st) until the gather that last read band_bufs[s] is done
                if ev is not None:
                    ev.record(st)
                ctx.render_rows_device(m, w, h, y0, y1, args.depth, flags, band_bufs[s].data_ptr(), 0, st.cuda_stream)
                pipe.submit(k, band_bufs[s], gather_lists[s] if backend == "nccl" else s)  # the collective is ordered after st's render of this frame
            return
        render_part(s, st, ev)
        if gather_mode == "rccl":  # stream order: render k -> gather k -> (slots frames later) render k + slots
            gather_part(s, st)

    def render_part(s, st, ev=None):
        # in place on rank 0 of a real multi-rank run; into the band buffer otherwise
        in_place = gather_mode == "rccl" and rank == 0 and rehearse_band <= 1
        d_out = frames[s].data_ptr() + y0 * row_bytes if in_place else band_bufs[s].data_ptr()
        if ev is not None:
            ev.record(st)
        ctx.render_rows_device(m, w, h, y0, y1, args.depth, flags, d_out, 0, st.cuda_stream)

    def gather_part(s, st):
        comm = rccl[s % n_comms]
        if world > 1:
            comm.gather(h, w, st.cuda_stream, band_ptr=0 if rank == 0 else band_bufs[s].data_ptr(), frame_ptr=frames[s].data_ptr() if rank == 0 else 0)
        else:  # one-rank rehearsal: the band travels rank 0 -> rank 0 through ncclSend/ncclRecv into its rows of the frame
            comm.loopback(band_bufs[s].data_ptr(), frames[s].data_ptr() + y0 * row_bytes, band_bytes, st.cuda_stream)

    def fence():
        if pipe is not None:
            pipe.drain()
        torch.cuda.synchronize(dev)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # unconditional priming, not counted as warm-up: the first step loads code objects and sets up the RCCL peer connections, and
    # about 50 ms of steps bring the clocks out of idle (a 20-step run without it measured 0.65 instead of 0.56 ms per frame), so
    # that even --warmup 0 times steady-state steps; then the W warm-up steps the contract asks for.  The number of priming steps
    # is fixed (not wall-clock driven) so that every rank issues the same collectives.
    step()
    fence()
    for _ in range(max(1, int(0.05 / (0.6e-3 / world)) if not general else 8)):
        step()
    fence()
    for _ in range(args.warmup):
        step()
    fence()
    # HIP events on the launch stream over the timed region: ONE per step, recorded in front of the step's launch (and one behind the
    # last), so the duration of launch k is event k -> event k + 1 on that stream.  An event is a packet of its own between two
    # kernels and costs ~4 us of GPU timeline: two per launch plus the two the library used to record cost 3 % of a C3 frame
    # (profiles/r02_tuning.md); this way the measured duration also holds the launch gap (it can only overstate the kernel).
    events = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + slots)]
    phase_ms = [0.0, 0.0, 0.0]
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(events[k])
    for s_ in range(slots):  # the closing event of every stream's last step: "step steps + s_" would run on stream (frame_no + s_) % slots
        events[args.steps + s_].record(streams[(frame_no[0] + s_) % slots])
    t_issued = time.perf_counter()
    fence()
    t1 = time.perf_counter()
    print("rank %d: host issued %d steps in %.3f ms (%.1f us per step), all done after %.3f ms" %
          (rank, args.steps, (t_issued - t0) * 1e3, (t_issued - t0) * 1e6 / max(1, args.steps), (t1 - t0) * 1e3), file=sys.stderr)
    elapsed = t1 - t0
    # timed step k runs on stream (F0 + k) % slots (F0 = frames issued before the timed region); the next event on that stream is the one
    # of step k + slots (or the closing one)
    kern_ms = sum(events[k].elapsed_time(events[k + slots]) for k in range(args.steps)) / max(1, args.steps)

    # General path: per-kernel durations of a launch come from the library's own HIP events between its kernels (rt_last_phase_ms):
    # a few more steps after the timed region, one at a time, averaged.  Also the number of hit records (= rays the general kernels trace).
    n_hits = None
    if general:
        n_ph = max(1, min(args.steps, 10))
        for _ in range(n_ph):
            ctx.render_rows_device(m, w, h, y0, y1, args.depth, flags | capi.RT_TIMED, band_bufs[0].data_ptr(), 0, streams[0].cuda_stream)
            p_ms = ctx.last_phase_ms()
            for i_ in range(3):
                phase_ms[i_] += p_ms[i_] / n_ph
        hit = torch.empty(((y1 - y0), w), dtype=torch.int32, device=dev)
        ctx.render_rows_device(m, w, h, y0, y1, 1, 0, band_bufs[0].data_ptr(), hit.data_ptr(), streams[0].cuda_stream)
        streams[0].synchronize()
        n_hits = int((hit >= 0).sum())
        del hit

    # After the timed region, multi-rank runs only: the two halves of a step on their own (same streams, same rotation), so that the
    # JSON line says which side bounds the pipeline - the renders of the bands or the gather over the links.  Not part of `value`.
    diag = None
    if gather_mode == "rccl":
        n_diag = 120
        fence()
        t_r0 = time.perf_counter()
        for k in range(n_diag):
            render_part(k % slots, streams[k % slots])
        fence()
        t_r1 = time.perf_counter()
        for k in range(n_diag):
            gather_part(k % slots, streams[k % slots])
        fence()
        t_g1 = time.perf_counter()
        d = torch.tensor([t_r1 - t_r0, t_g1 - t_r1], dtype=torch.float64, device=cdev)
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        peer_bytes = bands.max_band_rows(h, world) * row_bytes if world > 1 else band_bytes
        diag = {"render_only_us_per_step": float(d[0]) / n_diag * 1e6, "gather_only_us_per_step": float(d[1]) / n_diag * 1e6,
                "gather_only_GBps_per_peer_link": peer_bytes / (float(d[1]) / n_diag) / 1e9,
                "note": "%d steps each, after the timed region, same streams; the fence (device sync + barrier) is inside both figures" % n_diag}

    tmax = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=cdev)
    if use_dist:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    elapsed_max, kern_ms_max = float(tmax[0]), float(tmax[1])

    frame_check = None
    if backend == "gloo" and use_dist and rank == 0 and gather_mode == "torch":
        whole = torch.zeros((h, w, 4), dtype=torch.uint8, device=dev)
        ctx.render_rows_device(m, w, h, 0, h, args.depth, flags, whole.data_ptr(), 0, streams[0].cuda_stream)
        streams[0].synchronize()
        frame_check = all(bool(torch.equal(f_, whole)) for f_ in frames)
        if not frame_check:
            raise SystemExit("rehearsal: the frame assembled from the ranks' bands differs from the single launch")
    if rank == 0:
        rays_per_step = w * h if rehearse_band <= 1 else (y1 - y0) * w
        value = rays_per_step * args.steps / elapsed_max
        step_s = elapsed_max / args.steps
        mode = "depth %d%s" % (args.depth, ", RT_SHADOWS (extension)" if args.shadows else "")
        n_point = int((lights.kind == 0).sum()) if lights.kind is not None else lights.n
        # roofline of the dominant kernel (this rank's launch): executed flops of its filter loop / its own duration
        my_tests = (y1 - y0) * w * objs.n
        kernels = [("render_kernel<primary>" if not general else "render_kernel<emit>", my_tests, FLOPS_PER_TEST, kern_ms if not general else phase_ms[0],
                    "%d per (pixel,sphere) filter test x %d tests per launch" % (FLOPS_PER_TEST, my_tests))]
        cull_info = None
        if args.cull:
            # render_cull_kernel: the cone hierarchy tests only the spheres a tile's rays can reach, so the work is data-dependent: executed
            # flops = the counter-derived figure of the committed PMC pass of this command (None without one), beside the whole-list bound
            cname = "render_cull_kernel<primary>" if not general else "render_cull_kernel<emit>"
            pv, pnote = pmc_summary(args.config, args.depth, args.shadows, cname, cull=True)
            cu_flops = pv.get("flops") if pv else None
            bound = FLOPS_PER_TEST * my_tests
            cull_info = {"whole_list_bound_flops": bound, "executed_flops": cu_flops, "executed_over_bound": (cu_flops / bound) if cu_flops else None,
                         "lane_utilisation": pv.get("lane_utilisation") if pv else None, "source": ("counter-derived flops " + pnote) if cu_flops else pnote}
            kernels = [(cname, 1, cu_flops or 0.0, kern_ms if not general else phase_ms[0],
                        ("counter-derived: %.4g flops per launch, %.4f of the whole-list bound (%d per (pixel,sphere) filter test x %d tests)" % (cu_flops, cu_flops / bound, FLOPS_PER_TEST, my_tests))
                        if cu_flops else "executed flops unknown (%s); whole-list bound %.4g" % (pnote, bound))]
        shadow_info = None
        if general and args.shadows:
            # shadow_mask_kernel: its any-hit hierarchy tests only the spheres a wave's rays can reach, so the work is data-dependent:
            # the executed flops are the counter-derived figure of the committed PMC pass of this command (None without one); the
            # no-early-out bound (every shadow ray against every sphere with the 3-FMA filter) is reported beside it
            bound = FLOPS_PER_SHADOW_TEST * n_hits * n_point * objs.n * args.depth
            pv, pnote = pmc_summary(args.config, args.depth, True, "shadow_mask_kernel", cull=args.cull)
            sh_flops = pv.get("flops") if pv else None
            if sh_flops:
                sh_flops *= args.depth  # the summary averages the launches of one frame: one per iteration of renderer.rs:46
            shadow_info = {"no_early_out_bound_flops": bound, "executed_flops": sh_flops,
                           "executed_over_bound": (sh_flops / bound) if sh_flops else None,
                           "lane_utilisation": pv.get("lane_utilisation") if pv else None,
                           "source": ("counter-derived flops " + pnote) if sh_flops else pnote}
            kernels.append(("shadow_mask_kernel (+ record sort)", 1, sh_flops or 0.0, phase_ms[1],
                            ("counter-derived: 64 x (2 FMA + ADD + MUL + TRANS) of the committed PMC pass = %.4g flops per frame, %.3f of the no-early-out bound "
                             "(%d per (shadow ray,sphere) filter test x %d hit records x %d point lights x %d spheres x %d iteration(s))" %
                             (sh_flops, sh_flops / bound, FLOPS_PER_SHADOW_TEST, n_hits, n_point, objs.n, args.depth)) if sh_flops else
                            "executed flops unknown (%s); no-early-out bound %.4g" % (pnote, bound)))
        bounce_info = None
        if general and args.depth > 1:
            # shade_iter_kernel: shading + the bounce trace, whose hierarchy (trace_bundle) tests only the spheres a wave's bundle of rays
            # can reach: executed flops = the counter-derived figure of the committed PMC pass of this command (None without one), beside
            # the bound of walking the whole list with the 7-FMA filter for every bounce ray
            bound = FLOPS_PER_BOUNCE_TEST * n_hits * objs.n * (args.depth - 1)
            pv, pnote = pmc_summary(args.config, args.depth, args.shadows, "shade_iter_kernel", cull=args.cull)
            bo_flops = pv.get("flops") if pv else None
            if bo_flops:
                bo_flops *= args.depth  # the summary averages the launches of one frame: one per iteration of renderer.rs:46
            bounce_info = {"whole_list_bound_flops": bound, "executed_flops": bo_flops, "executed_over_bound": (bo_flops / bound) if bo_flops else None,
                           "lane_utilisation": pv.get("lane_utilisation") if pv else None,
                           "source": ("counter-derived flops " + pnote + ", x %d launches per frame" % args.depth) if bo_flops else pnote}
            kernels.append(("shade_iter_kernel (shading + trace_bundle)", 1, bo_flops or 0.0, phase_ms[2],
                            ("counter-derived: %.4g flops per frame, %.3f of the whole-list bound (%d per (bounce ray,sphere) filter test x %d hit records x %d spheres x %d bounce(s))" %
                             (bo_flops, bo_flops / bound, FLOPS_PER_BOUNCE_TEST, n_hits, objs.n, args.depth - 1)) if bo_flops else
                            "executed flops unknown (%s); whole-list bound %.4g" % (pnote, bound)))
        kernels.sort(key=lambda k_: -k_[3])
        dom = kernels[0]
        wall_based = slots > 1
        if wall_based:
            # several frames in flight: per-launch event durations overlap, so the fraction comes from the wall clock of the step (this
            # rank's flops / step time) and cannot exceed what the device did in that time
            dom_ms = step_s * 1e3
            achieved = sum(k_[1] * k_[2] for k_ in kernels) / step_s / 1e12
        else:
            dom_ms = dom[3]
            achieved = dom[1] * dom[2] / (dom_ms * 1e-3) / 1e12
        metric = {"C3": "primary rays/sec at 4K, 1024 spheres (config C3)",
                  "C4": "primary rays/sec at 8K, 8192 spheres (config C4)",
                  "C5": "primary rays/sec at 4K, 4096 spheres, 32 lights (config C5)"}[args.config]
        if args.cull:
            mode += ", RT_CULL (opt-in: primary rays through the cone hierarchy)"
        if general or args.cull:
            metric = metric.replace("primary rays/sec", "primary rays/sec [%s]" % mode)
        out = {
            "metric": metric,
            "value": value,
            "unit": "rays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": step_s * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic (SCENE-GEN v1, PCG32 seed 42)",
            "config": {"workload": "%s: %dx%d frame, %d spheres, %d point lights, %s, RGBA8 out" % (args.config, w, h, objs.n, lights.n, mode),
                       "parallelism": ("row bands x%d, RCCL gather to rank 0 (%s), %d frames in flight" % (world, ("grouped ncclSend/ncclRecv, %d communicators" % n_comms) if gather_mode == "rccl" else "torch.distributed.gather", slots)) if use_dist else "one rank, whole frame, one stream", "kernel": dom[0]},
            "roofline": {"bound": "valu_fp32", "achieved": achieved, "peak": PEAK_FP32_VALU_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP32_VALU_TFLOPS, "traffic": None,
                         "kernel": dom[0] if not wall_based else "all kernels of this rank's step",
                         "kernel_ms": dom_ms, "kernel_ms_max_over_ranks": kern_ms_max if not general else None,
                         "duration_source": ("wall clock of the step (max over ranks): %d frames in flight on %d streams, per-launch event durations would overlap" % (slots, slots))
                         if wall_based else ("HIP events on the launch stream, one in front of every timed launch (duration = event k -> event k + 1: kernel + launch gap)" if not general else
                                             "HIP events the library records between its kernels (rt_last_phase_ms), %d launches after the timed region" % max(1, min(args.steps, 10))),
                         "flops_counted": dom[4] if not wall_based else "; ".join(k_[4] for k_ in kernels)},
        }
        if general:
            out["general_path"] = {"hit_records_per_launch": n_hits, "launch_ms_by_events": kern_ms,
                                   "kernels": [{"kernel": k_[0], "ms": k_[3], "TFLOP/s": (k_[1] * k_[2] / (k_[3] * 1e-3) / 1e12) if k_[3] > 0 else None,
                                                "frac": (k_[1] * k_[2] / (k_[3] * 1e-3) / 1e12 / PEAK_FP32_VALU_TFLOPS) if k_[3] > 0 else None,
                                                "flops_counted": k_[4]} for k_ in kernels],
                                   "shadow_any_hit": shadow_info, "bounce_closest_hit": bounce_info,
                                   "rays_per_step": w * h + (n_hits * n_point * args.depth if args.shadows else 0) + n_hits * (args.depth - 1),
                                   "note": "value counts primary rays only; rays_per_step adds the shadow and bounce rays (upper bounds: later iterations hold fewer live records)"}
        out["roofline"]["traffic"], out["roofline"]["traffic_note"] = pmc_traffic(args.config, args.depth, args.shadows, dom[0], cull=args.cull)
        if args.cull:
            out["cull"] = cull_info
            if not general and not wall_based:
                # no FLOP roofline applies to the culled walk (it executes ~1 % of the whole-list flops): the line reports the frame write
                # against HBM — the only algorithmic bytes of the launch — and the executed flops beside it; what bounds the kernel is
                # VALU issue of the per-pixel ray setup in the reference's exact arithmetic (IEEE divisions and square roots)
                gbps = (y1 - y0) * w * 4 / (dom_ms * 1e-3) / 1e9
                out["roofline"].update({"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s", "frac": gbps / PEAK_HBM_GBPS,
                                        "flops_counted": "algorithmic bytes = the RGBA8 frame write, %d B per launch; %s" % ((y1 - y0) * w * 4, dom[4]),
                                        "valu_TFLOP/s_executed": (cu_flops / (dom_ms * 1e-3) / 1e12) if cu_flops else None,
                                        "note": "neither roofline binds: VALU issue of the exact per-pixel ray setup does (DESIGN.md 3.5)"})
        if diag is not None:
            out["diagnostics"] = diag
        if backend == "gloo" and use_dist:
            out["rehearsal"] = ("BENCH_BACKEND=gloo: %d ranks sharing %d GPU(s), bands gathered through host memory; control-flow rehearsal, not a result; "
                                "assembled frames equal the single launch: %s" % (world, torch.cuda.device_count(), frame_check))
        if rehearse_band > 1:
            out["rehearsal"] = "one rank rendering the band of 1/%d of the frame: host issue-rate probe, not a result" % rehearse_band
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(objs, lights, m, w, h, args.depth, 1 if args.shadows else 0)
        os.write(json_fd, (json.dumps(out) + "\n").encode())

    ctx.close()
    for comm in rccl or []:
        comm.close()
    if use_dist:
        dist.destroy_process_group()


class HostStagedGather:
    """BENCH_BACKEND=gloo only: the interface of bands.GatherPipeline with the bands staged through host memory (gloo has no device
    gather).  submit(k, band, slot): wait for the render, copy the band to the host, gather, rank 0 copies every band into its rows of
    frames[slot].  Synchronous: a rehearsal of the control flow, not of the overlap."""

    def __init__(self, torch, dist, world, rank, h, frames):
        self.torch, self.dist, self.world, self.rank, self.h, self.frames = torch, dist, world, rank, h, frames

    def begin(self, k):
        return 0

    def submit(self, k, band, slot):
        from raytracing_amd import bands as _b
        self.torch.cuda.current_stream().synchronize()
        host = band.cpu()
        parts = [self.torch.empty_like(host) for _ in range(self.world)] if self.rank == 0 else None
        self.dist.gather(host, gather_list=parts, dst=0)
        if self.rank == 0:
            for r in range(self.world):
                a, b = _b.band_rows(self.h, self.world, r)
                self.frames[slot][a:b].copy_(parts[r][: b - a])

    def drain(self):
        pass


def pick_concurrent_streams(torch, dev, n, candidates=16, pool=None):
    """n streams that demonstrably run concurrently.  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 here) and
    two streams that share one execute in order: a kernel trace of the band pipeline showed two of its three render streams on the same
    queue, i.e. frames serialised.  Probe: a ~0.3 ms spin kernel on stream A, then an event on candidate B — if B's event completes while
    A's spin is still running, A and B sit on different queues."""
    pool = pool or [torch.cuda.Stream(dev) for _ in range(candidates)]
    torch.cuda.synchronize(dev)

    def independent(a, b):
        done_a, done_b = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(a):
            torch.cuda._sleep(600000)
            done_a.record(a)
        done_b.record(b)  # nothing precedes it on b: completes at once unless b executes behind a
        done_b.synchronize()
        ok = not done_a.query()
        done_a.synchronize()
        return ok

    chosen = [pool[0]]
    for c in pool[1:]:
        if len(chosen) == n:
            break
        if all(independent(a, c) and independent(c, a) for a in chosen):
            chosen.append(c)
    if len(chosen) < n:
        print("only %d mutually concurrent streams found among %d candidates" % (len(chosen), candidates), file=sys.stderr)
    return chosen[:n]


def kernel_source_sha256():
    """Identity of the kernels a PMC summary was taken with: SHA-256 over render_kernels.hip + render_kernels.h."""
    import hashlib
    hsh = hashlib.sha256()
    for name in ("render_kernels.hip", "render_kernels.h"):
        with open(os.path.join(ROOT, "raytracing_amd", "csrc", name), "rb") as f:
            hsh.update(f.read())
    return hsh.hexdigest()


def pmc_summary(config, depth=1, shadows=False, kernel="render_kernel<primary>", cull=False):
    """Figures of the committed rocprofv3 PMC passes of THIS command for one kernel (profiles/r03_<tag>_pmc.txt, written by
    tools/summarize_prof.py): HBM bytes per launch (TCC_EA0_WRREQ x 64 B + TCC_EA0_RDREQ x 64 B x 2, the gfx950 correction of
    MI355X_MICROARCH.md) and the counter-derived f32 flops per launch.  Counters cannot be read inside an unprofiled run, so these are
    the values of the profiled run of the same command — and only if the summary was taken with the kernel sources this build comes
    from (its `kernel source sha256` line).  Returns (dict or None, note)."""
    tag = config.lower() + ("_depth%d" % depth if depth > 1 else "") + ("_shadows" if shadows else "") + ("_cull" if cull else "")
    path = os.path.join(ROOT, "profiles", "r03_%s_pmc.txt" % tag)
    if not os.path.exists(path):
        return None, "no PMC pass committed for this command (%s)" % os.path.basename(path)
    want = {"render_kernel<primary>": "render_kernel<16, false>", "render_kernel<emit>": "render_kernel<16, true>",
            "render_cull_kernel<primary>": "render_cull_kernel<false", "render_cull_kernel<emit>": "render_cull_kernel<true"}.get(kernel, kernel.split(" ")[0])
    sha = None
    vals = {}
    inside = False
    for line in open(path):
        if line.startswith("kernel source sha256"):
            sha = line.rsplit("=", 1)[1].strip()
        elif line.startswith("## "):
            inside = want in line
        elif inside and line.startswith("HBM write bytes"):
            vals["wr"] = float(line.rsplit("=", 1)[1])
        elif inside and line.startswith("HBM read bytes"):
            vals["rd"] = float(line.rsplit("=", 1)[1])
        elif inside and line.startswith("counter-derived f32 flops per dispatch"):
            vals["flops"] = float(line.rsplit("=", 1)[1])
        elif inside and line.startswith("VALU lane utilisation"):
            vals["lane_utilisation"] = float(line.rsplit("=", 1)[1])
    if sha != kernel_source_sha256():
        return None, "%s was taken with other kernel sources (sha256 %s...): stale, not reported" % (os.path.basename(path), (sha or "none")[:12])
    if "wr" not in vals or "rd" not in vals:
        return None, "%s has no HBM lines for %s" % (os.path.basename(path), want)
    return vals, "per launch of %s, from %s (separate --pmc passes of this command, same kernel sources)" % (want, os.path.relpath(path, ROOT))


def pmc_traffic(config, depth=1, shadows=False, kernel="render_kernel<primary>", cull=False):
    vals, note = pmc_summary(config, depth, shadows, kernel, cull)
    return (vals["wr"] + vals["rd"], "bytes " + note) if vals else (None, note)


def cpu_baseline(objs, lights, m, w, h, depth=1, flags=0):
    """The CPU oracle (C restatement of the reference, row-parallel like rayon) timed on this host on a bounded sample."""
    from tests import oracle_py
    orc = oracle_py.load()
    threads = min(orc.default_threads(), 16)
    row_step = 2 if objs.n <= 1024 else (8 if objs.n <= 4096 else 32)  # about 10-30 s of CPU work whatever the config
    if flags:
        row_step *= 4
    t0 = time.perf_counter()
    orc.render(objs, lights, m, w, h, row_step=row_step, nthreads=threads, depth=depth, flags=flags)
    dt = time.perf_counter() - t0
    rows = (h + row_step - 1) // row_step
    # the same oracle on ONE thread, on a 16 x sparser sample (SURVEY 8d asks for both figures)
    row_step1 = 16 * row_step
    t0 = time.perf_counter()
    orc.render(objs, lights, m, w, h, row_step=row_step1, nthreads=1, depth=depth, flags=flags)
    dt1 = time.perf_counter() - t0
    rows1 = (h + row_step1 - 1) // row_step1
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": rows * w / dt, "unit": "rays/s", "cores": threads, "kind": "port",
            "sample": "every %dth row of the same frame (%d rows x %d px), %d threads, %.2f s wall" % (row_step, rows, w, threads, dt),
            "value_one_thread": rows1 * w / dt1,
            "sample_one_thread": "every %dth row (%d rows x %d px), 1 thread, %.2f s wall" % (row_step1, rows1, w, dt1),
            "host": "%s, %d logical CPUs visible" % (model, os.cpu_count() or 0),
            "note": "C restatement of the reference (oracle/rt_oracle.c), not the Rust binary: no Rust toolchain in the image"}


if __name__ == "__main__":
    main()
