"""GPU: rt_render_rows_device_noalloc (scratch in the library's fixed per-GPU array, nothing allocated in the call) against
rt_render_rows_device on the same inputs — byte-exact, two calls in a row, the second with a smaller input; general-ray launches
(depth > 1, RT_SHADOWS: the hit records live in the array) and primary-only ones (only the tile-queue counters do)."""
import numpy as np
import pytest

from raytracing_amd import capi, scenegen

pytestmark = pytest.mark.gpu


def _pair(ctx, torch, m, w, h, y0, y1, depth, flags, st):
    """(existing entry point, allocation-free entry point) images + hit-index planes of the same band."""
    ch = 3 if flags & capi.RT_OUT_RGB8 else 4
    outs = []
    for fn in (ctx.render_rows_device, ctx.render_rows_device_noalloc):
        img = torch.full((y1 - y0, w, ch), 7, dtype=torch.uint8, device="cuda")
        hit = torch.full((y1 - y0, w), -9, dtype=torch.int32, device="cuda")
        fn(m, w, h, y0, y1, depth, flags, img.data_ptr(), hit.data_ptr(), st.cuda_stream)
        st.synchronize()
        outs.append((img.cpu().numpy(), hit.cpu().numpy()))
    return outs


def test_noalloc_matches_allocating_entry_point(oracle):
    import torch
    w, h = 640, 360
    objs, lights = scenegen.generate(w, h, 1500, 5, seed=123)  # two LDS chunks of spheres
    m = oracle.default_camera(w, h)
    st = torch.cuda.Stream()
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        # two calls in a row per mode, the second with a smaller input (stale records of the first call must not leak into it)
        for depth, flags in ((2, capi.RT_SHADOWS), (1, capi.RT_SHADOWS | capi.RT_OUT_RGB8), (3, 0), (1, 0)):
            for (y0, y1) in ((0, h), (101, 178)):
                (a, ah), (b, bh) = _pair(ctx, torch, m, w, h, y0, y1, depth, flags, st)
                assert np.array_equal(a, b), f"depth {depth} flags {flags} rows {y0}:{y1}: images differ"
                assert np.array_equal(ah, bh), f"depth {depth} flags {flags} rows {y0}:{y1}: hit planes differ"
                assert (ah >= 0).any()
        # and against the oracle once (the existing entry point is itself checked against it elsewhere)
        (_, _), (b, bh) = _pair(ctx, torch, m, w, h, 101, 178, 2, capi.RT_SHADOWS, st)
        ref, ref_hit = oracle.render(objs, lights, m, w, h, y0=101, y1=178, depth=2, flags=1, want_hit_idx=True)
        assert np.array_equal(b, ref) and np.array_equal(bh, ref_hit)
        # a smaller frame after the larger ones, another scene on the same context
        objs2, lights2 = scenegen.generate(97, 61, 40, 2, seed=5)
        ctx.scene_upload(objs2, lights2)
        m2 = oracle.default_camera(97, 61)
        (a, ah), (b, bh) = _pair(ctx, torch, m2, 97, 61, 0, 61, 2, capi.RT_SHADOWS, st)
        assert np.array_equal(a, b) and np.array_equal(ah, bh)
        assert np.array_equal(b, oracle.render(objs2, lights2, m2, 97, 61, depth=2, flags=1))


def test_noalloc_limits_and_whole_4k_primary(oracle):
    import torch
    w, h, objs, lights = scenegen.config("C3")
    m = oracle.default_camera(w, h)
    st = torch.cuda.Stream()
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        img = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        # general launch beyond the fixed array: refused, nothing enqueued
        with pytest.raises(capi.RtError) as e:
            ctx.render_rows_device_noalloc(m, w, h, 0, h, 2, 0, img.data_ptr(), 0, st.cuda_stream)
        assert e.value.code == capi.RT_ERR_TOO_LARGE
        # 16 planes (depth 2: 3 + 9 of ray state + 4 of the record sort) x 4 B per pixel + the sort bins: 136 rows of 3840 pixels fit, 137 do not
        limit_rows = (capi.RT_NOALLOC_SCRATCH_BYTES - 272 - 4 * (objs.n + 4)) // (16 * 4 * w)
        with pytest.raises(capi.RtError):
            ctx.render_rows_device_noalloc(m, w, h, 0, limit_rows + 1, 2, 0, img.data_ptr(), 0, st.cuda_stream)
        ctx.render_rows_device_noalloc(m, w, h, 0, limit_rows, 2, 0, img.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        ref = torch.zeros((limit_rows, w, 4), dtype=torch.uint8, device="cuda")
        ctx.render_rows_device(m, w, h, 0, limit_rows, 2, 0, ref.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        assert torch.equal(img[:limit_rows], ref)
        # primary-only: any size (the whole C3 frame), twice, then a band
        full = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        ctx.render_rows_device(m, w, h, 0, h, 1, 0, full.data_ptr(), 0, st.cuda_stream)
        for _ in range(2):
            img.zero_()
            ctx.render_rows_device_noalloc(m, w, h, 0, h, 1, 0, img.data_ptr(), 0, st.cuda_stream)
            st.synchronize()
            assert torch.equal(img, full)
        band = torch.zeros((270, w, 4), dtype=torch.uint8, device="cuda")
        ctx.render_rows_device_noalloc(m, w, h, 945, 1215, 1, 0, band.data_ptr(), 0, st.cuda_stream)
        st.synchronize()
        assert torch.equal(band, full[945:1215])


def test_noalloc_under_hip_graph_capture(oracle):
    """The callers the entry point is for: captured into a HIP graph (torch.cuda.CUDAGraph) and replayed.  Every replay re-runs the
    scratch clears (they are kernels: hipMemsetAsync nodes on this scratch did not replay faithfully — stale occlusion bits from the
    second replay on), so each replay into a cleared buffer gives the bytes of the allocating entry point."""
    import torch
    w, h = 1280, 720
    objs, lights = scenegen.generate(w, h, 400, 4, seed=3)
    m = oracle.default_camera(w, h)
    st = torch.cuda.Stream()
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        for depth, flags, rows in ((1, 0, h), (1, capi.RT_SHADOWS, 300), (2, 0, 300), (2, capi.RT_SHADOWS, 300), (1, 0, 64)):
            want = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
            ctx.render_rows_device(m, w, h, 0, rows, depth, flags, want.data_ptr(), 0, st.cuda_stream)
            st.synchronize()
            out = torch.zeros((rows, w, 4), dtype=torch.uint8, device="cuda")
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=st):
                ctx.render_rows_device_noalloc(m, w, h, 0, rows, depth, flags, out.data_ptr(), 0, st.cuda_stream)
            for rep in range(3):
                out.zero_()
                torch.cuda.synchronize()
                g.replay()
                torch.cuda.synchronize()
                assert torch.equal(out, want), f"depth {depth} flags {flags} rows {rows}: replay {rep} differs"
            if (depth, flags) == (2, capi.RT_SHADOWS):  # and once against the oracle
                assert np.array_equal(out.cpu().numpy(), oracle.render(objs, lights, m, w, h, y0=0, y1=rows, depth=2, flags=1))
            del g
