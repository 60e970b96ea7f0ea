"""GPU: the HIP path (through the C ABI) against the oracle / golden fixtures. Bar: BIT-EXACT bytes (the north-star
tolerance is +-1 LSB on >= 99.9 % of pixels; the filter+exact-test design makes the image identical, so we assert 0)."""
import os

import numpy as np
import pytest

from raytracing_amd import capi, scene, scenegen

pytestmark = pytest.mark.gpu

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _default_flat():
    elements, lights = scene.default_scene()
    return scene.flatten_elements(elements), scene.flatten_lights(lights)


def _cam(oracle, w, h):
    return oracle.default_camera(w, h)


def _assert_same(gpu, ref, what=""):
    if not np.array_equal(gpu, ref):
        diff = np.abs(gpu.astype(np.int32) - ref.astype(np.int32))
        bad = np.argwhere(diff.max(axis=-1) > 0)
        raise AssertionError(f"{what}: {len(bad)} pixels differ, max |d| = {diff.max()}, first at {bad[:5].tolist()}")


def test_c1_default_scene_matches_result_png(gpu_ctx, oracle):
    """Config 1: main.rs scene 100x100 vs the reference's own result.png."""
    gold = np.load(os.path.join(GOLDEN, "result_100x100.npy"))
    objs, lights = _default_flat()
    gpu_ctx.scene_upload(objs, lights)
    m = capi.camera_transform(*capi.camera_default(), np.float32(1.0))
    rgba = gpu_ctx.render(m, 100, 100, 1)
    assert rgba.shape == (100, 100, 4) and (rgba[..., 3] == 255).all()
    _assert_same(rgba[..., :3], gold, "C1 RGBA")
    rgb = gpu_ctx.render(m, 100, 100, 1, capi.RT_OUT_RGB8)
    _assert_same(rgb, gold, "C1 RGB8")


def test_c1_through_renderer_mirror(tmp_path, oracle):
    """Same through the Renderer/Framebuffer mirror of renderer.rs / display.rs (writes result.png, clears the frame)."""
    gold = np.load(os.path.join(GOLDEN, "result_100x100.npy"))
    elements, lights = scene.default_scene()
    r = scene.Renderer(scene.Framebuffer(100, 100, scene.Camera()), elements, lights)
    out = tmp_path / "result.png"
    r.render(1, display_path=str(out))
    _assert_same(r.last_frame, gold, "Renderer.render")
    assert not r.frame.pixels.any()  # renderer.rs:80 clear()
    from PIL import Image
    _assert_same(np.array(Image.open(out)), gold, "result.png written by display()")
    with pytest.raises(ValueError):
        r.render(0)


def test_c2_default_scene_1080p(gpu_ctx, oracle):
    """Config 2: same scene at 1920x1080 (non-square aspect), every pixel vs the oracle."""
    objs, lights = _default_flat()
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, 1920, 1080)
    gpu, hit = gpu_ctx.render_rows(m, 1920, 1080, 0, 1080, 1, 0, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, 1920, 1080, want_hit_idx=True)
    _assert_same(gpu, ref, "C2")
    assert np.array_equal(hit, ref_hit)


@pytest.mark.parametrize("w,h,n,l,seed", [(257, 131, 37, 3, 1), (64, 64, 1, 1, 2), (333, 77, 300, 5, 3), (16, 4, 16, 2, 4), (1, 1, 5, 1, 5),
                                          (640, 360, 1500, 4, 6)])
def test_random_scenes_ragged_sizes(gpu_ctx, oracle, w, h, n, l, seed):
    objs, lights = scenegen.generate(w, h, n, l, seed=seed)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, w, h, want_hit_idx=True)
    _assert_same(gpu, ref, f"random {w}x{h} n={n}")
    assert np.array_equal(hit, ref_hit)


def test_empty_scene_and_no_lights(gpu_ctx, oracle):
    z = np.zeros(0, np.float32)
    empty_o = capi.FlatObjects(z, z, z, z, z, z, z)
    empty_l = capi.FlatLights(z, z, z, z, z, z, z)
    gpu_ctx.scene_upload(empty_o, empty_l)
    m = _cam(oracle, 50, 30)
    assert not gpu_ctx.render(m, 50, 30)[..., :3].any()
    objs, _ = _default_flat()
    gpu_ctx.scene_upload(objs, empty_l)
    gpu = gpu_ctx.render(m, 50, 30)
    ref = oracle.render(objs, empty_l, m, 50, 30)
    _assert_same(gpu, ref, "no lights")


def test_general_path_degenerate_scenes(gpu_ctx, oracle):
    """depth > 1 and RT_SHADOWS on scenes that leave parts of the general path with nothing to do: no elements, no lights, ambient lights
    only (nothing casts a shadow ray), planes only (no sphere list, every hit point outside the scene ball), one sphere, every pixel a
    hit (the camera inside a sphere), no pixel a hit.  Images and hit planes byte-equal to the oracle."""
    f = np.float32
    z = np.zeros(0, f)
    w, h = 97, 45
    m = _cam(oracle, w, h)
    one = lambda *v: np.array(v, f)
    point = capi.FlatLights(one(0.5, -3), one(2.0, 1), one(-1.0, -6), one(1, 0.5), one(1, 0.7), one(1, 0.9), one(8, 30))
    ambient = capi.FlatLights(one(0, 0), one(0, 0), one(0, 0), one(0.2, 0.1), one(0.3, 0.1), one(0.1, 0.4), one(0, 0), kind=np.array([1, 1], np.uint8))
    none_l = capi.FlatLights(z, z, z, z, z, z, z)
    none_o = capi.FlatObjects(z, z, z, z, z, z, z)
    single = capi.FlatObjects(one(0.3), one(-0.2), one(-4), one(1.2), one(0.9), one(0.8), one(0.7))
    inside = capi.FlatObjects(one(0, 1.5), one(0, 0), one(0, -3), one(50, 0.7), one(0.9, 0.5), one(0.8, 0.6), one(0.7, 0.9))  # the camera sits in the first
    behind = capi.FlatObjects(one(0, 2), one(0, 1), one(5, 9), one(1, 2), one(1, 1), one(1, 1), one(1, 1))               # nothing in view
    planes = _with_planes(none_o, [((0.0, -1.0, 0.0), (0.0, 1.0, 0.0), (0.8, 0.8, 0.8)), ((0.0, 0.0, -9.0), (0.0, 0.1, 1.0), (0.5, 0.9, 0.5))])
    cases = [("no elements", none_o, point), ("no lights", single, none_l), ("ambient only", single, ambient), ("planes only", planes, point),
             ("one sphere", single, point), ("camera inside a sphere", inside, point), ("nothing in view", behind, point),
             ("nothing at all", none_o, none_l)]
    for name, objs, lights in cases:
        gpu_ctx.scene_upload(objs, lights)
        for depth, flags in ((1, capi.RT_SHADOWS), (2, 0), (3, capi.RT_SHADOWS), (2, capi.RT_SHADOWS | capi.RT_OUT_RGB8)):
            gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
            ref, ref_hit = oracle.render(objs, lights, m, w, h, depth=depth, flags=flags & 1, channels=3 if flags & 2 else 4, want_hit_idx=True)
            _assert_same(gpu, ref, f"{name}: depth {depth} flags {flags}")
            assert np.array_equal(hit, ref_hit), f"{name}: hit plane, depth {depth} flags {flags}"


def test_edge_case_spheres(gpu_ctx, oracle):
    """Camera inside a sphere, on its surface, sphere behind the camera, huge/tiny radii, saturating colours, duplicates (ties)."""
    f = np.float32
    cx = np.array([0.0, 0.0, 0.0, 0.3, 0.3, -0.4, 0.0, 2.0, 0.0], f)
    cy = np.array([0.0, 0.0, 0.0, 0.2, 0.2, 0.1, 0.0, 0.0, -1.0], f)
    cz = np.array([0.0, -1.0, 3.0, -2.0, -2.0, -1.5, -50.0, -3.0, -1.0], f)
    r = np.array([5.0, 1.0, 1.0, 0.4, 0.4, 1e-3, 20.0, 1.0, 1e-6], f)  # [0] contains the camera, [1] touches it, [3]==[4] tie
    rr = np.array([0.2, 1.0, 1.0, 50.0, 0.1, 1.0, 0.5, 1.0, 1.0], f)
    rg = np.array([0.3, 0.5, 1.0, 0.1, 50.0, 1.0, 0.5, 0.0, 1.0], f)
    rb = np.array([0.4, 0.2, 1.0, 0.1, 0.1, 1.0, 0.5, 0.0, 1.0], f)
    lights = capi.FlatLights([0.0, 1.0], [0.0, 2.0], [0.0, -1.0], [1, 0.5], [1, 0.5], [1, 0.5], [0.5, 3.0], kind=np.array([0, 0], np.uint8))
    for keep in ([0, 3, 4, 5, 6, 7, 8], [1, 3, 4, 5, 6, 7], [2, 3, 4, 5, 6, 7, 8], [3, 4], [4, 3, 6]):
        k = np.array(keep)
        objs = capi.FlatObjects(cx[k], cy[k], cz[k], r[k], rr[k], rg[k], rb[k])
        gpu_ctx.scene_upload(objs, lights)
        m = _cam(oracle, 200, 120)
        gpu, hit = gpu_ctx.render_rows(m, 200, 120, 0, 120, 1, 0, want_hit_idx=True)
        ref, ref_hit = oracle.render(objs, lights, m, 200, 120, want_hit_idx=True)
        _assert_same(gpu, ref, f"edge set {keep}")
        assert np.array_equal(hit, ref_hit)


def test_row_bands_are_seamless(gpu_ctx, oracle):
    """Rendering a frame as bands (the multi-GPU split) gives exactly the single-launch image."""
    w, h = 320, 200
    objs, lights = scenegen.generate(w, h, 200, 4, seed=9)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    full = gpu_ctx.render(m, w, h)
    for cuts in ([0, 100, 200], [0, 25, 50, 75, 100, 125, 150, 175, 200], [0, 1, 7, 199, 200], [0, 0, 200]):
        parts = [gpu_ctx.render_rows(m, w, h, a, b)[0] for a, b in zip(cuts[:-1], cuts[1:])]
        _assert_same(np.concatenate(parts), full, f"bands {cuts}")


@pytest.mark.parametrize("depth", [2, 3])
def test_depth_bounces(gpu_ctx, oracle, depth):
    """renderer.rs:67-71 bounce path (reference quirks kept: normal at the old origin, no epsilon offset)."""
    w, h = 192, 108
    for objs, lights in (_default_flat(), scenegen.generate(w, h, 64, 3, seed=11)):
        gpu_ctx.scene_upload(objs, lights)
        m = _cam(oracle, w, h)
        gpu = gpu_ctx.render(m, w, h, depth)
        ref = oracle.render(objs, lights, m, w, h, depth=depth)
        _assert_same(gpu, ref, f"depth {depth}")


def test_shadows_extension(gpu_ctx, oracle):
    """RT_SHADOWS (extension; semantics defined by the oracle)."""
    w, h = 240, 135
    objs, lights = scenegen.generate(w, h, 256, 6, seed=21)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu = gpu_ctx.render(m, w, h, 1, capi.RT_SHADOWS)
    ref = oracle.render(objs, lights, m, w, h, flags=1)
    _assert_same(gpu, ref, "shadows")
    plain = gpu_ctx.render(m, w, h, 1, 0)
    assert not np.array_equal(plain, gpu)  # the extension actually darkens something
    gpu2 = gpu_ctx.render(m, w, h, 2, capi.RT_SHADOWS)
    ref2 = oracle.render(objs, lights, m, w, h, depth=2, flags=1)
    _assert_same(gpu2, ref2, "shadows depth 2")


def _tail_sections_vs_oracle(oracle, objs, lights, m, w, h, gpu, hit, flags, what):
    """The primary kernel cuts a frame into 16-row tiles, then (last 2 %) 8-row tiles, then 4-row tiles (plan_tiles): compare the first
    8 rows of the 8-row section and the whole 4-row section (the last 8 to 16 rows of the frame) with the oracle, so that every tile
    shape is checked at this configuration's list length."""
    plan = capi.tile_plan(w, h, 1024)
    assert plan["tiles2"] > 0 and plan["tiles1"] > 0, "a whole frame of a BASELINE config uses all three tile shapes"
    spans = [(plan["rows4"], plan["rows4"] + 8), (plan["rows4"] + plan["rows2"], h)]
    assert spans[0][1] <= plan["rows4"] + plan["rows2"] and 0 < h - spans[1][0] <= 32
    for y0, y1 in spans:
        ref, ref_hit = oracle.render(objs, lights, m, w, h, y0=y0, y1=y1, flags=flags & 1, want_hit_idx=True)
        _assert_same(gpu[y0:y1], ref, f"{what} rows {y0}:{y1} (tail tile sections)")
        if hit is not None:
            assert np.array_equal(hit[y0:y1], ref_hit)


def test_c3_full_size_sampled_rows(gpu_ctx, oracle):
    """Config 3 at full size (3840x2160, 1024 spheres, 4 lights): every 24th row vs the oracle, plus whole-frame properties."""
    w, h, objs, lights = scenegen.config("C3")
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, w, h, row_step=24, want_hit_idx=True)
    _assert_same(gpu[::24], ref[::24], "C3 sampled rows")
    assert np.array_equal(hit[::24], ref_hit[::24])
    _tail_sections_vs_oracle(oracle, objs, lights, m, w, h, gpu, hit, 0, "C3")
    # size-independent properties: misses are black, hits index a real element, alpha is opaque, bands re-assemble
    assert (gpu[..., 3] == 255).all()
    assert not gpu[hit < 0][:, :3].any()
    assert hit.max() < objs.n and hit.min() >= -1
    cover = (hit >= 0).mean()
    assert 0.05 < cover < 0.9
    band = gpu_ctx.render_rows(m, w, h, 1000, 1100)[0]
    _assert_same(band, gpu[1000:1100], "C3 band")


def test_argument_errors(gpu_ctx, oracle):
    objs, lights = _default_flat()
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, 8, 8)
    with pytest.raises(capi.RtError) as e:
        gpu_ctx.render(m, 8, 8, 0)
    assert e.value.code == capi.RT_ERR_INVALID
    with pytest.raises(capi.RtError):
        gpu_ctx.render_rows(m, 8, 8, 4, 9)
    with pytest.raises(capi.RtError):
        gpu_ctx.render(m, 8, 8, 1, 64)
    bad = capi.FlatObjects([0], [0], [-3], [1], [1], [1], [1], kind=np.array([7], np.uint8))
    with pytest.raises(capi.RtError):
        gpu_ctx.scene_upload(bad, lights)
    fresh = capi.Context(0)
    with pytest.raises(capi.RtError) as e:
        fresh.render(m, 8, 8, 1)
    assert e.value.code == capi.RT_ERR_NO_SCENE
    fresh.close()


def test_cpp_host_mirror_main(tmp_path):
    """examples/main (main.rs written against the C++ mirror of the reference's API) writes the reference's result.png."""
    import subprocess
    exe = os.path.join(os.path.dirname(GOLDEN), "..", "examples", "main")
    assert os.path.exists(exe), "build it first: make -C examples (done by __graft_entry__.build())"
    r = subprocess.run([os.path.abspath(exe)], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    assert "1298 lit pixels" in r.stdout
    from PIL import Image
    got = np.array(Image.open(tmp_path / "result.png"))
    _assert_same(got, np.load(os.path.join(GOLDEN, "result_100x100.npy")), "C++ main result.png")
    bad = subprocess.run([os.path.abspath(exe), "10", "10", "0"], cwd=tmp_path, capture_output=True, text=True, timeout=120)
    assert bad.returncode == 1 and "depth - 1" in bad.stderr


def test_shadows_adversarial_lights(gpu_ctx, oracle):
    """RT_SHADOWS with lights inside spheres, on surfaces, far outside the scene, plus ambient lights in the list; 2 chunks of spheres."""
    w, h = 200, 112
    objs, _ = scenegen.generate(w, h, 1300, 1, seed=33)
    k = 7
    f = np.float32
    px = np.array([0.0, objs.cx[k], objs.cx[3] + objs.radius[3], 300.0, -2.0, 0.0, 1.5], f)
    py = np.array([0.0, objs.cy[k], objs.cy[3], 200.0, 3.0, 0.0, -0.5], f)
    pz = np.array([0.0, objs.cz[k], objs.cz[3], 100.0, -15.0, 0.0, -6.0], f)
    kind = np.array([0, 0, 0, 0, 0, 1, 0], np.uint8)  # [1] sits at a sphere centre, [2] on a sphere surface, [3] far away, [5] ambient
    col = np.array([1.0, 0.8, 0.6, 1.0, 0.9, 0.05, 0.7], f)
    inten = np.array([3.0, 4.0, 2.0, 20000.0, 6.0, 0.0, 5.0], f)
    lights = capi.FlatLights(px, py, pz, col, col[::-1].copy(), col.copy(), inten, kind=kind)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    for depth in (1, 2):
        gpu = gpu_ctx.render(m, w, h, depth, capi.RT_SHADOWS)
        ref = oracle.render(objs, lights, m, w, h, depth=depth, flags=1)
        _assert_same(gpu, ref, f"adversarial shadows depth {depth}")


def test_c5_scene_reduced_resolution(gpu_ctx, oracle):
    """Config 5's scene (4096 spheres, 32 lights; SCENE-GEN for 3840x2160) rendered at 320x180, shadows on and off."""
    _, _, objs, lights = scenegen.config("C5")
    w, h = 320, 180
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    _assert_same(gpu_ctx.render(m, w, h, 1, 0), oracle.render(objs, lights, m, w, h), "C5 scene, no shadows")
    _assert_same(gpu_ctx.render(m, w, h, 1, capi.RT_SHADOWS), oracle.render(objs, lights, m, w, h, flags=1), "C5 scene, shadows")


def test_c4_scene_reduced_resolution(gpu_ctx, oracle):
    """Config 4's scene (8192 spheres = 8 LDS chunks, 8 lights) at 480x270 and as two bands."""
    _, _, objs, lights = scenegen.config("C4")
    w, h = 480, 270
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    ref = oracle.render(objs, lights, m, w, h)
    _assert_same(gpu_ctx.render(m, w, h), ref, "C4 scene")
    parts = [gpu_ctx.render_rows(m, w, h, 0, 135)[0], gpu_ctx.render_rows(m, w, h, 135, 270)[0]]
    _assert_same(np.concatenate(parts), ref, "C4 scene bands")


def test_general_path_rgb8_and_hit_idx(gpu_ctx, oracle):
    """depth 2 + shadows through the general-ray kernels with RGB8 output and the hit-index plane (written by the emit kernel)."""
    w, h = 150, 90
    objs, lights = scenegen.generate(w, h, 120, 3, seed=44)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu, hit = gpu_ctx.render_rows(m, w, h, 10, 77, 2, capi.RT_SHADOWS | capi.RT_OUT_RGB8, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, w, h, y0=10, y1=77, depth=2, flags=1, channels=3, want_hit_idx=True)
    assert gpu.shape == (67, w, 3)
    _assert_same(gpu, ref, "general path RGB8 band")
    assert np.array_equal(hit, ref_hit)
    # the record scratch is reused (and regrown) across calls of different sizes
    big = gpu_ctx.render(m, w, h, 2, capi.RT_SHADOWS)
    _assert_same(big, oracle.render(objs, lights, m, w, h, depth=2, flags=1), "general path full frame after a band")
    small = gpu_ctx.render_rows(m, w, h, 0, 5, 2, capi.RT_SHADOWS)[0]
    _assert_same(small, big[:5], "general path small band after a full frame")


def test_c3_full_size_general_path(gpu_ctx, oracle):
    """Full 4K frame through the general-ray kernels (emit, shadow mask, per-iteration shading): depth 2 with shadows, every 60th row."""
    w, h, objs, lights = scenegen.config("C3")
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu = gpu_ctx.render(m, w, h, 2, capi.RT_SHADOWS)
    ref = oracle.render(objs, lights, m, w, h, row_step=60, depth=2, flags=1)
    _assert_same(gpu[::60], ref[::60], "C3 full frame, depth 2 + shadows")


def test_fuzz_scenes_cameras_modes(gpu_ctx, oracle):
    """60 random scenes: random sphere clouds (some around / behind the camera), mixed point/ambient lights (some inside spheres),
    non-default and non-perspective camera matrices (rotation, shear, row 2 not constant), depth 1-3, shadows on/off, RGB8/RGBA8,
    every third scene with one or two planes (extension) mixed in, every twentieth on a frame of 0.3 - 13 Mpixel.  RT_FUZZ_CASES / RT_FUZZ_SEED widen it for a soak run."""
    rng = np.random.default_rng(int(os.environ.get("RT_FUZZ_SEED", "2024")))
    f = np.float32
    npix = nlit = 0
    bands = int(os.environ.get("RT_FUZZ_BANDS", "0"))  # soak runs: the same cases through a multi-device context with virtual bands
    if bands > 0:
        gpu_ctx = capi.Context(device_ids=[0] * bands)
    for case in range(int(os.environ.get("RT_FUZZ_CASES", "60"))):
        w, h = int(rng.integers(1, 140)), int(rng.integers(1, 90))
        n, nl = int(rng.integers(0, 180)), int(rng.integers(0, 6))
        if case % 10 == 9:  # several LDS chunks of spheres, more than 32 lights now and then
            n, nl = int(rng.integers(1000, 3000)), int(rng.integers(0, 40))
        if case % 20 == 11:  # a large frame now and then: 16-row tiles on a plain grid, the sectioned queue, the chunked host copy
            w, h, n = int(rng.choice([2560, 3840, 5000])), int(rng.integers(120, 2600)), min(n, 48)
        spread = f(rng.choice([3.0, 10.0, 40.0]))
        if case % 7 == 3:  # wild scales: tiny and huge scenes, radii from far below to far above the spacing
            spread = f(rng.choice([1e-3, 0.05, 700.0, 3e4]))
        cx = (rng.standard_normal(n) * spread).astype(f)
        cy = (rng.standard_normal(n) * spread).astype(f)
        cz = (rng.standard_normal(n) * spread - (0 if case % 5 == 0 else spread)).astype(f)
        rad = (rng.random(n) ** 2 * spread * 0.4 + f(1e-3) * min(spread, f(1.0))).astype(f)
        if case % 7 == 3 and n:
            rad[:: 5] *= f(rng.choice([1e-3, 30.0]))
        refl = rng.random((3, n)).astype(f) * f(1.5)
        objs = capi.FlatObjects(cx, cy, cz, rad, refl[0], refl[1], refl[2])
        if case % 3 == 1:
            planes = [((rng.standard_normal(3) * spread).astype(f), rng.standard_normal(3).astype(f) + f(0.01), rng.random(3).astype(f))
                      for _ in range(int(rng.integers(1, 3)))]
            objs = _with_planes(objs, planes)
            n = objs.n
        kind = (rng.random(nl) < 0.25).astype(np.uint8)
        lp = (rng.standard_normal((3, nl)) * spread).astype(f)
        if nl and n and case % 3 == 0:  # put the first light at a sphere centre
            lp[0][0], lp[1][0], lp[2][0] = objs.cx[-1], objs.cy[-1], objs.cz[-1]
        lc = rng.random((3, nl)).astype(f)
        inten = (rng.random(nl) * spread * spread).astype(f)
        lights = capi.FlatLights(lp[0], lp[1], lp[2], lc[0], lc[1], lc[2], inten, kind=kind)
        if case % 2 == 0:
            m = oracle.camera_transform(f(rng.uniform(0.3, 2.5)), f(rng.uniform(0.01, 1.0)), f(rng.uniform(5.0, 2000.0)), f(w) / f(h))
        else:
            m = (rng.standard_normal(16) * 0.7).astype(f)
            m[10] -= f(1.5)  # keep most rays pointing down -z
        depth = int(rng.integers(1, 4))
        flags = (capi.RT_SHADOWS if rng.random() < 0.5 else 0) | (capi.RT_OUT_RGB8 if rng.random() < 0.3 else 0)
        gpu_ctx.scene_upload(objs, lights)
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
        ref, ref_hit = oracle.render(objs, lights, m, w, h, depth=depth, flags=flags & 1, channels=3 if flags & 2 else 4, want_hit_idx=True)
        _assert_same(gpu, ref, f"fuzz case {case}: {w}x{h} n={n} l={nl} depth={depth} flags={flags}")
        assert np.array_equal(hit, ref_hit), f"fuzz case {case}: hit index plane"
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags | capi.RT_CULL, want_hit_idx=True)  # primary rays through the cone hierarchy
        _assert_same(gpu, ref, f"fuzz case {case} with RT_CULL: {w}x{h} n={n} l={nl} depth={depth} flags={flags}")
        assert np.array_equal(hit, ref_hit), f"fuzz case {case} with RT_CULL: hit index plane"
        npix += w * h
        nlit += int((hit >= 0).sum())
    print(f"fuzz: {case + 1} scenes, {npix} pixels compared, {nlit} with a hit")


def test_shadow_hierarchy_fuzz(gpu_ctx, oracle):
    """RT_SHADOWS on scenes built to stress the any-hit hierarchy of shadow_mask_kernel (spatial record sort, wave cones, cluster /
    member / length votes): dense and clustered sphere clouds with thousands of hit records, radii from specks to spheres that fill the
    view, lights just outside / on / inside surfaces (wide cones, always-lists), far lights (thin cones), overlapping spheres, depth 1-2,
    a plane now and then (far hit points).  Every image and hit plane byte-equal to the oracle.  RT_SHADOW_FUZZ_CASES / _SEED widen it."""
    rng = np.random.default_rng(int(os.environ.get("RT_SHADOW_FUZZ_SEED", "31")))
    f = np.float32
    nrec = 0
    for case in range(int(os.environ.get("RT_SHADOW_FUZZ_CASES", "24"))):
        w, h = int(rng.integers(200, 420)), int(rng.integers(120, 260))
        n = int(rng.choice([12, 60, 250, 700, 1300]))
        nl = int(rng.integers(1, 7)) if case % 6 else int(rng.integers(33, 40))
        spread = f(rng.choice([4.0, 12.0]))
        k = int(rng.integers(1, 6))  # clusters of spheres
        cc = rng.standard_normal((k, 3)) * spread * np.array([1.0, 0.6, 0.5]) + np.array([0, 0, -2.2 * spread])
        which = rng.integers(0, k, n)
        c = (cc[which] + rng.standard_normal((n, 3)) * spread * rng.choice([0.1, 0.35, 0.8])).astype(f)
        rad = (rng.random(n) ** 3 * spread * 0.25 + 1e-3).astype(f)
        if case % 4 == 1:
            rad[0] = f(spread * 1.5)  # one sphere that fills much of the view
        refl = (rng.random((3, n)) * 1.2).astype(f)
        objs = capi.FlatObjects(c[:, 0], c[:, 1], c[:, 2], rad, refl[0], refl[1], refl[2])
        lp = (rng.standard_normal((nl, 3)) * spread * 1.5 + np.array([0, 0, -1.5 * spread])).astype(f)
        for i in range(nl):  # most lights relative to a sphere: just outside, on, inside its surface; some far away
            mode = rng.integers(0, 6)
            j = int(rng.integers(0, n))
            d = rng.standard_normal(3); d /= np.linalg.norm(d)
            if mode == 0:
                lp[i] = c[j] + d * rad[j] * f(rng.choice([1.0005, 1.02, 1.3]))
            elif mode == 1:
                lp[i] = c[j] + d * rad[j]
            elif mode == 2:
                lp[i] = c[j] + d * rad[j] * f(rng.random() * 0.9)
            elif mode == 3:
                lp[i] = (d * spread * 60).astype(f)
        kind = (rng.random(nl) < 0.15).astype(np.uint8)
        lc = rng.random((3, nl)).astype(f)
        inten = (rng.random(nl) * spread * spread * 3).astype(f)
        lights = capi.FlatLights(lp[:, 0], lp[:, 1], lp[:, 2], lc[0], lc[1], lc[2], inten, kind=kind)
        if case % 5 == 4:
            objs = _with_planes(objs, [((0.0, float(-spread), 0.0), (0.0, 1.0, 0.05), (0.5, 0.5, 0.5))])
        m = oracle.camera_transform(f(rng.uniform(0.6, 1.4)), f(0.01), f(1000.0), f(w) / f(h))
        depth = 1 + case % 2
        flags = capi.RT_SHADOWS | (capi.RT_OUT_RGB8 if case % 7 == 0 else 0)
        gpu_ctx.scene_upload(objs, lights)
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
        ref, ref_hit = oracle.render(objs, lights, m, w, h, depth=depth, flags=1, channels=3 if flags & 2 else 4, want_hit_idx=True)
        _assert_same(gpu, ref, f"shadow fuzz case {case}: {w}x{h} n={objs.n} l={nl} depth={depth}")
        assert np.array_equal(hit, ref_hit)
        gpu_c, hit_c = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags | capi.RT_CULL, want_hit_idx=True)
        _assert_same(gpu_c, ref, f"shadow fuzz case {case} with RT_CULL")
        assert np.array_equal(hit_c, ref_hit)
        lit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags & ~capi.RT_SHADOWS)[0]
        nrec += int((hit >= 0).sum())
        if depth == 1:
            assert (gpu[..., :3] <= lit[..., :3]).all()  # occlusion only removes light
    assert nrec > 200000
    print(f"shadow fuzz: {case + 1} scenes, {nrec} hit records")


def test_bounce_hierarchy_fuzz(gpu_ctx, oracle):
    """depth 2-4 on scenes built to stress trace_bundle (origin ball + direction cone per wave, cluster / member votes, per-wave list):
    dense and clustered clouds with tens of thousands of bounce rays, overlapping spheres (bounce origins inside other spheres: the
    Less branch), an enclosing sphere (every ray hits from inside, bundles wider than any cone), specks and huge spheres, planes (hit
    points far outside the scene ball), wild scales, general camera matrices.  Byte-equal to the oracle.  RT_BOUNCE_FUZZ_CASES / _SEED widen it."""
    rng = np.random.default_rng(int(os.environ.get("RT_BOUNCE_FUZZ_SEED", "47")))
    f = np.float32
    nrec = 0
    for case in range(int(os.environ.get("RT_BOUNCE_FUZZ_CASES", "24"))):
        w, h = int(rng.integers(160, 400)), int(rng.integers(100, 240))
        n = int(rng.choice([3, 40, 200, 700, 1300, 2100]))
        spread = f(rng.choice([3.0, 10.0, 40.0]) if case % 8 != 7 else rng.choice([1e-2, 2e3]))
        k = int(rng.integers(1, 5))
        cc = rng.standard_normal((k, 3)) * spread * np.array([1.0, 0.6, 0.5]) + np.array([0, 0, -2.0 * spread])
        c = (cc[rng.integers(0, k, n)] + rng.standard_normal((n, 3)) * spread * rng.choice([0.15, 0.4, 0.9])).astype(f)
        rad = (rng.random(n) ** 3 * spread * rng.choice([0.1, 0.3]) + 1e-3 * spread).astype(f)
        if case % 4 == 2:
            c[0] = (0, 0, 0); rad[0] = f(spread * 6)  # the camera and the whole cloud inside one sphere
        if case % 4 == 3:
            rad[: n // 3] *= f(4.0)  # heavy overlap
        refl = (rng.random((3, n)) * 1.1).astype(f)
        objs = capi.FlatObjects(c[:, 0], c[:, 1], c[:, 2], rad, refl[0], refl[1], refl[2])
        if case % 5 == 1:
            objs = _with_planes(objs, [((0.0, float(-spread), 0.0), (0.0, 1.0, 0.03), (0.6, 0.6, 0.6)), ((0.0, 0.0, float(-5 * spread)), (0.1, 0.0, 1.0), (0.4, 0.7, 0.9))])
        nl = int(rng.integers(1, 5))
        lp = (rng.standard_normal((nl, 3)) * spread * 1.5 + np.array([0, 0, -1.0 * spread])).astype(f)
        lc = rng.random((3, nl)).astype(f)
        lights = capi.FlatLights(lp[:, 0], lp[:, 1], lp[:, 2], lc[0], lc[1], lc[2], (rng.random(nl) * spread * spread * 2).astype(f),
                                 kind=(rng.random(nl) < 0.2).astype(np.uint8))
        if case % 3 == 0:
            m = (rng.standard_normal(16) * 0.6).astype(f)
            m[10] -= f(1.5)
        else:
            m = oracle.camera_transform(f(rng.uniform(0.5, 1.6)), f(0.01), f(1000.0), f(w) / f(h))
        depth = 2 + case % 3
        flags = (capi.RT_SHADOWS if case % 2 else 0) | (capi.RT_OUT_RGB8 if case % 9 == 0 else 0)
        gpu_ctx.scene_upload(objs, lights)
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
        ref, ref_hit = oracle.render(objs, lights, m, w, h, depth=depth, flags=flags & 1, channels=3 if flags & 2 else 4, want_hit_idx=True)
        _assert_same(gpu, ref, f"bounce fuzz case {case}: {w}x{h} n={objs.n} depth={depth} flags={flags}")
        assert np.array_equal(hit, ref_hit)
        gpu_c, hit_c = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags | capi.RT_CULL, want_hit_idx=True)
        _assert_same(gpu_c, ref, f"bounce fuzz case {case} with RT_CULL")
        assert np.array_equal(hit_c, ref_hit)
        nrec += int((hit >= 0).sum())
    assert nrec > 150000
    print(f"bounce fuzz: {case + 1} scenes, {nrec} hit records")


def test_cull_default_scene_edges_and_bands(gpu_ctx, oracle):
    """RT_CULL (primary rays through the cone hierarchy, render_cull_kernel) gives the bytes of the reference path: C1 against the
    reference's result.png, C2 on every pixel, ragged frame sizes (tiles cut by the right and bottom edges, frames smaller than a tile),
    row bands that start inside a tile row, RGB8, spheres around / on / behind the camera (always-list), planes, empty scenes."""
    gold = np.load(os.path.join(GOLDEN, "result_100x100.npy"))
    objs, lights = _default_flat()
    gpu_ctx.scene_upload(objs, lights)
    m = capi.camera_transform(*capi.camera_default(), np.float32(1.0))
    _assert_same(gpu_ctx.render(m, 100, 100, 1, capi.RT_CULL)[..., :3], gold, "C1 RT_CULL")
    _assert_same(gpu_ctx.render(m, 100, 100, 1, capi.RT_CULL | capi.RT_OUT_RGB8), gold, "C1 RT_CULL RGB8")
    m = _cam(oracle, 1920, 1080)
    gpu, hit = gpu_ctx.render_rows(m, 1920, 1080, 0, 1080, 1, capi.RT_CULL, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, 1920, 1080, want_hit_idx=True)
    _assert_same(gpu, ref, "C2 RT_CULL")
    assert np.array_equal(hit, ref_hit)
    for (w, h, n, l, seed) in [(257, 131, 37, 3, 1), (64, 64, 1, 1, 2), (333, 77, 300, 5, 3), (16, 4, 16, 2, 4), (1, 1, 5, 1, 5), (640, 360, 1500, 4, 6),
                               (65, 17, 90, 2, 7), (15, 33, 20, 1, 8)]:
        o, li = scenegen.generate(w, h, n, l, seed=seed)
        gpu_ctx.scene_upload(o, li)
        m = _cam(oracle, w, h)
        ref, ref_hit = oracle.render(o, li, m, w, h, want_hit_idx=True)
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, capi.RT_CULL, want_hit_idx=True)
        _assert_same(gpu, ref, f"RT_CULL {w}x{h} n={n}")
        assert np.array_equal(hit, ref_hit)
        if h > 20:  # bands whose first row is not a multiple of the tile height
            for y0, y1 in ((3, h - 5), (h // 2 + 1, h), (7, 8)):
                band, bhit = gpu_ctx.render_rows(m, w, h, y0, y1, 1, capi.RT_CULL | capi.RT_OUT_RGB8, want_hit_idx=True)
                _assert_same(band, ref[y0:y1, :, :3], f"RT_CULL band {y0}:{y1} of {w}x{h}")
                assert np.array_equal(bhit, ref_hit[y0:y1])
    # camera matrices that stress tile_cone: rows that cancel (large x-slope against a large constant), a 3000 : 1 pixel aspect, a
    # tiny and a huge field of view, a scale of 1e-6 and of 1e6 on the whole matrix
    f = np.float32
    w, h = 640, 360
    o, li = scenegen.generate(w, h, 500, 3, seed=21)
    gpu_ctx.scene_upload(o, li)
    base = _cam(oracle, w, h)
    mats = []
    adv = base.copy()
    for r_, k_ in enumerate((80.0, -60.0, 50.0)):
        adv[4 * r_ + 0] += f(k_)
        adv[4 * r_ + 3] -= f(k_) * f(0.31)
    mats.append(adv)
    an = base.copy(); an[0] *= f(3000.0); mats.append(an)
    an = base.copy(); an[5] *= f(3000.0); mats.append(an)
    mats.append(oracle.camera_transform(f(0.02), f(0.01), f(1000.0), f(w) / f(h)))
    mats.append(oracle.camera_transform(f(3.1), f(0.01), f(1000.0), f(w) / f(h)))
    mats.append((base * f(1e-6)).astype(f))
    mats.append((base * f(1e6)).astype(f))
    for k, m in enumerate(mats):
        ref, ref_hit = oracle.render(o, li, m, w, h, want_hit_idx=True)
        gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, capi.RT_CULL, want_hit_idx=True)
        _assert_same(gpu, ref, f"RT_CULL camera matrix {k}")
        assert np.array_equal(hit, ref_hit), f"RT_CULL camera matrix {k}: hit plane"
    # camera inside / on / next to spheres, duplicates, a sphere behind the camera, a huge far one; two planes
    c = np.array([[0, 0, 0, 3.0], [0, 0, -1, 1.0], [0.2, 0.1, -1, 0.99999], [0.5, 0.5, 4, 1.0], [0.1, 0.1, -3, 1e-4], [0, 0, -500, 400.0],
                  [1, 0.5, -4, 0.7], [1, 0.5, -4, 0.7], [-2, -1, -6, 1.1], [3, 2, -5, 6.16]], f)
    one = np.ones(len(c), f)
    for drop_first in (False, True):
        cc = c[1:] if drop_first else c
        o = capi.FlatObjects(cc[:, 0], cc[:, 1], cc[:, 2], cc[:, 3], one[: len(cc)] * f(0.9), one[: len(cc)] * f(0.5), one[: len(cc)] * f(0.7))
        o = _with_planes(o, [((0.0, -2.0, 0.0), (0.0, 1.0, 0.02), (0.5, 0.6, 0.7))])
        li = capi.FlatLights(f([1, -4]), f([2, 1]), f([0, -3]), f([1, 0.3]), f([1, 0.3]), f([1, 0.3]), f([6, 9]), kind=np.array([0, 1], np.uint8))
        gpu_ctx.scene_upload(o, li)
        m = _cam(oracle, 200, 120)
        for depth, sh in ((1, 0), (2, 0), (2, capi.RT_SHADOWS)):
            ref, ref_hit = oracle.render(o, li, m, 200, 120, depth=depth, flags=sh, want_hit_idx=True)
            gpu, hit = gpu_ctx.render_rows(m, 200, 120, 0, 120, depth, sh | capi.RT_CULL, want_hit_idx=True)
            _assert_same(gpu, ref, f"RT_CULL awkward spheres depth={depth} shadows={sh} drop_first={drop_first}")
            assert np.array_equal(hit, ref_hit)
    z = np.zeros(0, f)
    gpu_ctx.scene_upload(capi.FlatObjects(z, z, z, z, z, z, z), capi.FlatLights(z, z, z, z, z, z, z))
    assert not gpu_ctx.render(_cam(oracle, 50, 30), 50, 30, 1, capi.RT_CULL)[..., :3].any()
    assert not gpu_ctx.render(_cam(oracle, 50, 30), 50, 30, 2, capi.RT_CULL | capi.RT_SHADOWS)[..., :3].any()


@pytest.mark.parametrize("cfg", ["C3", "C5", "C4"])
def test_cull_full_size_equals_the_default_path(gpu_ctx, oracle, cfg):
    """RT_CULL at the BASELINE sizes: the WHOLE frame and hit plane byte-equal to the default launch (which the tests above hold to the
    oracle on sampled rows), sampled rows against the oracle directly, and for C3 / C5 also depth 2 and RT_SHADOWS (the hit records the
    culled emit kernel appends feed the same general-ray kernels)."""
    import torch
    w, h, objs, lights = scenegen.config(cfg)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    modes = [(1, 0)] if cfg == "C4" else [(1, 0), (2, 0), (1, capi.RT_SHADOWS)]
    out = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(2)]
    hit = [torch.full((h, w), -7, dtype=torch.int32, device="cuda") for _ in range(2)]
    for depth, sh in modes:
        for k, fl in enumerate((sh, sh | capi.RT_CULL)):
            gpu_ctx.render_rows_device(m, w, h, 0, h, depth, fl, out[k].data_ptr(), hit[k].data_ptr())
        torch.cuda.synchronize()
        assert torch.equal(out[0], out[1]), f"{cfg} depth={depth} shadows={sh}: RT_CULL frame differs in {int((out[0] != out[1]).any(dim=-1).sum())} pixels"
        assert torch.equal(hit[0], hit[1])
        assert int((hit[1] >= 0).sum()) > w * h // 20
    step = {"C3": 90, "C5": 216, "C4": 1080}[cfg]
    rows = list(range(7, h, step))
    ref = oracle.render(objs, lights, m, w, h, row_step=1, y0=0, y1=1)  # row 0
    gpu_ctx.render_rows_device(m, w, h, 0, h, 1, capi.RT_CULL, out[1].data_ptr(), hit[1].data_ptr())
    torch.cuda.synchronize()
    got = out[1].cpu().numpy()
    _assert_same(got[0:1], ref, f"{cfg} RT_CULL row 0")
    for y in rows:
        _assert_same(got[y:y + 1], oracle.render(objs, lights, m, w, h, y0=y, y1=y + 1), f"{cfg} RT_CULL row {y}")


def test_cull_through_bands_and_the_noalloc_entry_point(oracle):
    """RT_CULL through a multi-device context (virtual bands, gathered by RCCL) and through rt_render_rows_device_noalloc: same bytes as
    the single default launch."""
    import torch
    w, h = 1280, 720
    objs, lights = scenegen.generate(w, h, 600, 3, seed=12)
    m = _cam(oracle, w, h)
    with capi.Context(0) as one:
        one.scene_upload(objs, lights)
        ref = one.render(m, w, h, 1)
        ref2 = one.render(m, w, h, 2, capi.RT_SHADOWS)
        out = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        one.render_rows_device_noalloc(m, w, h, 0, h, 1, capi.RT_CULL, out.data_ptr())
        torch.cuda.synchronize()
        _assert_same(out.cpu().numpy(), ref, "RT_CULL noalloc")
        half = torch.zeros((200, w, 4), dtype=torch.uint8, device="cuda")
        one.render_rows_device_noalloc(m, w, h, 100, 300, 2, capi.RT_CULL | capi.RT_SHADOWS, half.data_ptr())
        torch.cuda.synchronize()
        _assert_same(half.cpu().numpy(), ref2[100:300], "RT_CULL noalloc general band")
    with capi.Context(device_ids=[0, 0, 0]) as multi:
        multi.scene_upload(objs, lights)
        _assert_same(multi.render(m, w, h, 1, capi.RT_CULL), ref, "RT_CULL 3 virtual bands")
        _assert_same(multi.render(m, w, h, 2, capi.RT_CULL | capi.RT_SHADOWS), ref2, "RT_CULL 3 virtual bands, general path")


def test_wavefront_many_lights_depth_and_bands(gpu_ctx, oracle):
    """General-ray kernels (shadow_mask_kernel + shade_iter_kernel): more than 32 lights (two occlusion words per record) with ambient
    lights in between, depth 3 with shadows, row bands, and the scratch planes reused across calls with different needs."""
    w, h = 160, 96
    objs, pts = scenegen.generate(w, h, 96, 37, seed=33)
    n = pts.n + 3
    ins = [0, 17, 36]  # ambient lights in front, in the middle, near the end of the list
    kind = np.zeros(n, np.uint8)
    px, py, pz, cr, cg, cb, it = (np.zeros(n, np.float32) for _ in range(7))
    src = 0
    for i in range(n):
        if i in ins:
            kind[i] = 1
            cr[i], cg[i], cb[i] = 0.01, 0.02, 0.03
        else:
            px[i], py[i], pz[i], cr[i], cg[i], cb[i], it[i] = (pts.px[src], pts.py[src], pts.pz[src], pts.cr[src], pts.cg[src], pts.cb[src],
                                                               pts.intensity[src])
            src += 1
    lights = capi.FlatLights(px, py, pz, cr, cg, cb, it, kind=kind)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    for depth, flags in ((1, capi.RT_SHADOWS), (3, 0), (3, capi.RT_SHADOWS), (1, 0), (2, capi.RT_SHADOWS)):
        gpu = gpu_ctx.render(m, w, h, depth, flags)
        ref = oracle.render(objs, lights, m, w, h, depth=depth, flags=1 if flags else 0)
        _assert_same(gpu, ref, f"40 lights depth {depth} flags {flags}")
    ref = oracle.render(objs, lights, m, w, h, depth=3, flags=1)
    parts = [gpu_ctx.render_rows(m, w, h, a, b, 3, capi.RT_SHADOWS)[0] for a, b in ((0, 31), (31, 32), (32, 96))]
    _assert_same(np.concatenate(parts), ref, "bands, depth 3, shadows")


def test_c4_c5_full_size_sampled_rows(gpu_ctx, oracle):
    """Configs 4 and 5 at BASELINE.json's full sizes: C4 (7680x4320, 8192 spheres, 8 lights) every 540th row, C5 (3840x2160,
    4096 spheres, 32 lights) with the shadow extension every 360th row, against the oracle; whole-frame properties on the rest."""
    w, h, objs, lights = scenegen.config("C4")
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
    ref, ref_hit = oracle.render(objs, lights, m, w, h, row_step=540, want_hit_idx=True)
    _assert_same(gpu[::540], ref[::540], "C4 sampled rows")
    assert np.array_equal(hit[::540], ref_hit[::540])
    _tail_sections_vs_oracle(oracle, objs, lights, m, w, h, gpu, hit, 0, "C4")  # the 8-chunk list in the 8- and 4-row tile shapes
    assert (gpu[..., 3] == 255).all() and not gpu[hit < 0][:, :3].any() and hit.max() < objs.n
    del gpu, hit
    w, h, objs, lights = scenegen.config("C5")
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    lit = gpu_ctx.render(m, w, h, 1, 0)
    shadowed = gpu_ctx.render(m, w, h, 1, capi.RT_SHADOWS)
    ref = oracle.render(objs, lights, m, w, h, row_step=360, flags=1)
    _assert_same(shadowed[::360], ref[::360], "C5 shadows sampled rows")
    _tail_sections_vs_oracle(oracle, objs, lights, m, w, h, shadowed, None, capi.RT_SHADOWS, "C5 shadows")
    # occlusion only removes light: no channel gets brighter, background stays black, something does get darker
    assert (shadowed[..., :3] <= lit[..., :3]).all()
    assert (shadowed[..., :3] < lit[..., :3]).any()


def test_host_store_pinned_and_chunked_copy(gpu_ctx, oracle):
    """rt_render_rows into a caller-provided store: a frame large enough for the chunked render/copy pipeline (>= 2 Mpixel), in
    pageable memory and in a page-locked store from rt_host_alloc, RGBA8 and RGB8, with the hit-index plane — same bytes as the
    device-resident path (one launch) and, on sampled rows, as the oracle."""
    import torch
    w, h = 2048, 1100
    objs, lights = scenegen.generate(w, h, 300, 3, seed=77)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    dev = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
    gpu_ctx.render_rows_device(m, w, h, 0, h, 1, 0, dev.data_ptr())
    torch.cuda.synchronize()
    one_launch = dev.cpu().numpy()
    ref = oracle.render(objs, lights, m, w, h, row_step=100)
    _assert_same(one_launch[::100], ref[::100], "device-resident frame")
    pageable = np.zeros((h, w, 4), np.uint8)
    got, hit = gpu_ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True, out=pageable)
    assert got is pageable
    _assert_same(pageable, one_launch, "chunked copy into pageable memory")
    assert ((hit >= 0) == (one_launch[..., :3].sum(axis=-1) > 0)).mean() > 0.99  # hit pixels are lit (a few may quantise to black)
    pinned = gpu_ctx.host_frame((h, w, 4))
    try:
        gpu_ctx.render(m, w, h, 1, 0, out=pinned)
        _assert_same(pinned, one_launch, "chunked copy into the rt_host_alloc store")
        pinned[:] = 0
        gpu_ctx.render(m, w, h, 2, capi.RT_SHADOWS, out=pinned)  # general-ray kernels chunk by chunk
        gen = oracle.render(objs, lights, m, w, h, row_step=100, depth=2, flags=1)
        _assert_same(pinned[::100], gen[::100], "chunked general path")
    finally:
        gpu_ctx.host_free(pinned)
    rgb = np.zeros((h, w, 3), np.uint8)
    gpu_ctx.render(m, w, h, 1, capi.RT_OUT_RGB8, out=rgb)
    _assert_same(rgb, one_launch[..., :3], "chunked RGB8")
    with pytest.raises(ValueError):
        gpu_ctx.render(m, w, h, 1, 0, out=rgb)


def _with_planes(objs, planes):
    """Append plane elements (point, normal, reflectiveness) to a sphere-only FlatObjects; planes go in front of, between and behind the spheres."""
    f = np.float32
    n = objs.n
    cols = [list(map(float, a)) for a in (objs.cx, objs.cy, objs.cz, objs.radius, objs.refl_r, objs.refl_g, objs.refl_b)]
    kind, nx, ny, nz = [0] * n, [0.0] * n, [0.0] * n, [0.0] * n
    for j, (pt, nrm, refl) in enumerate(planes):
        nrm = np.asarray(nrm, f)
        nrm = nrm / np.sqrt((nrm * nrm).sum(), dtype=f)
        at = 0 if j == 0 else (len(kind) // 2 if j == 1 else len(kind))
        vals = (pt[0], pt[1], pt[2], float("nan"), refl[0], refl[1], refl[2])  # radius of a plane is ignored (NaN on purpose)
        for c, v in zip(cols, vals):
            c.insert(at, float(v))
        kind.insert(at, 1); nx.insert(at, float(nrm[0])); ny.insert(at, float(nrm[1])); nz.insert(at, float(nrm[2]))
    return capi.FlatObjects(*cols, kind=np.asarray(kind, np.uint8), nx=nx, ny=ny, nz=nz)


def test_plane_extension(gpu_ctx, oracle):
    """RT_PLANE (extension, semantics defined by the oracle): floor / wall / oblique planes mixed into sphere scenes — primary rays,
    bounces off and onto planes, shadows cast on planes (hit points far outside the sphere cloud take the exact shadow path),
    bands, hit indices, planes only, and the host mirror's Plane class."""
    w, h = 256, 144
    m = _cam(oracle, w, h)
    spheres, lights = scenegen.generate(w, h, 120, 5, seed=55)
    floor = ((0.0, 2.5, 0.0), (0.0, -1.0, 0.0), (0.8, 0.7, 0.6))          # below the cloud (image y is down), reaches the horizon
    wall = ((0.0, 0.0, -40.0), (0.0, 0.2, 1.0), (0.3, 0.5, 0.9))         # behind everything, slightly tilted
    cut = ((0.5, 0.0, -12.0), (1.0, 0.3, 0.4), (0.9, 0.2, 0.2))          # oblique, cuts through the cloud
    for planes in ([floor], [floor, wall], [floor, wall, cut]):
        objs = _with_planes(spheres, planes)
        gpu_ctx.scene_upload(objs, lights)
        for depth, flags in ((1, 0), (1, capi.RT_SHADOWS), (2, 0), (3, capi.RT_SHADOWS)):
            gpu, hit = gpu_ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
            ref, ref_hit = oracle.render(objs, lights, m, w, h, depth=depth, flags=1 if flags else 0, want_hit_idx=True)
            _assert_same(gpu, ref, f"{len(planes)} planes depth {depth} flags {flags}")
            assert np.array_equal(hit, ref_hit)
        assert (hit == 0).any()  # the floor (element 0) is visible
    parts = [gpu_ctx.render_rows(m, w, h, a, b, 2, capi.RT_SHADOWS)[0] for a, b in ((0, 50), (50, 51), (51, 144))]
    _assert_same(np.concatenate(parts), oracle.render(objs, lights, m, w, h, depth=2, flags=1), "planes, bands")
    # planes only (no sphere at all), light close to a plane
    z = np.zeros(0, np.float32)
    only = _with_planes(capi.FlatObjects(z, z, z, z, z, z, z), [floor, wall])
    near = capi.FlatLights([0.0, 3.0], [2.4, -1.0], [-6.0, -20.0], [1, 1], [1, 0.5], [1, 0.2], [2.0, 40.0])
    gpu_ctx.scene_upload(only, near)
    for depth, flags in ((1, 0), (2, capi.RT_SHADOWS)):
        _assert_same(gpu_ctx.render(m, w, h, depth, flags), oracle.render(only, near, m, w, h, depth=depth, flags=1 if flags else 0), "planes only")
    # through the host mirror: Plane + Sphere elements
    from raytracing_amd.scene import Element, Material, Plane, Sphere, Vec3, Point, flatten_elements, flatten_lights
    els = [Element(Sphere(Vec3(0.0, 0.0, -4.0), 1.0), Material(Vec3.ZERO, Vec3(1.0, 0.2, 0.2))),
           Element(Plane(Vec3(0.0, 1.0, 0.0), Vec3(0.0, -2.0, 0.0)), Material(Vec3.ZERO, Vec3(0.5, 0.5, 0.5)))]
    lts = [Point(Vec3(2.0, -4.0, -2.0), Vec3(1.0, 1.0, 1.0), 30.0)]
    fo, fl = flatten_elements(els), flatten_lights(lts)
    assert fo.kind.tolist() == [0, 1] and fo.ny[1] == np.float32(-1.0)
    gpu_ctx.scene_upload(fo, fl)
    got = gpu_ctx.render(m, w, h, 2, capi.RT_SHADOWS)
    _assert_same(got, oracle.render(fo, fl, m, w, h, depth=2, flags=1), "mirror scene with a plane")
    lit = gpu_ctx.render(m, w, h, 1, 0)
    shd = gpu_ctx.render(m, w, h, 1, capi.RT_SHADOWS)
    assert (shd[..., :3] < lit[..., :3]).any()  # the sphere casts a shadow on the floor
    bad = capi.FlatObjects([0], [0], [-5], [0], [1], [1], [1], kind=np.array([1], np.uint8))
    with pytest.raises(capi.RtError):
        gpu_ctx.scene_upload(bad, fl)  # plane without normal arrays


def test_native_bench_harness(tmp_path, gpu_ctx):
    """examples/bench_scene (C++: SCENE-GEN, C ABI, page-locked frame store, no Python) renders the same bytes as the Python path."""
    import subprocess
    exe = os.path.abspath(os.path.join(os.path.dirname(GOLDEN), "..", "examples", "bench_scene"))
    assert os.path.exists(exe), "build it first: make -C examples (done by __graft_entry__.build())"
    for name, shadows, extra in (("C3", 0, 0), ("C5", 1, 0), ("C3", 0, capi.RT_CULL)):
        dump = tmp_path / f"{name}.rgba"
        r = subprocess.run([exe, name, "2", str(shadows), "1", str(dump)], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, RT_FLAGS_EXTRA=str(extra)))
        assert r.returncode == 0, r.stderr
        assert "primary rays/s" in r.stdout
        w, h, objs, lights = scenegen.config(name)
        gpu_ctx.scene_upload(objs, lights)
        m = capi.camera_transform(*capi.camera_default(), np.float32(w) / np.float32(h))
        ref = gpu_ctx.render(m, w, h, 1, capi.RT_SHADOWS if shadows else 0)
        got = np.fromfile(dump, dtype=np.uint8).reshape(h, w, 4)
        _assert_same(got, ref, f"bench_scene {name}")
    assert subprocess.run([exe, "C9"], capture_output=True, timeout=60).returncode == 2


def test_rccl_band_gather_self_send(gpu_ctx, oracle):
    """raytracing_amd.rccl_gather on one GPU: a one-rank RCCL communicator where rank 0 sends its band to itself and receives it into
    its rows of the frame — the exact ncclSend / ncclRecv / group calls a peer and the root make in a multi-GPU run."""
    import torch
    from raytracing_amd import rccl_gather
    w, h = 640, 360
    objs, lights = scenegen.generate(w, h, 150, 3, seed=88)
    gpu_ctx.scene_upload(objs, lights)
    m = _cam(oracle, w, h)
    ref = gpu_ctx.render(m, w, h)
    g = rccl_gather.RcclBandGather(1, 0, 0)
    try:
        assert g.plan(h, w) == [(0, h, h * w * 4)]
        st = torch.cuda.Stream()
        band = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        frame = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        for _ in range(3):  # several frames back to back on one stream: render -> gather is plain stream order
            gpu_ctx.render_rows_device(m, w, h, 0, h, 1, 0, band.data_ptr(), 0, st.cuda_stream)
            g.gather(h, w, st.cuda_stream, band_ptr=band.data_ptr(), frame_ptr=frame.data_ptr(), self_send=True)
        st.synchronize()
        _assert_same(frame.cpu().numpy(), ref, "frame assembled by ncclSend/ncclRecv")
        g.gather(h, w, st.cuda_stream, frame_ptr=frame.data_ptr())  # one rank, no self-send: nothing to do
    finally:
        g.close()


def test_many_launches_in_flight_on_several_streams(oracle):
    """Tile-queue counters are per stream: hundreds of band launches enqueued at once on three streams (the host far ahead of the GPU,
    as in the multi-GPU pipeline) all produce the exact band; counter slots are recycled, so any number of streams works."""
    import torch
    w, h = 3840, 1000
    objs, lights = scenegen.generate(w, h, 64, 3, seed=91)
    m = _cam(oracle, w, h)
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        ref = ctx.render(m, w, h)
        oref = oracle.render(objs, lights, m, w, h, row_step=8)
        _assert_same(ref[::8], oref[::8], "reference frame of this test vs the oracle")
        streams = [torch.cuda.Stream() for _ in range(3)]
        # the three tile plans of DESIGN.md 3: 8-row tiles on a plain grid, 16-row tiles on a plain grid, the sectioned queue
        cuts = [(0, 100), (100, 350), (0, 1000)]
        plans = [capi.tile_plan(w, b - a, 1024) for a, b in cuts]  # 1024 = 256 CUs x 4 resident workgroups of render_kernel
        assert plans[0]["tiles4"] == 0 and plans[0]["tiles1"] == 0
        assert plans[1]["tiles2"] == 0 and plans[1]["tiles1"] == 0
        assert plans[2]["tiles4"] and plans[2]["tiles2"] and plans[2]["tiles1"]
        outs = [[torch.zeros((b - a, w, 4), dtype=torch.uint8, device="cuda") for (a, b) in cuts] for _ in streams]
        for k in range(240):
            s = k % 3
            a, b = cuts[(k // 3) % 3]
            ctx.render_rows_device(m, w, h, a, b, 1, 0, outs[s][(k // 3) % 3].data_ptr(), 0, streams[s].cuda_stream)
        torch.cuda.synchronize()
        for s in range(3):
            for (a, b), o in zip(cuts, outs[s]):
                _assert_same(o.cpu().numpy(), ref[a:b], f"stream {s} rows {a}:{b}")
        # Slots are recycled (rt_capi.hip: take_tile_slot): 40 fresh streams one after the other, each idle before the next launch ...
        buf = torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda")
        for _ in range(40):
            st = torch.cuda.Stream()
            ctx.render_rows_device(m, w, h, 0, 350, 1, 0, buf.data_ptr(), 0, st.cuda_stream)
            st.synchronize()
        _assert_same(buf[:350].cpu().numpy(), ref[:350], "40 fresh streams in sequence")
        # ... and 24 streams busy at once, more than the 16 slots: the extra launches wait for a slot instead of failing
        many = [torch.cuda.Stream() for _ in range(24)]
        bufs = [torch.zeros((1000, w, 4), dtype=torch.uint8, device="cuda") for _ in many]
        for rep in range(2):
            for st, b in zip(many, bufs):
                ctx.render_rows_device(m, w, h, 0, 1000, 1, 0, b.data_ptr(), 0, st.cuda_stream)
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            _assert_same(b.cpu().numpy(), ref, f"24 streams at once, stream {i}")
    # the same on a FRESH context: its first 8 streams are launched without slot events (nothing can need recycling yet), so the 17th
    # busy stream finds slots it knows nothing about and the table is reset after one device synchronisation (take_tile_slot)
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        for b in bufs:
            b.zero_()
        for rep in range(2):
            for st, b in zip(many, bufs):
                ctx.render_rows_device(m, w, h, 0, 1000, 1, 0, b.data_ptr(), 0, st.cuda_stream)
        torch.cuda.synchronize()
        for i, b in enumerate(bufs):
            _assert_same(b.cpu().numpy(), ref, f"fresh context, 24 streams at once, stream {i}")
        with pytest.raises(capi.RtError) as e:  # nothing was timed: the device entry point records events only with RT_TIMED
            ctx.last_kernel_ms()
        assert e.value.code == capi.RT_ERR_INVALID
        ctx.render_rows_device(m, w, h, 0, 1000, 1, capi.RT_TIMED, bufs[0].data_ptr(), 0, many[0].cuda_stream)
        assert 0.05 < ctx.last_kernel_ms() < 50.0


def test_general_launches_on_different_streams_are_ordered(oracle):
    """Launches with depth > 1 / RT_SHADOWS share the context's hit records; the library orders them (stream-level event wait), so a host
    that pipelines them over several streams gets the right images (rt_mi355x.h, rt_render_rows_device)."""
    import torch
    w, h = 1280, 720
    objs, lights = scenegen.generate(w, h, 300, 4, seed=17)
    m = _cam(oracle, w, h)
    with capi.Context(0) as ctx:
        ctx.scene_upload(objs, lights)
        modes = [(2, capi.RT_SHADOWS), (1, capi.RT_SHADOWS), (2, 0)]
        refs = [ctx.render(m, w, h, d, f) for d, f in modes]
        oref = oracle.render(objs, lights, m, w, h, row_step=45, depth=2, flags=1)
        _assert_same(refs[0][::45], oref[::45], "reference frame of this test vs the oracle")
        streams = [torch.cuda.Stream() for _ in range(3)]
        outs = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
        for k in range(6):  # six general launches enqueued at once, rotating over three streams
            d, f = modes[k % 3]
            ctx.render_rows_device(m, w, h, 0, h, d, f, outs[k].data_ptr(), 0, streams[k % 3].cuda_stream)
        torch.cuda.synchronize()
        for k in range(6):
            _assert_same(outs[k].cpu().numpy(), refs[k % 3], f"general launch {k}")


def test_multi_device_context_virtual_bands(oracle):
    """rt_ctx_create_multi: the band split and the RCCL gather BEHIND the C ABI.  One GPU here, so the device list
    repeats device 0 ("virtual bands"): bands 1.. are rendered into staging and travel through the library's own RCCL group (a send of
    the device to itself) into their rows of the frame — every call a real multi-GPU run makes except the link.  Byte-equal to the
    single launch: C3 at full size with {1, 8} bands through rt_render (host store), hit-index plane, RGB8, odd heights with
    3 and 5 bands, the general path, the device-resident entry point on a caller stream, and errors."""
    import torch
    w, h, objs, lights = scenegen.config("C3")
    m = _cam(oracle, w, h)
    with capi.Context(0) as one:
        one.scene_upload(objs, lights)
        ref, ref_hit = one.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
    oref = oracle.render(objs, lights, m, w, h, row_step=120)
    _assert_same(ref[::120], oref[::120], "single-launch C3 vs the oracle")
    for ids in ([0], [0] * 8):
        with capi.Context(device_ids=ids) as ctx:
            ctx.scene_upload(objs, lights)
            got = ctx.render(m, w, h, 1, 0)
            _assert_same(got, ref, f"C3 through rt_render, device list {ids}")
            assert ctx.last_kernel_ms() > 0.1
            if len(ids) > 1:
                got, hit = ctx.render_rows(m, w, h, 0, h, 1, 0, want_hit_idx=True)
                _assert_same(got, ref, "C3, 8 bands, with the hit-index plane")
                assert np.array_equal(hit, ref_hit)
                rgb = ctx.render(m, w, h, 1, capi.RT_OUT_RGB8)
                _assert_same(rgb, ref[..., :3], "C3, 8 bands, RGB8")
                band = ctx.render_rows(m, w, h, 1001, 1100)[0]  # 99 rows over 8 bands: uneven bands
                _assert_same(band, ref[1001:1100], "a 99-row band over 8 bands")
                # device-resident entry point on a caller stream, several frames back to back
                st = torch.cuda.Stream()
                frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(3)]
                for f in frames:
                    ctx.render_rows_device(m, w, h, 0, h, 1, 0, f.data_ptr(), 0, st.cuda_stream)
                st.synchronize()
                for f in frames:
                    _assert_same(f.cpu().numpy(), ref, "rt_render_rows_device on a multi-device context")
                with pytest.raises(capi.RtError) as e:
                    ctx.render_rows_device_noalloc(m, w, h, 0, h, 1, 0, frames[0].data_ptr(), 0, st.cuda_stream)
                assert e.value.code == capi.RT_ERR_INVALID
    # BASELINE configs 4 and 5 are the multi-GPU ones: C4 (8K, 8192 spheres) over 2 / 4 / 8 bands and C5 with the shadow extension over
    # 8 bands, at full size, byte-equal to the single launch (itself checked against the oracle in test_c4_c5_full_size_sampled_rows)
    for name, flags, splits in (("C4", 0, (2, 4, 8)), ("C5", capi.RT_SHADOWS, (8,))):
        w, h, objs, lights = scenegen.config(name)
        m = _cam(oracle, w, h)
        with capi.Context(0) as one:
            one.scene_upload(objs, lights)
            ref = one.render(m, w, h, 1, flags)
        for n_bands in splits:
            with capi.Context(device_ids=[0] * n_bands) as ctx:
                ctx.scene_upload(objs, lights)
                _assert_same(ctx.render(m, w, h, 1, flags), ref, f"{name} full size over {n_bands} bands")
        del ref
    # odd sizes, more bands than rows, the general path (every band's kernels use its part's hit records, in stream order)
    w, h = 333, 77
    objs, lights = scenegen.generate(w, h, 300, 5, seed=3)
    m = _cam(oracle, w, h)
    for n_bands in (3, 5, 64):
        with capi.Context(device_ids=[0] * n_bands) as ctx:
            ctx.scene_upload(objs, lights)
            for depth, flags in ((1, 0), (2, capi.RT_SHADOWS), (1, capi.RT_SHADOWS | capi.RT_OUT_RGB8)):
                got, hit = ctx.render_rows(m, w, h, 0, h, depth, flags, want_hit_idx=True)
                oref, ohit = oracle.render(objs, lights, m, w, h, depth=depth, flags=flags & 1, channels=3 if flags & 2 else 4, want_hit_idx=True)
                _assert_same(got, oref, f"{n_bands} bands, depth {depth}, flags {flags}")
                assert np.array_equal(hit, ohit)
            tiny = ctx.render_rows(m, w, h, 10, 12)[0]  # 2 rows over n bands: most bands are empty
            _assert_same(tiny, oracle.render(objs, lights, m, w, h, y0=10, y1=12), f"2 rows over {n_bands} bands")
            with pytest.raises(capi.RtError):
                ctx.render(m, w, h, 0)
    with pytest.raises(capi.RtError) as e:
        capi.Context(device_ids=[0, 99])
    assert e.value.code == capi.RT_ERR_NO_DEVICE
    with pytest.raises(capi.RtError):
        capi.Context(device_ids=[])


def test_multi_device_context_alternating_caller_streams(oracle):
    """A multi-device context whose device list repeats the ROOT device ([0] * 4) called with a different caller stream every frame, and
    with changing scenes' brightness so that a stale band would show: the staged bands of the root device are rendered and sent on the
    part's own stream, so frame N + 1's render into the staging is ordered after frame N's send out of it whatever the caller's
    streams are (they used to run on the caller's stream of each call)."""
    import torch
    w, h = 1920, 1080
    objs, lights = scenegen.generate(w, h, 600, 3, seed=9)
    m = _cam(oracle, w, h)
    variants = []
    for k in range(3):
        lk = capi.FlatLights(lights.px, lights.py, lights.pz, lights.cr, lights.cg, lights.cb, lights.intensity * np.float32(1.0 + 0.5 * k))
        variants.append(lk)
    refs = []
    with capi.Context(0) as one:
        for lk in variants:
            one.scene_upload(objs, lk)
            refs.append(one.render(m, w, h, 1, 0))
    streams = [torch.cuda.Stream() for _ in range(3)]
    with capi.Context(device_ids=[0] * 4) as ctx:
        for rep in range(2):
            for k, lk in enumerate(variants):
                ctx.scene_upload(objs, lk)  # (synchronises the devices of the context)
                frames = [torch.zeros((h, w, 4), dtype=torch.uint8, device="cuda") for _ in range(6)]
                for i, f in enumerate(frames):  # six frames in flight, the caller's stream changes every frame
                    ctx.render_rows_device(m, w, h, 0, h, 1, 0, f.data_ptr(), 0, streams[i % 3].cuda_stream)
                torch.cuda.synchronize()
                for i, f in enumerate(frames):
                    _assert_same(f.cpu().numpy(), refs[k], f"rep {rep} scene {k} frame {i} on alternating caller streams")


def test_multi_device_through_host_mirrors(tmp_path):
    """examples/main with RT_DEVICES=0,0,0 (a process without torch: the library loads RCCL itself) and the Python Renderer mirror with
    a device list both write the reference's result.png."""
    import subprocess
    gold = np.load(os.path.join(GOLDEN, "result_100x100.npy"))
    exe = os.path.abspath(os.path.join(os.path.dirname(GOLDEN), "..", "examples", "main"))
    env = dict(os.environ, RT_DEVICES="0,0,0")
    r = subprocess.run([exe], cwd=tmp_path, capture_output=True, text=True, timeout=180, env=env)
    assert r.returncode == 0, r.stderr
    from PIL import Image
    _assert_same(np.array(Image.open(tmp_path / "result.png")), gold, "C++ main, three virtual bands")
    # the native harness over four virtual bands renders the bytes of the single-device run
    bench = os.path.abspath(os.path.join(os.path.dirname(GOLDEN), "..", "examples", "bench_scene"))
    dumps = []
    for tag, e in (("one", dict(os.environ)), ("four", dict(os.environ, RT_DEVICES="0,0,0,0"))):
        dump = tmp_path / f"c3_{tag}.rgba"
        r = subprocess.run([bench, "C3", "2", "0", "1", str(dump)], capture_output=True, text=True, timeout=300, env=e)
        assert r.returncode == 0, r.stderr
        assert ("multi-device context: 4 row bands" in r.stdout) == (tag == "four")
        dumps.append(np.fromfile(dump, dtype=np.uint8))
    assert np.array_equal(dumps[0], dumps[1]) and dumps[0].any()
    elements, lights = scene.default_scene()
    rr = scene.Renderer(scene.Framebuffer(100, 100, scene.Camera()), elements, lights, device_ids=[0, 0, 0, 0, 0])
    rr.render(1, display_path=None)
    _assert_same(rr.last_frame, gold, "Python Renderer mirror, five virtual bands")


def test_multi_device_parts_with_fake_rccl():
    """The per-PART logic of a multi-device context (what runs for every device other than the root: own context, stream, staging,
    communicator rank, peer indices of the sends and receives) on the one-GPU box: tests/fake_rccl stands in for librccl (HIP copies
    with RCCL's matching and ordering rules; real RCCL refuses two ranks on one device) and RT_MULTI_SPLIT_PARTS=1 gives every list
    entry its own part.  Own process: a process loads one RCCL.  See tests/multi_parts_worker.py for what is asserted."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    fake = os.path.join(here, "fake_rccl", "libfake_rccl.so")
    assert os.path.exists(fake), "build it first: make -C tests/fake_rccl (done by __graft_entry__.build())"
    env = dict(os.environ, RT_RCCL_LIB=fake, RT_MULTI_SPLIT_PARTS="1")
    r = subprocess.run([sys.executable, os.path.join(here, "multi_parts_worker.py")], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("ranks", [2, 3])
def test_bench_multi_rank_control_flow_rehearsal(ranks):
    """bench.py with more than one rank, for real, on the one-GPU box: BENCH_BACKEND=gloo lets the ranks share the GPU and gathers the
    bands through host memory, everything else is the code a multi-GPU run executes — every rank's set-up, the collective-safe attempt
    at the direct RCCL gather (real RCCL refuses two ranks on one device: BOTH ranks must report it and agree on the fall-back, nobody
    hangs), the slot agreement, the band offsets of rank > 0, the timing reduction.  Rank 0 compares every assembled frame with a
    single launch.  3 ranks: uneven bands (2160 / 3 = 720 is even, so the height is not the point; the rank count is)."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(GOLDEN), ".."))
    port = 29650 + ranks
    env = dict(os.environ, BENCH_BACKEND="gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "12", "--warmup", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(line) == 1, r.stdout[-2000:]
    out = json.loads(line[0])
    assert out["n_gpus"] == ranks and out["steps"] == 12 and out["scaling"] == "strong"
    assert "assembled frames equal the single launch: True" in out["rehearsal"]
    assert "torch.distributed.gather" in out["config"]["parallelism"]  # the agreed fall-back
    assert 0 < out["roofline"]["frac"] < 1
    for k in range(ranks):
        assert ("rank %d: direct RCCL gather unavailable" % k) in r.stderr
