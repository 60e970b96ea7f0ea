/*
 * rt_mi355x.h — C ABI of librt_mi355x.so: the per-pixel render path of Aandreba/raytracing on MI355X (gfx950).
 *
 * The reference (Rust, /root/reference/src) has no FFI or plugin interface; this boundary is new and sits at the
 * seam `Renderer::render` (renderer.rs:32-82) / `Framebuffer::update` (display.rs:74-105): everything between
 * "scene slices + camera + W x H" and "bytes in Framebuffer.pixels".  Each entry point cites what it replaces.
 * INTEGRATION.md shows the Rust `extern "C"` block and the `Renderer::render` body a maintainer would add.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative RT_ERR_*; no exception crosses the boundary;
 *   - a context from rt_ctx_create is bound to ONE device (one process per GPU: one ctx per rank, row bands, see rt_render_rows*); a
 *     context from rt_ctx_create_multi spans a list of devices of this process and splits every call into row bands itself.  Neither
 *     is thread-safe (mirrors `&mut self` of Renderer::render, renderer.rs:32);
 *   - arrays passed in are borrowed for the duration of the call only;
 *   - every entry point makes ctx's device current while it runs and restores the caller's current device before it returns;
 *   - there is no CPU fallback: without a usable gfx950 device rt_ctx_create fails with RT_ERR_NO_DEVICE.
 */
#ifndef RT_MI355X_H
#define RT_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1)   /* bad argument (null pointer, depth == 0, unknown kind, rows out of range ...) */
#define RT_ERR_NO_DEVICE (-2) /* no HIP device / device id out of range / not gfx950 */
#define RT_ERR_HIP (-3)       /* a HIP runtime call failed; text in rt_last_error */
#define RT_ERR_NO_SCENE (-4)  /* rt_render* before rt_scene_upload */
#define RT_ERR_TOO_LARGE (-5) /* size beyond a documented limit */
#define RT_ERR_RCCL (-6)      /* multi-device context: RCCL could not be loaded or an RCCL call failed; text in rt_last_error */

/* object kinds — object/mod.rs:7-10 (trait Object); only Sphere exists in the reference (object/sphere.rs) */
#define RT_SPHERE 0u
#define RT_PLANE 1u /* EXTENSION (no reference counterpart): two-sided plane, an `impl Object` in the shape of object/sphere.rs:19-45 */
/* light kinds — light/mod.rs:8-10 (trait Light): light/point.rs, light/ambient.rs */
#define RT_POINT 0u
#define RT_AMBIENT 1u

/* rt_render* flags */
#define RT_SHADOWS 1u  /* EXTENSION (no reference counterpart): point lights are occluded by elements, see DESIGN.md */
#define RT_OUT_RGB8 2u /* write 3 B/pixel exactly like Framebuffer.pixels (display.rs:16-20,50-57); default RGBA8, A=255 */
#define RT_TIMED 4u    /* rt_render_rows_device only: record HIP events around the launch (and between its kernels) for rt_last_kernel_ms /
                          rt_last_phase_ms.  Off by default: an event costs ~4 us of GPU time between back-to-back launches (1.3 % of a
                          C3 frame for the pair).  The blocking host-output entry points always time their kernels; ignored elsewhere. */
#define RT_CULL 8u     /* OPT-IN, same bytes out: the primary rays skip the walk over the whole element list.  A wave's 16 x 16 pixel tile
                          is a narrow cone of directions from the common origin (renderer.rs:43), so clusters of spheres and then single
                          spheres outside that cone are voted out before the per-pixel filter and the exact test of object/sphere.rs:26-44
                          run on what is left.  Off by default: the default path is the per-pixel walk of the object list
                          that renderer.rs:48-54 is, measured against the FP32 roofline; with RT_CULL a C3 frame takes a fraction of the
                          time and no FLOP roofline applies. */

/* shadow-ray acceptance window of the RT_SHADOWS extension: occluded iff RT_SHADOW_EPS < t < distance(light) */
#define RT_SHADOW_EPS 1.0e-3f

/*
 * Flattened `&[Element<DynObject>]` (renderer.rs:37, element.rs:10-19, object/sphere.rs:6-9), SoA, element order
 * preserved (ties: lowest index wins, renderer.rs:49-53).  Only Material.reflectiveness is read by the reference
 * (renderer.rs:66), so only it crosses the boundary.
 */
typedef struct rt_objects {
    int32_t n;
    const uint8_t* kind; /* n entries, RT_SPHERE / RT_PLANE; NULL => all spheres */
    const float *cx, *cy, *cz, *radius; /* RT_PLANE: (cx,cy,cz) = a point of the plane, radius ignored */
    const float *refl_r, *refl_g, *refl_b;
    /* RT_PLANE only: unit normal.  is_hit_by(ray) = Some(t) iff t = dot3(point - origin, n) / dot3(direction, n) is finite and >= 0;
       normal(at) = n.  Ignored for spheres; may be NULL when no element is a plane. */
    const float *nx, *ny, *nz;
} rt_objects;

/* Flattened `&[DynLight]` (renderer.rs:38, light/point.rs:5-9, light/ambient.rs:5-7), list order preserved (renderer.rs:60-64). */
typedef struct rt_lights {
    int32_t n;
    const uint8_t* kind; /* n entries, RT_POINT / RT_AMBIENT; NULL => all point lights */
    const float *px, *py, *pz; /* ignored for RT_AMBIENT */
    const float *cr, *cg, *cb;
    const float* intensity; /* ignored for RT_AMBIENT */
} rt_lights;

typedef struct rt_ctx rt_ctx;

/* library/ABI version string, static storage; "rt_mi355x 0.2": rt_objects carries plane normals (0.1 did not) */
const char* rt_version(void);

/* Replaces nothing in the reference (it owns no device state): creates the per-device context (stream, events, scene buffers). */
int32_t rt_ctx_create(rt_ctx** out, int32_t device_id);
/*
 * The reference's one parallel axis is the rayon row split of display.rs:90-93; this is its multi-GPU form BEHIND the boundary
 * (SURVEY 8b / 8e): ONE context over a list of devices, so that `Renderer::render` (renderer.rs:32-82) calling rt_render once uses
 * them all.  Every entry of `device_ids` is one contiguous ROW BAND: entry g of n renders rows [y0 + rows*g/n, y0 + rows*(g+1)/n) of
 * each rt_render* call with global row indices (display.rs:99) on device device_ids[g]; the scene is replicated by rt_scene_upload;
 * band 0 is rendered in place in the frame on the ROOT device device_ids[0], the other bands into staging on their devices and
 * gathered into their rows of the frame by one group of ncclSend/ncclRecv per call (RCCL over xGMI, loaded at run time: $RT_RCCL_LIB if
 * set, else an RCCL already in the process, else librccl.so.1).  A device may appear more than once ("virtual bands": such a band
 * travels through RCCL too, as a send of the device to itself — how the exchange is exercised on a one-GPU machine).
 * The context works with every rt_* call except rt_render_rows_device_noalloc and rt_last_phase_ms; device pointers and streams
 * passed to rt_render_rows_device belong to the root device.  One host thread issues all launches: at 8 devices a C3-sized frame is
 * host-bound (about 10 us of launch calls per device); one process per GPU with rt_render_rows_device per rank (bench.py) is not.
 * n_devices in [1, 64].  RT_ERR_RCCL if RCCL cannot be loaded or initialised (n_devices == 1 never needs it).
 */
int32_t rt_ctx_create_multi(rt_ctx** out, const int32_t* device_ids, int32_t n_devices);
int32_t rt_ctx_destroy(rt_ctx* ctx);

/* Text of the last error on this ctx (or of the last failed rt_ctx_create when ctx == NULL); never freed by the caller. */
const char* rt_last_error(const rt_ctx* ctx);

/* Camera::transform (display.rs:30-41): row-major 4x4; host-only helper, touches no device. */
int32_t rt_camera_transform(float fov, float z_near, float z_far, float aspect_ratio, float m[16]);
/* Camera::default() (display.rs:118-123): fov = 60f32.to_radians(), z_near = 0.01, z_far = 1000 */
int32_t rt_camera_default(float* fov, float* z_near, float* z_far);

/*
 * Transform::apply (math/mod.rs:40-43): out = position + rotation.apply(v) (x) scale, rotation = a versor (r, i, j, k) applied as
 * (q * (0, v)) * conj(q) (quat.rs:159-166, quaternion product of quat.rs:201-226, same lane arithmetic).  Host-only helper: the
 * reference attaches no Transform to any object, so the host mirrors use this to pre-transform a sphere's centre
 * before flattening (radius x scale, uniform scale only); nothing of it runs on the device.
 */
int32_t rt_transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3]);

/*
 * Replaces the borrow of self.elements / self.lights at renderer.rs:37-38: copies the flattened scene to the device and
 * precomputes the per-sphere constants the kernels use.  May be called again to replace the scene.  At most 2^22 - 1 elements
 * (RT_ERR_INVALID beyond); for RT_SHADOWS the device keeps 20 bytes per (point light, element) of filter constants.
 */
int32_t rt_scene_upload(rt_ctx* ctx, const rt_objects* objects, const rt_lights* lights);

/*
 * Replaces the pixel loop of Renderer::render (renderer.rs:40-77 inside Framebuffer::update, display.rs:74-105) for the
 * whole frame: writes h rows of w pixels, row 0 = top, into caller-owned HOST memory (w*h*4 bytes, or w*h*3 with
 * RT_OUT_RGB8).  Blocking.  `m` is the row-major Mat4 of Camera::transform.  depth == 0 => RT_ERR_INVALID (the reference
 * underflows `depth - 1` at renderer.rs:36).
 */
int32_t rt_render(rt_ctx* ctx, const float m[16], uint32_t w, uint32_t h, uint32_t depth, uint32_t flags, uint8_t* out_host);

/*
 * Same for the row band [y0, y1) of a w x h frame (the rayon row split of display.rs:90-93 turned into a per-GPU band):
 * `out_host` receives (y1-y0) rows.  hit_idx_host (nullable, (y1-y0)*w int32) receives the index of the element the
 * PRIMARY ray hit, or -1 — the data behind the grazing-edge report.
 */
int32_t rt_render_rows(rt_ctx* ctx, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                       uint8_t* out_host, int32_t* hit_idx_host);

/*
 * Device-resident variant: `d_out` (and nullable `d_hit_idx`) are DEVICE pointers on ctx's device; the kernels are enqueued on
 * `stream` (a hipStream_t; NULL => ctx's own stream) and the call returns without synchronising.  This is what a multi-GPU
 * host uses per rank before gathering the bands (RCCL) — and what bench.py times.
 * Primary-only launches (depth == 1, no RT_SHADOWS) of one ctx overlap freely on different streams: every stream in use gets its own
 * tile-queue counters out of 16 slots, recycled as launches complete — any number of streams over the life of a ctx; with more than
 * 16 streams busy AT ONCE the extra launches wait (stream-level) for a slot.  Launches with depth > 1 or RT_SHADOWS share the ctx's
 * hit-record scratch: the library orders each one after the previous one (stream-level wait on an event), whatever their streams.
 * A stream must be idle when the caller destroys it: the library remembers streams by handle (their tile-queue slot), and a new stream
 * that re-uses the handle of one destroyed with work in flight would share that work's counters.
 * Multi-device contexts: the root device's bands run on `stream`, every other device's on a stream the context owns; a call on another
 * `stream` than the previous call's is ordered after that stream's work (the root's staging is shared between frames), so the
 * previous stream must still exist at that moment.
 */
int32_t rt_render_rows_device(rt_ctx* ctx, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                              void* d_out, void* d_hit_idx, void* stream);

/*
 * NO COUNTERPART IN THE REFERENCE (it owns no device memory).  For callers that cannot allocate device memory while they render:
 * code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.  Same arguments and same result as
 * rt_render_rows_device, but the call allocates and frees nothing: the kernels' scratch (tile-queue counters; for depth > 1 or
 * RT_SHADOWS the dense hit records) lives in a fixed array of RT_NOALLOC_SCRATCH_BYTES that every GPU the library runs on holds from
 * load to process exit.  Primary-only launches (depth == 1, no RT_SHADOWS) fit at any size; a general launch needs
 * 272 + 4 * (n + 4) + 4 * P * (y1 - y0) * w bytes, P = 7 + (depth > 1 ? 9 : 0) + (RT_SHADOWS ? ceil(n_lights / 32) : 0), n = elements of
 * the scene, and is refused with RT_ERR_TOO_LARGE beyond (e.g. shadows with <= 32 lights: 1.04 Mpixel per call; depth 2: 524 kpixel; cut
 * larger frames into row bands).
 * All calls on one GPU — of any context, of any thread — share that GPU's array: calls that overlap in time on one GPU race; order
 * them (one stream, or events).  Does not update rt_last_kernel_ms.  Capturable: everything the call enqueues is a kernel launch (the
 * scratch is cleared by a kernel too), so a HIP graph captured around it replays to the same bytes (tests/test_zz_no_alloc.py).
 */
#define RT_NOALLOC_SCRATCH_BYTES (32u << 20)
int32_t rt_render_rows_device_noalloc(rt_ctx* ctx, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth,
                                      uint32_t flags, void* d_out, void* d_hit_idx, void* stream);

/*
 * Framebuffer::new (display.rs:46-61) allocates the pixel store the host entry points write into (a raw allocation turned into
 * an ImageBuffer with from_raw, display.rs:50-57).  Taking that store from rt_host_alloc (page-locked host memory, hipHostMalloc)
 * lets rt_render / rt_render_rows copy the frame at PCIe rate and overlap the copy of one row chunk with the render of the
 * next; with ordinary (pageable) memory the copy goes through the driver's bounce buffers and blocks.  Optional.
 * (hipHostRegister of an existing allocation is refused on the pool this was developed on, hence alloc/free.)
 */
int32_t rt_host_alloc(rt_ctx* ctx, size_t bytes, void** out);
int32_t rt_host_free(rt_ctx* ctx, void* host_ptr);

/* Elapsed GPU time (ms, hipEvent) of the kernels of the most recent rt_render / rt_render_rows, or rt_render_rows_device launched with
   RT_TIMED, on this ctx; synchronises on its end event.  RT_ERR_INVALID if the most recent launch was not timed. */
int32_t rt_last_kernel_ms(rt_ctx* ctx, float* ms);

/*
 * Per-kernel GPU time (ms, hipEvent) of the most recent launch with depth > 1 or RT_SHADOWS on this ctx (the last row chunk, for the
 * host-output entry points): ms[0] = primary kernel that emits the hit records, ms[1] = shadow_mask_kernel launches (RT_SHADOWS; with
 * the clears of their occlusion words), ms[2] = shade_iter_kernel launches (shading + bounce trace), summed over the first 4
 * iterations of renderer.rs:46.  RT_ERR_INVALID if the most recent launch was primary-only or an rt_render_rows_device without RT_TIMED.
 * Synchronises on the launch's end.
 */
int32_t rt_last_phase_ms(rt_ctx* ctx, float ms[3]);

/*
 * Host-only introspection of the scene preprocessing done by rt_scene_upload (no device needed; used by the CPU tests):
 * for n spheres writes filt[4*n] = (cx', cy', cz', 0) of the conservative primary-ray filter  d.c' - 1 >= 0  and
 * cls[n] = 1 if the sphere is filterable, 0 if it is always tested exactly.  See DESIGN.md "primary-ray filter".
 */
int32_t rt_debug_prepare_spheres(const rt_objects* objects, float* filt, uint8_t* cls);

/*
 * Host-only introspection of the RT_SHADOWS any-hit hierarchy rt_scene_upload builds for point light `light` (no device needed; used
 * by the CPU tests; no reference counterpart, like RT_SHADOWS itself).  With n_pad = n rounded up to 16 (at least 16):
 * order[n_pad] = element at spatial position j (-1 = padding); lfilt[4*n_pad] = per position (c''x, c''y, c''z, P2): the shadow filter
 * w.c'' - 1 >= 0 and the constant of the member cone test; cl_a / cl_b[4*n_pad/16] = per cluster of 16 positions (cone axis, dmin) and
 * (cos, sin of its half-angle); always[<= n] / *n_always = the elements this light tests exactly for every ray.  See DESIGN.md 3.3.
 */
int32_t rt_debug_prepare_shadow_cull(const rt_objects* objects, const rt_lights* lights, int32_t light, int32_t* order, float* lfilt,
                                     float* cl_a, float* cl_b, int32_t* always, int32_t* n_always);

/*
 * Host-only introspection of the RT_CULL hierarchy rt_scene_upload builds for the primary rays (no device needed; used by the CPU tests;
 * no reference counterpart).  With n_pad = n rounded up to 16 (at least 16): order[n_pad] = element at position j of the direction order
 * (-1 = no entry: padding, planes, spheres the primary filter cannot cover); pfilt[4*n_pad] = per position (cx', cy', cz', P2): the primary
 * filter d.c' - 1 >= 0 of rt_debug_prepare_spheres and the constant of the member cone test; cl_a / cl_b[4*n_pad/16] = per cluster of 16
 * positions (cone axis, 0) and (cos, sin of its half-angle).  See DESIGN.md 3.5.
 */
int32_t rt_debug_prepare_primary_cull(const rt_objects* objects, int32_t* order, float* pfilt, float* cl_a, float* cl_b);

/*
 * Host-only introspection of how a band of `rows` rows x `w` columns is cut into the tile queue of the primary kernel when the
 * device holds `slots` resident workgroups (no device needed; used by the CPU tests): plan = {strips of 64 columns, rows in
 * 16-row tiles, rows in 8-row tiles, number of 16-, 8- and 4-row tiles}.  See DESIGN.md §3.
 */
int32_t rt_debug_tile_plan(uint32_t w, uint32_t rows, uint32_t slots, uint32_t plan[6]);

#ifdef __cplusplus
}
#endif
#endif
