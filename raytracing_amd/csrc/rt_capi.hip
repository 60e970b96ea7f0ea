// rt_capi.hip — the extern "C" boundary declared in include/rt_mi355x.h (host code; kernels live in render_kernels.hip).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "render_kernels.h"
#include "rt_mi355x.h"
#include "rt_multi.h"
#include "scene_prep.h"

using namespace rtamd;

// The band plan of a multi-device context for one call geometry (multi_plan)
struct MultiPlan {
    struct Band { uint32_t b0 = 0, b1 = 0; int part = 0; size_t off = 0, hit_off = 0; };
    bool valid = false;
    uint32_t w = 0, y0 = 0, y1 = 0;
    size_t bpp = 0;
    bool hit = false;
    std::vector<Band> bands;      // one per entry of the device list
    std::vector<size_t> need;     // staging bytes per part
    std::vector<uint32_t> order;  // non-empty bands, grouped by part
    std::vector<BandXfer> xfers;  // reused call after call
};

struct rt_ctx {
    int device = -1;
    hipStream_t stream = nullptr, copy_stream = nullptr;
    hipStream_t stream2 = nullptr;  // second render stream of the chunked host-output path (chunks alternate, so one chunk's drain overlaps the next one's ramp)
    hipEvent_t ev0 = nullptr;  // start of the most recent launch (its end: ev_end)
    hipEvent_t ev_chunk[16] = {};
    hipEvent_t ev_host_end = nullptr;  // end of the kernels of the most recent host-output call (both render streams)
    bool timed = false;
    std::string err;
    // device copies of the prepared scene
    bool have_scene = false;
    int32_t n = 0, n_pad = 0, n_always = 0, n_lights = 0;
    void* d_scene = nullptr;  // one allocation, sub-arrays below point into it
    size_t scene_bytes = 0;
    const float4 *d_filt = nullptr, *d_sph = nullptr, *d_sph_r = nullptr, *d_gen_o = nullptr, *d_gcl = nullptr, *d_refl = nullptr, *d_lpos = nullptr, *d_lcol = nullptr;
    const int32_t* d_always = nullptr;
    const float4 *d_lfilt = nullptr, *d_lclA = nullptr, *d_lclB = nullptr, *d_sph_o = nullptr, *d_pfilt_o = nullptr, *d_pclA = nullptr, *d_pclB = nullptr;
    const int32_t* d_porder = nullptr;
    const int32_t *d_order = nullptr, *d_rank = nullptr;
    const int32_t *d_lalways = nullptr, *d_lalways_off = nullptr;
    const float4* d_pln = nullptr;
    const int32_t* d_planes = nullptr;
    int32_t n_planes = 0;
    float ball[4] = {0, 0, 0, 0};
    // staging for the host-output entry points
    void* d_out = nullptr;
    size_t d_out_bytes = 0;
    void* d_hit = nullptr;
    size_t d_hit_bytes = 0;
    // dense hit records for the general path: [count (16 B) | pix | idx | t], capacity in pixels
    void* d_rec = nullptr;
    size_t rec_capacity = 0;   // records
    size_t rec_planes = 0;     // 4-byte planes of rec_capacity entries allocated after the 16-byte counter
    size_t rec_bin_words = 0;  // words of sort bins allocated after the planes
    // tile-queue counters of the persistent primary kernel: a ring of slots (64 B apart), one per launch in flight
    uint32_t* d_tile_ctr = nullptr;
    hipStream_t tile_slot_stream[16] = {};  // slot i was last used by this stream (launches of one stream execute in order and can share it)
    hipEvent_t tile_slot_ev[16] = {};       // recorded after the last launch that used slot i (once slot events are on): the slot is free again once it has completed
    bool tile_slot_ev_valid[16] = {};       // that event really follows the slot's last launch
    bool slot_events_on = false;            // a context that has seen few streams records nothing per launch (an event costs ~4 us of GPU timeline between launches)
    uint64_t tile_slot_use[16] = {};        // use clock of that launch (least recently used slot is the one recycled)
    uint64_t use_clock = 0;
    uint32_t tile_slots_used = 0;
    hipEvent_t ev_end = nullptr;            // end of the most recent TIMED launch (ev_dev_end / ev_host_end / ev_multi_end), for rt_last_kernel_ms
    hipEvent_t ev_dev_end = nullptr;        // end of the most recent rt_render_rows_device launch with RT_TIMED
    // launches with depth > 1 or RT_SHADOWS share d_rec: each one waits (stream-level) for the previous one
    hipEvent_t ev_general = nullptr;
    PhaseEvents phases = {};                // per-kernel timing of the most recent general launch (rt_last_phase_ms)
    bool phases_valid = false;
    bool general_pending = false;
    hipStream_t general_stream = nullptr;
    // this device's copy of the file-scope scratch array of the allocation-free entry point (rt_render_rows_device_noalloc)
    unsigned char* noalloc_scratch = nullptr;
    // staging of the bands this context renders as a PART of a multi-device context: [pixels | hit indices] per band
    void* d_band = nullptr;
    size_t d_band_bytes = 0;
    // ---- multi-device context (rt_ctx_create_multi): owns no scene or scratch of its own, only the parts, the band list and two events
    bool multi = false;
    std::vector<rt_ctx*> parts;   // one single-device context per DISTINCT device of the list; parts[0] is on the root device
    std::vector<int> band_part;   // per band (= entry of the device list): index into parts
    BandGather gather;            // RCCL communicators over the parts (only when there is more than one band)
    MultiPlan plan;
    hipStream_t last_root_stream = nullptr;  // the caller stream of the previous call (multi_enqueue orders a call on another stream after it)
    bool last_root_stream_set = false;
    hipEvent_t ev_stream_switch = nullptr;
    hipEvent_t ev_multi_end = nullptr;
};

static constexpr uint32_t kHostChunks = 4;   // row chunks of a large frame in the host-output entry points (render / copy pipeline; 6 / 8 / 12 / 16 measured no better, profiles/r02_tuning.md)
static constexpr uint32_t kTileSlots = 16;  // launches of one context that can be in flight on different streams without waiting for each other

static thread_local std::string g_create_err;

static int32_t fail(rt_ctx* ctx, int32_t code, const std::string& msg)
{
    if (ctx)
        ctx->err = msg;
    else
        g_create_err = msg;
    return code;
}

static int32_t fail_hip(rt_ctx* ctx, hipError_t e, const char* what)
{
    return fail(ctx, RT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define RT_HIP(ctx, call)                                      \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return fail_hip(ctx, e_, #call); \
    } while (0)

// Makes the context's device current for the duration of an entry point and puts the caller's device back afterwards
// (a host with several contexts, or a framework that tracks its own current device, must not find it changed).
struct DeviceGuard {
    int prev = -1;
    hipError_t status = hipSuccess;
    explicit DeviceGuard(int device)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != device) {
            status = hipSetDevice(device);
            if (status == hipSuccess) prev = cur;
        }
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

#define RT_DEVICE(ctx)                    \
    DeviceGuard device_guard_((ctx)->device); \
    RT_HIP(ctx, device_guard_.status)

// ---- scratch of the allocation-free entry point (rt_render_rows_device_noalloc) -----------------------------------------------------
// A fixed file-scope device array instead of hipMalloc'ed buffers: every GPU this library runs on holds its own copy at its own address
// until the process ends.  In host code the NAME `scratch` is only a stand-in for HIP_SYMBOL(); the address a kernel, a memset or a copy
// can use is the one hipGetSymbolAddress returns with that device current, looked up once per device (rt_ctx_create) and kept here.
static __device__ __align__(256) unsigned char scratch[RT_NOALLOC_SCRATCH_BYTES];
static constexpr size_t kNoAllocRecOffset = kTileSlotWords * 4 < 256 ? 256 : kTileSlotWords * 4;  // [0, slot): tile-queue counters; then the hit-record count + planes

static unsigned char* g_scratch_ptr[64] = {};  // per device; a benign race: concurrent callers store the same value

static hipError_t noalloc_scratch_of_device(int device, unsigned char** out)  // `device` must be current
{
    *out = nullptr;
    if (device < 0 || device >= 64) return hipErrorInvalidDevice;
    if (!g_scratch_ptr[device]) {
        void* p = nullptr;
        const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        // the tile-queue counters start at zero ONCE: the kernels re-arm them as they end (render_kernel), so no call has to clear them
        const hipError_t ez = hipMemset(p, 0, kNoAllocRecOffset);
        if (ez != hipSuccess) return ez;
        g_scratch_ptr[device] = static_cast<unsigned char*>(p);
    }
    *out = g_scratch_ptr[device];
    return hipSuccess;
}


extern "C" {

const char* rt_version(void) { return "rt_mi355x 0.2 (gfx950)"; }

const char* rt_last_error(const rt_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_err.c_str(); }

int32_t rt_camera_transform(float fov, float z_near, float z_far, float aspect_ratio, float m[16])
{
    if (!m) return RT_ERR_INVALID;
    camera_transform(fov, z_near, z_far, aspect_ratio, m);
    return RT_OK;
}

int32_t rt_transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3])
{
    if (!position || !versor || !scale || !v || !out) return RT_ERR_INVALID;
    transform_apply(position, versor, scale, v, out);
    return RT_OK;
}

int32_t rt_camera_default(float* fov, float* z_near, float* z_far)
{
    if (!fov || !z_near || !z_far) return RT_ERR_INVALID;
    const float pi = 3.14159265358979323846f;
    const float rads_per_deg = pi / 180.0f;  // f32::to_radians (display.rs:121)
    *fov = 60.0f * rads_per_deg;
    *z_near = 0.01f;
    *z_far = 1000.0f;
    return RT_OK;
}

int32_t rt_debug_prepare_spheres(const rt_objects* o, float* filt, uint8_t* cls)
{
    if (!o || !filt || !cls || o->n < 0) return RT_ERR_INVALID;
    if (o->n > 0 && (!o->cx || !o->cy || !o->cz || !o->radius)) return RT_ERR_INVALID;
    prepare_filter(o->n, o->cx, o->cy, o->cz, o->radius, reinterpret_cast<f4*>(filt), cls, nullptr);
    return RT_OK;
}

int32_t rt_debug_prepare_shadow_cull(const rt_objects* o, const rt_lights* l, int32_t light, int32_t* order, float* lfilt, float* cl_a, float* cl_b,
                                     int32_t* always, int32_t* n_always)
{
    if (!o || !l || !order || !lfilt || !cl_a || !cl_b || !always || !n_always || light < 0 || light >= l->n) return RT_ERR_INVALID;
    PreparedScene ps;
    if (!prepare_scene(o, l, ps).empty()) return RT_ERR_INVALID;
    const size_t np = (size_t)ps.n_pad, ncl = np / kShadowCluster;
    std::memcpy(order, ps.order.data(), np * sizeof(int32_t));
    std::memcpy(lfilt, ps.lfilt.data() + (size_t)light * np, np * 16);
    std::memcpy(cl_a, ps.lclA.data() + (size_t)light * ncl, ncl * 16);
    std::memcpy(cl_b, ps.lclB.data() + (size_t)light * ncl, ncl * 16);
    const int32_t a0 = ps.lalways_off[light], a1 = ps.lalways_off[light + 1];
    *n_always = a1 - a0;
    if (a1 > a0) std::memcpy(always, ps.lalways.data() + a0, (size_t)(a1 - a0) * sizeof(int32_t));
    return RT_OK;
}

int32_t rt_debug_prepare_primary_cull(const rt_objects* o, int32_t* order, float* pfilt, float* cl_a, float* cl_b)
{
    if (!o || !order || !pfilt || !cl_a || !cl_b) return RT_ERR_INVALID;
    const rt_lights none = {};
    PreparedScene ps;
    if (!prepare_scene(o, &none, ps).empty()) return RT_ERR_INVALID;
    const size_t np = (size_t)ps.n_pad, ncl = np / kShadowCluster;
    std::memcpy(order, ps.porder.data(), np * sizeof(int32_t));
    std::memcpy(pfilt, ps.pfilt_o.data(), np * 16);
    std::memcpy(cl_a, ps.pclA.data(), ncl * 16);
    std::memcpy(cl_b, ps.pclB.data(), ncl * 16);
    return RT_OK;
}

#ifdef RT_MASK_STATS
// DEV build only: the 16 words of tile-queue slot 0 (words 2.. hold the shadow kernel's statistics), then zero words 2..15
int32_t rt_debug_read_stats(rt_ctx* c, uint32_t out[16])
{
    if (!c || !out) return RT_ERR_INVALID;
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(out, c->d_tile_ctr, 64, hipMemcpyDeviceToHost);
    (void)hipMemset(c->d_tile_ctr + 2, 0, 56);
    return RT_OK;
}
#endif

int32_t rt_debug_tile_plan(uint32_t w, uint32_t rows, uint32_t slots, uint32_t plan[6])
{
    if (!plan || w == 0 || slots == 0) return RT_ERR_INVALID;
    const TilePlan tp = plan_tiles(w, rows, slots);
    plan[0] = tp.strips; plan[1] = tp.rows4; plan[2] = tp.rows2; plan[3] = tp.tiles4; plan[4] = tp.tiles2; plan[5] = tp.tiles1;
    return RT_OK;
}

int32_t rt_ctx_create(rt_ctx** out, int32_t device_id)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create: out is null");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) return fail(nullptr, RT_ERR_NO_DEVICE, std::string("no HIP device: ") + hipGetErrorString(e));
    if (device_id < 0 || device_id >= count) return fail(nullptr, RT_ERR_NO_DEVICE, "device id out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device_id);
    if (e != hipSuccess) return fail_hip(nullptr, e, "hipGetDeviceProperties");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RT_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName + ", this library carries gfx950 code only");
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return fail(nullptr, RT_ERR_HIP, "out of host memory");
    c->device = device_id;
    DeviceGuard guard(device_id);
    if ((e = guard.status) != hipSuccess || (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev_dev_end)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_general, hipEventDisableTiming)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess || (e = hipEventCreate(&c->ev_host_end)) != hipSuccess ||
        (e = hipMalloc(reinterpret_cast<void**>(&c->d_tile_ctr), kTileSlots * kTileSlotWords * 4)) != hipSuccess ||
        (e = hipMemset(c->d_tile_ctr, 0, kTileSlots * kTileSlotWords * 4)) != hipSuccess) {
        int32_t rc = fail_hip(nullptr, e, "context setup");
        rt_ctx_destroy(c);
        return rc;
    }
    for (hipEvent_t& ce : c->ev_chunk)
        if ((e = hipEventCreateWithFlags(&ce, hipEventDisableTiming)) != hipSuccess) {
            int32_t rc = fail_hip(nullptr, e, "context setup (chunk events)");
            rt_ctx_destroy(c);
            return rc;
        }
    for (hipEvent_t& pe : c->phases.ev)
        if ((e = hipEventCreate(&pe)) != hipSuccess) {
            int32_t rc = fail_hip(nullptr, e, "context setup (phase events)");
            rt_ctx_destroy(c);
            return rc;
        }
    for (uint32_t i = 0; i < kTileSlots; ++i)
        if ((e = hipEventCreateWithFlags(&c->tile_slot_ev[i], hipEventDisableTiming)) != hipSuccess) {
            int32_t rc = fail_hip(nullptr, e, "context setup (slot events)");
            rt_ctx_destroy(c);
            return rc;
        }
    // Everything a launch would otherwise look up on first use (resident workgroup slots of the persistent kernels, the device
    // address of the allocation-free entry point's scratch array) is looked up here, so that no render call does set-up work.
    if ((e = prime_launch_caches()) != hipSuccess || (e = noalloc_scratch_of_device(device_id, &c->noalloc_scratch)) != hipSuccess) {
        int32_t rc = fail_hip(nullptr, e, "context setup (launch caches / scratch symbol)");
        rt_ctx_destroy(c);
        return rc;
    }
    *out = c;
    return RT_OK;
}

int32_t rt_ctx_create_multi(rt_ctx** out, const int32_t* device_ids, int32_t n_devices)
{
    if (!out) return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create_multi: out is null");
    *out = nullptr;
    if (!device_ids || n_devices < 1 || n_devices > 64) return fail(nullptr, RT_ERR_INVALID, "rt_ctx_create_multi: 1 to 64 bands (entries of the device list)");
    rt_ctx* c = new (std::nothrow) rt_ctx();
    if (!c) return fail(nullptr, RT_ERR_HIP, "out of host memory");
    c->multi = true;
    c->device = device_ids[0];
    std::vector<int> devices;  // distinct, in order of first appearance: devices[0] is the root
    // TEST HOOK: RT_MULTI_SPLIT_PARTS=1 gives every list entry a part of its own even when devices repeat, so that the per-part logic
    // (own stream, staging, communicator rank) runs on a one-GPU machine.  Real RCCL refuses two ranks on one device, so the hook is only
    // honoured together with an explicitly named RCCL stand-in (RT_RCCL_LIB = the fake RCCL of tests/fake_rccl); a production process
    // that sets neither variable can never reach it.
    const char* split_env = std::getenv("RT_MULTI_SPLIT_PARTS");
    const char* rccl_env = std::getenv("RT_RCCL_LIB");
    const bool split_parts = split_env && split_env[0] == '1' && rccl_env && rccl_env[0];
    for (int32_t g = 0; g < n_devices; ++g) {
        size_t p = 0;
        while (p < devices.size() && (split_parts || devices[p] != device_ids[g])) ++p;
        if (p == devices.size()) {
            rt_ctx* part = nullptr;
            const int32_t rc = rt_ctx_create(&part, device_ids[g]);  // error text is already in g_create_err
            if (rc != RT_OK) {
                rt_ctx_destroy(c);
                return rc;
            }
            devices.push_back(device_ids[g]);
            c->parts.push_back(part);
        }
        c->band_part.push_back((int)p);
    }
    {
        DeviceGuard guard(c->device);
        hipError_t e;
        if ((e = guard.status) != hipSuccess || (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev_multi_end)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&c->ev_stream_switch, hipEventDisableTiming)) != hipSuccess) {
            const int32_t rc = fail_hip(nullptr, e, "multi-device context setup");
            rt_ctx_destroy(c);
            return rc;
        }
    }
    if (n_devices > 1) {  // one band needs no exchange at all: RCCL is not even loaded
        const std::string msg = c->gather.init(devices);
        if (!msg.empty()) {
            rt_ctx_destroy(c);
            return fail(nullptr, RT_ERR_RCCL, "rt_ctx_create_multi: " + msg);
        }
    }
    *out = c;
    return RT_OK;
}

int32_t rt_ctx_destroy(rt_ctx* c)
{
    if (!c) return RT_OK;
    if (c->multi) {
        for (rt_ctx* p : c->parts) {  // every launch and transfer has finished before the communicators go
            DeviceGuard g(p->device);
            (void)hipDeviceSynchronize();
        }
        c->gather.destroy();
        {
            DeviceGuard g(c->device >= 0 ? c->device : 0);
            if (c->ev0) (void)hipEventDestroy(c->ev0);
            if (c->ev_multi_end) (void)hipEventDestroy(c->ev_multi_end);
            if (c->ev_stream_switch) (void)hipEventDestroy(c->ev_stream_switch);
        }
        for (rt_ctx* p : c->parts) (void)rt_ctx_destroy(p);
        delete c;
        return RT_OK;
    }
    DeviceGuard guard(c->device >= 0 ? c->device : 0);
    (void)hipDeviceSynchronize();  // launches of this context may still run on caller streams and read its buffers
    if (c->d_scene) (void)hipFree(c->d_scene);
    if (c->d_out) (void)hipFree(c->d_out);
    if (c->d_hit) (void)hipFree(c->d_hit);
    if (c->d_rec) (void)hipFree(c->d_rec);
    if (c->d_band) (void)hipFree(c->d_band);
    if (c->d_tile_ctr) (void)hipFree(c->d_tile_ctr);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev_dev_end) (void)hipEventDestroy(c->ev_dev_end);
    if (c->ev_general) (void)hipEventDestroy(c->ev_general);
    for (hipEvent_t e : c->tile_slot_ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->phases.ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : c->ev_chunk)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_host_end) (void)hipEventDestroy(c->ev_host_end);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return RT_OK;
}

int32_t rt_scene_upload(rt_ctx* c, const rt_objects* o, const rt_lights* l)
{
    if (!c) return RT_ERR_INVALID;
    if (c->multi) {  // the scene is replicated: every device keeps its own copy (at most a few MiB)
        c->have_scene = false;
        for (rt_ctx* p : c->parts) {
            const int32_t rc = rt_scene_upload(p, o, l);
            if (rc != RT_OK) return fail(c, rc, p->err);
        }
        c->have_scene = true;
        return RT_OK;
    }
    PreparedScene ps;
    std::string msg = prepare_scene(o, l, ps);
    if (!msg.empty()) return fail(c, RT_ERR_INVALID, "rt_scene_upload: " + msg);
    RT_DEVICE(c);

    // one packed allocation: filt | sph | sph_r | gen | refl | lpos | lcol | always | lfilt | lalways | lalways_off | pln | planes (all 16-byte aligned)
    const size_t np = (size_t)ps.n_pad, nl = (size_t)ps.n_lights, na = ps.always.size();
    const size_t off_filt = 0, off_sph = off_filt + np * 16, off_sph_r = off_sph + np * 16, off_gen = off_sph_r + np * 16, off_refl = off_gen + np * 16;
    const size_t off_lpos = off_refl + np * 16, off_lcol = off_lpos + (nl ? nl : 1) * 16, off_alw = off_lcol + (nl ? nl : 1) * 16;
    const size_t off_lfilt = off_alw + ((na ? na : 1) * 4 + 15) / 16 * 16;
    const size_t nla = ps.lalways.size();
    const size_t off_lalw = off_lfilt + (nl ? nl : 1) * np * 16;
    const size_t off_laoff = off_lalw + ((nla ? nla : 1) * 4 + 15) / 16 * 16;
    const size_t off_pln = off_laoff + ((nl + 1) * 4 + 15) / 16 * 16;
    const size_t npl = ps.planes.size();
    const size_t off_planes = off_pln + np * 16;
    const size_t ncl = np / kShadowCluster;
    const size_t off_clA = off_planes + ((npl ? npl : 1) * 4 + 15) / 16 * 16;
    const size_t off_clB = off_clA + (nl ? nl : 1) * ncl * 16;
    const size_t off_order = off_clB + (nl ? nl : 1) * ncl * 16;
    const size_t off_rank = off_order + np * 4;
    const size_t off_sph_o = off_rank + np * 4;
    const size_t off_gcl = off_sph_o + np * 16;
    const size_t off_porder = off_gcl + ncl * 16;
    const size_t off_pfilt = off_porder + np * 4;
    const size_t off_pclA = off_pfilt + np * 16;
    const size_t off_pclB = off_pclA + ncl * 16;
    const size_t total = off_pclB + ncl * 16;
    std::vector<unsigned char> host(total, 0);
    std::memcpy(host.data() + off_filt, ps.filt.data(), np * 16);
    std::memcpy(host.data() + off_sph, ps.sph.data(), np * 16);
    std::memcpy(host.data() + off_sph_r, ps.sph_r.data(), np * 16);
    std::memcpy(host.data() + off_gen, ps.gen_o.data(), np * 16);
    std::memcpy(host.data() + off_refl, ps.refl.data(), np * 16);
    if (nl) std::memcpy(host.data() + off_lpos, ps.lpos.data(), nl * 16);
    if (nl) std::memcpy(host.data() + off_lcol, ps.lcol.data(), nl * 16);
    if (na) std::memcpy(host.data() + off_alw, ps.always.data(), na * 4);
    if (nl) std::memcpy(host.data() + off_lfilt, ps.lfilt.data(), nl * np * 16);
    if (nla) std::memcpy(host.data() + off_lalw, ps.lalways.data(), nla * 4);
    std::memcpy(host.data() + off_laoff, ps.lalways_off.data(), (nl + 1) * 4);
    std::memcpy(host.data() + off_pln, ps.pln.data(), np * 16);
    if (npl) std::memcpy(host.data() + off_planes, ps.planes.data(), npl * 4);
    if (nl) std::memcpy(host.data() + off_clA, ps.lclA.data(), nl * ncl * 16);
    if (nl) std::memcpy(host.data() + off_clB, ps.lclB.data(), nl * ncl * 16);
    std::memcpy(host.data() + off_order, ps.order.data(), np * 4);
    std::memcpy(host.data() + off_rank, ps.rank.data(), np * 4);
    std::memcpy(host.data() + off_sph_o, ps.sph_o.data(), np * 16);
    std::memcpy(host.data() + off_gcl, ps.gcl.data(), ncl * 16);
    std::memcpy(host.data() + off_porder, ps.porder.data(), np * 4);
    std::memcpy(host.data() + off_pfilt, ps.pfilt_o.data(), np * 16);
    std::memcpy(host.data() + off_pclA, ps.pclA.data(), ncl * 16);
    std::memcpy(host.data() + off_pclB, ps.pclB.data(), ncl * 16);

    RT_HIP(c, hipDeviceSynchronize());  // no render may still read the old scene (launches may sit on caller streams too)
    if (total > c->scene_bytes) {
        if (c->d_scene) RT_HIP(c, hipFree(c->d_scene));
        c->d_scene = nullptr;
        c->scene_bytes = 0;
        c->have_scene = false;
        RT_HIP(c, hipMalloc(&c->d_scene, total));
        c->scene_bytes = total;
    }
    RT_HIP(c, hipMemcpyAsync(c->d_scene, host.data(), total, hipMemcpyHostToDevice, c->stream));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    const char* base = static_cast<const char*>(c->d_scene);
    c->d_filt = reinterpret_cast<const float4*>(base + off_filt);
    c->d_sph = reinterpret_cast<const float4*>(base + off_sph);
    c->d_sph_r = reinterpret_cast<const float4*>(base + off_sph_r);
    c->d_gen_o = reinterpret_cast<const float4*>(base + off_gen);
    c->d_refl = reinterpret_cast<const float4*>(base + off_refl);
    c->d_lpos = reinterpret_cast<const float4*>(base + off_lpos);
    c->d_lcol = reinterpret_cast<const float4*>(base + off_lcol);
    c->d_always = reinterpret_cast<const int32_t*>(base + off_alw);
    c->d_lfilt = reinterpret_cast<const float4*>(base + off_lfilt);
    c->d_lalways = reinterpret_cast<const int32_t*>(base + off_lalw);
    c->d_lalways_off = reinterpret_cast<const int32_t*>(base + off_laoff);
    c->d_pln = reinterpret_cast<const float4*>(base + off_pln);
    c->d_planes = reinterpret_cast<const int32_t*>(base + off_planes);
    c->d_lclA = reinterpret_cast<const float4*>(base + off_clA);
    c->d_lclB = reinterpret_cast<const float4*>(base + off_clB);
    c->d_order = reinterpret_cast<const int32_t*>(base + off_order);
    c->d_rank = reinterpret_cast<const int32_t*>(base + off_rank);
    c->d_sph_o = reinterpret_cast<const float4*>(base + off_sph_o);
    c->d_gcl = reinterpret_cast<const float4*>(base + off_gcl);
    c->d_porder = reinterpret_cast<const int32_t*>(base + off_porder);
    c->d_pfilt_o = reinterpret_cast<const float4*>(base + off_pfilt);
    c->d_pclA = reinterpret_cast<const float4*>(base + off_pclA);
    c->d_pclB = reinterpret_cast<const float4*>(base + off_pclB);
    c->n_planes = (int32_t)npl;
    std::memcpy(c->ball, ps.ball, sizeof(c->ball));
    c->n = ps.n;
    c->n_pad = ps.n_pad;
    c->n_always = (int32_t)na;
    c->n_lights = ps.n_lights;
    c->have_scene = true;
    return RT_OK;
}

static int32_t check_render_args(rt_ctx* c, const float* m, uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags)
{
    if (!c) return RT_ERR_INVALID;
    if (!m) return fail(c, RT_ERR_INVALID, "camera matrix is null");
    if (!c->have_scene) return fail(c, RT_ERR_NO_SCENE, "rt_scene_upload has not been called");
    if (depth == 0) return fail(c, RT_ERR_INVALID, "depth must be >= 1 (the reference computes depth - 1 on usize, renderer.rs:36)");
    if (w == 0 || h == 0) return fail(c, RT_ERR_INVALID, "empty frame");
    if (y0 > y1 || y1 > h) return fail(c, RT_ERR_INVALID, "row band outside the frame");
    if (flags & ~(RT_SHADOWS | RT_OUT_RGB8 | RT_TIMED | RT_CULL)) return fail(c, RT_ERR_INVALID, "unknown flag bits");
    if ((uint64_t)w * (uint64_t)h > (1ull << 31)) return fail(c, RT_ERR_TOO_LARGE, "frame larger than 2^31 pixels");
    return RT_OK;
}

}  // extern "C"

// Where a launch finds its scratch (device memory beyond inputs and outputs): the context's own, grown on demand, or the fixed
// file-scope array of the allocation-free entry point.
enum class ScratchKind { kContext, kStatic };

// The tile-queue counters of a launch on `st`: kernels that can run concurrently must not share a queue, kernels of one stream run in
// order and can.  Slots are recycled: same stream first; then a free slot; with the table full, a slot whose last launch has completed
// (its event), else the least recently used slot — and `st` then waits (stream-level, the host does not block) for that slot's last
// launch.  Events cost GPU time between launches (~4 us each), so a context records them only once it has seen more than half the
// table's worth of distinct streams (a host that churns streams); slots last used before that have no event: if one of those must be
// recycled the device is synchronised once and the whole table is free again.
static constexpr uint32_t kSlotEventsFrom = 8;

static int32_t take_tile_slot(rt_ctx* c, hipStream_t st, uint32_t& slot_out)
{
    uint32_t slot = kTileSlots;
    for (uint32_t i = 0; i < c->tile_slots_used; ++i)
        if (c->tile_slot_stream[i] == st) { slot = i; break; }
    if (slot == kTileSlots) {
        if (c->tile_slots_used < kTileSlots) {
            slot = c->tile_slots_used++;
            if (c->tile_slots_used > kSlotEventsFrom) c->slot_events_on = true;
        } else {
            uint32_t lru = 0;
            bool unknown = false;
            for (uint32_t i = 0; i < kTileSlots && slot == kTileSlots; ++i) {
                if (!c->tile_slot_ev_valid[i]) { unknown = true; continue; }
                const hipError_t q = hipEventQuery(c->tile_slot_ev[i]);
                if (q == hipSuccess) slot = i;
                else if (q != hipErrorNotReady) return fail_hip(c, q, "hipEventQuery(tile slot)");
                else (void)hipGetLastError();  // "not ready" is an answer, not an error: keep it out of the launch's hipGetLastError()
                if (c->tile_slot_use[i] < c->tile_slot_use[lru]) lru = i;
            }
            if (slot == kTileSlots && unknown) {
                RT_HIP(c, hipDeviceSynchronize());  // once: slots whose last launch predates the events; everything is idle afterwards
                for (uint32_t i = 0; i < kTileSlots; ++i) {
                    c->tile_slot_stream[i] = nullptr;
                    c->tile_slot_ev_valid[i] = false;
                }
                c->tile_slots_used = 1;
                slot = 0;
            } else if (slot == kTileSlots) {
                slot = c->tile_slot_ev_valid[lru] ? lru : 0;
                RT_HIP(c, hipStreamWaitEvent(st, c->tile_slot_ev[slot], 0));
            }
        }
        c->tile_slot_stream[slot] = st;
        c->tile_slot_ev_valid[slot] = false;
    }
    c->tile_slot_use[slot] = ++c->use_clock;
    slot_out = slot;
    return RT_OK;
}

// Enqueue the kernels of the band [y0, y1) on `st` (arguments already checked, ctx's device current).
static int32_t enqueue_rows(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                            void* d_out, void* d_hit_idx, hipStream_t st, ScratchKind kind = ScratchKind::kContext, bool time_it = true)
{
    RenderArgs a;
    std::memcpy(a.m, m, sizeof(a.m));
    a.w = w; a.h = h; a.y0 = y0; a.y1 = y1; a.depth = depth; a.flags = flags;
    a.n = c->n; a.n_pad = c->n_pad; a.n_always = c->n_always; a.n_lights = c->n_lights;
    a.filt = c->d_filt; a.sph = c->d_sph; a.sph_r = c->d_sph_r; a.gen_o = c->d_gen_o; a.gcl = c->d_gcl; a.refl = c->d_refl; a.always = c->d_always; a.lpos = c->d_lpos; a.lcol = c->d_lcol; a.lfilt = c->d_lfilt; a.lclA = c->d_lclA; a.lclB = c->d_lclB; a.order = c->d_order; a.sph_o = c->d_sph_o; a.rank = c->d_rank; a.lalways = c->d_lalways; a.lalways_off = c->d_lalways_off; a.porder = c->d_porder; a.pfilt_o = c->d_pfilt_o; a.pclA = c->d_pclA; a.pclB = c->d_pclB; a.pln = c->d_pln; a.planes = c->d_planes; a.n_planes = c->n_planes; std::memcpy(a.ball, c->ball, sizeof(a.ball));
    a.out = static_cast<uint8_t*>(d_out);
    a.hit_idx = static_cast<int32_t*>(d_hit_idx);
    a.rec_count = nullptr; a.rec_pix = nullptr; a.rec_idx = nullptr; a.rec_t = nullptr;
    a.strips = a.tiles4 = a.tiles2 = a.tiles1 = a.rows4 = a.rows2 = 0;
    a.rec_cap = 0; a.iter = 0; a.sorted = 0; a.rec_state = nullptr; a.rec_occ = nullptr; a.rec_perm = nullptr; a.rec_hit = nullptr; a.sort_bins = nullptr;
    const bool general = render_needs_records(a);
    const size_t npix = (size_t)(y1 - y0) * w;
    bool need_state = false;
    uint32_t occ_words = 0, sort_planes = 0, bin_words = 0;
    render_scratch_layout(a, need_state, occ_words, sort_planes, bin_words);
    const size_t planes = 3 + (need_state ? 9 : 0) + occ_words + sort_planes;

    uint32_t slot = 0;
    char* rec_base = nullptr;
    size_t rec_cap = 0;
    if (kind == ScratchKind::kStatic) {
        // [tile-queue counters (64 B) | pad to 256 | count (16 B) | planes of npix words | sort bins]; the counters were zeroed when the
        // array's address was resolved and re-arm themselves; the record count, occlusion words and sort bins are cleared by launch_render
        if (general && kNoAllocRecOffset + 16 + npix * planes * 4 + (size_t)bin_words * 4 > (size_t)RT_NOALLOC_SCRATCH_BYTES)
            return fail(c, RT_ERR_TOO_LARGE, "rt_render_rows_device_noalloc: this band needs " + std::to_string(kNoAllocRecOffset + 16 + npix * planes * 4 + (size_t)bin_words * 4) +
                                                 " bytes of hit-record scratch, the fixed array holds " + std::to_string((size_t)RT_NOALLOC_SCRATCH_BYTES));
        a.tile_ctr = reinterpret_cast<uint32_t*>(c->noalloc_scratch);
        rec_base = reinterpret_cast<char*>(c->noalloc_scratch) + kNoAllocRecOffset;
        rec_cap = npix;
    } else {
        const int32_t rc = take_tile_slot(c, st, slot);
        if (rc != RT_OK) return rc;
        a.tile_ctr = c->d_tile_ctr + (size_t)slot * kTileSlotWords;
        if (general) {
            // scratch: [count (16 B) | pix | idx | t | 9 state planes (depth > 1) | occlusion planes (RT_SHADOWS)], planes of rec_capacity words
            if (npix > c->rec_capacity || planes > c->rec_planes || bin_words > c->rec_bin_words) {  // grow (the previous launches that used it must be finished first)
                const size_t cap = npix > c->rec_capacity ? npix : c->rec_capacity;
                const size_t pl = planes > c->rec_planes ? planes : c->rec_planes;
                const size_t bw = bin_words > c->rec_bin_words ? bin_words : c->rec_bin_words;
                RT_HIP(c, hipDeviceSynchronize());
                if (c->d_rec) RT_HIP(c, hipFree(c->d_rec));
                c->d_rec = nullptr;
                c->rec_capacity = 0;
                c->rec_planes = 0;
                c->rec_bin_words = 0;
                RT_HIP(c, hipMalloc(&c->d_rec, 16 + cap * pl * 4 + bw * 4));
                c->rec_capacity = cap;
                c->rec_planes = pl;
                c->rec_bin_words = bw;
            }
            rec_base = static_cast<char*>(c->d_rec);
            rec_cap = c->rec_capacity;
            // the records are one per context: a general launch on another stream than the previous one waits for it (stream-level)
            if (c->general_pending && c->general_stream != st) RT_HIP(c, hipStreamWaitEvent(st, c->ev_general, 0));
        }
    }
    if (general) {
        const size_t plane = rec_cap * 4;
        a.rec_cap = (uint32_t)rec_cap;
        a.rec_count = reinterpret_cast<uint32_t*>(rec_base);
        a.rec_pix = reinterpret_cast<uint32_t*>(rec_base + 16);
        a.rec_idx = reinterpret_cast<int32_t*>(rec_base + 16 + plane);
        a.rec_t = reinterpret_cast<float*>(rec_base + 16 + 2 * plane);
        if (need_state) a.rec_state = reinterpret_cast<float*>(rec_base + 16 + 3 * plane);
        if (occ_words) a.rec_occ = reinterpret_cast<uint32_t*>(rec_base + 16 + (3 + (need_state ? 9 : 0)) * plane);
        if (sort_planes) {
            a.rec_perm = reinterpret_cast<uint32_t*>(rec_base + 16 + (3 + (need_state ? 9 : 0) + occ_words) * plane);
            a.rec_hit = reinterpret_cast<float*>(rec_base + 16 + (3 + (need_state ? 9 : 0) + occ_words + 1) * plane);
            // the bins sit behind ALL planes of the layout in use (context scratch: behind its allocated planes)
            a.sort_bins = reinterpret_cast<uint32_t*>(rec_base + 16 + (kind == ScratchKind::kStatic ? planes : c->rec_planes) * plane);
        }
    }
    const bool time_phases = general && kind == ScratchKind::kContext && time_it;
    if (kind == ScratchKind::kContext) c->phases_valid = false;
    RT_HIP(c, launch_render(a, st, time_phases ? &c->phases : nullptr));
    if (kind == ScratchKind::kContext) {
        c->phases_valid = time_phases && c->phases.iters > 0;
        if (c->slot_events_on) {
            RT_HIP(c, hipEventRecord(c->tile_slot_ev[slot], st));
            c->tile_slot_ev_valid[slot] = true;
        }
        if (general) {
            RT_HIP(c, hipEventRecord(c->ev_general, st));
            c->general_pending = true;
            c->general_stream = st;
        }
    }
    return RT_OK;
}

// Multi-device context: cut [y0, y1) into one contiguous band per entry of the device list (global row indices in the ray formula,
// display.rs:99), render band 0 in place in the caller's frame on the root device and every other band into its part's staging, on
// that part's own stream; then ONE RCCL group gathers the staged bands into their rows of the frame (rt_multi.cpp).  `st_root` is the
// stream the caller waits on: the receives are enqueued on it, so "st_root idle" means the whole frame has landed.
// The band plan (rows, staging offsets, launch order grouped by device) is cached in the context and rebuilt only when the geometry of
// the call changes, every device is made current once per call, and nothing is allocated per call (profiles/r03_multi_ctx_probe.txt).
static void multi_plan(rt_ctx* c, uint32_t w, uint32_t y0, uint32_t y1, size_t bpp, bool hit)
{
    MultiPlan& mp = c->plan;
    if (mp.valid && mp.w == w && mp.y0 == y0 && mp.y1 == y1 && mp.bpp == bpp && mp.hit == hit) return;
    const uint32_t G = (uint32_t)c->band_part.size(), rows = y1 - y0;
    mp.bands.assign(G, MultiPlan::Band());
    mp.need.assign(c->parts.size(), 0);
    for (uint32_t g = 0; g < G; ++g) {
        MultiPlan::Band& b = mp.bands[g];
        b.b0 = y0 + (uint32_t)((uint64_t)rows * g / G);
        b.b1 = y0 + (uint32_t)((uint64_t)rows * (g + 1) / G);
        b.part = c->band_part[g];
        b.off = b.hit_off = 0;
        if (g > 0) {  // staged: [pixels | pad to 256 | hit indices | pad to 256] (the hit plane is written with 4-byte stores and sent by RCCL)
            const size_t npix = (size_t)(b.b1 - b.b0) * w;
            b.off = mp.need[(size_t)b.part];
            b.hit_off = b.off + (npix * bpp + 255) / 256 * 256;
            mp.need[(size_t)b.part] = b.hit_off + (hit ? (npix * 4 + 255) / 256 * 256 : 0);
        }
    }
    mp.order.clear();  // launch order: part-major, so that every device is made current once
    for (size_t p = 0; p < c->parts.size(); ++p)
        for (uint32_t g = 0; g < G; ++g)
            if ((size_t)mp.bands[g].part == p && mp.bands[g].b1 > mp.bands[g].b0) mp.order.push_back(g);
    mp.xfers.reserve(2 * (size_t)G);
    mp.w = w; mp.y0 = y0; mp.y1 = y1; mp.bpp = bpp; mp.hit = hit;
    mp.valid = true;
}

static int32_t multi_enqueue(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                             void* d_out, void* d_hit_idx, hipStream_t st_root)
{
    const size_t bpp = (flags & RT_OUT_RGB8) ? 3 : 4;
    multi_plan(c, w, y0, y1, bpp, d_hit_idx != nullptr);
    MultiPlan& mp = c->plan;
    for (size_t p = 0; p < c->parts.size(); ++p) {
        rt_ctx* part = c->parts[p];
        if (mp.need[p] <= part->d_band_bytes) continue;
        DeviceGuard guard(part->device);
        RT_HIP(c, guard.status);
        RT_HIP(c, hipDeviceSynchronize());  // earlier frames may still be sending out of the old staging
        if (part->d_band) RT_HIP(c, hipFree(part->d_band));
        part->d_band = nullptr;
        part->d_band_bytes = 0;
        RT_HIP(c, hipMalloc(&part->d_band, mp.need[p]));
        part->d_band_bytes = mp.need[p];
    }
    // The root device's staging is reused frame after frame and its bands run on the CALLER's stream: when that stream differs from the
    // previous call's, order this call after everything the previous stream holds (an event recorded on it now, a stream-level wait
    // here), or the render of this frame could overwrite staging the previous frame is still sending from.  Nothing is recorded while
    // the caller stays on one stream.  (The previous stream must still exist: see rt_render_rows_device in the header.)
    if (c->last_root_stream_set && c->last_root_stream != st_root && c->parts[0]->d_band) {
        if (hipEventRecord(c->ev_stream_switch, c->last_root_stream) == hipSuccess) RT_HIP(c, hipStreamWaitEvent(st_root, c->ev_stream_switch, 0));
        else (void)hipGetLastError();
    }
    c->last_root_stream = st_root;
    c->last_root_stream_set = true;
    mp.xfers.clear();
    int current = c->device;  // the entry point made the root device current (and restores the caller's when it returns)
    for (const uint32_t g : mp.order) {
        const MultiPlan::Band& b = mp.bands[g];
        rt_ctx* part = c->parts[(size_t)b.part];
        if (part->device != current) {
            RT_HIP(c, hipSetDevice(part->device));
            current = part->device;
        }
        const size_t npix = (size_t)(b.b1 - b.b0) * w, dst_off = (size_t)(b.b0 - y0) * w;
        // Bands of the root device run on the caller's stream (band 0 in place, repeated entries of the root device into its staging);
        // every other device's bands on that part's own stream.
        hipStream_t st = b.part == 0 ? st_root : part->stream;
        char* out = g == 0 ? static_cast<char*>(d_out) : static_cast<char*>(part->d_band) + b.off;
        char* hit = !d_hit_idx ? nullptr : (g == 0 ? static_cast<char*>(d_hit_idx) : static_cast<char*>(part->d_band) + b.hit_off);
        const int32_t rc = enqueue_rows(part, m, w, h, b.b0, b.b1, depth, flags, out, hit, st, ScratchKind::kContext, false);
        if (rc != RT_OK) {
            if (current != c->device) (void)hipSetDevice(c->device);
            return fail(c, rc, part->err);
        }
        if (g > 0) {
            mp.xfers.push_back(BandXfer{b.part, out, static_cast<char*>(d_out) + dst_off * bpp, npix * bpp, st});
            if (hit) mp.xfers.push_back(BandXfer{b.part, hit, static_cast<char*>(d_hit_idx) + dst_off * 4, npix * 4, st});
        }
    }
    if (current != c->device) RT_HIP(c, hipSetDevice(c->device));
    const std::string msg = c->gather.gather(mp.xfers, st_root);
    if (!msg.empty()) return fail(c, RT_ERR_RCCL, "band gather: " + msg);
    return RT_OK;
}

extern "C" {

int32_t rt_render_rows_device(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                              void* d_out, void* d_hit_idx, void* stream)
{
    int32_t rc = check_render_args(c, m, w, h, y0, y1, depth, flags);
    if (rc != RT_OK) return rc;
    if (y1 > y0 && !d_out) return fail(c, RT_ERR_INVALID, "output pointer is null");
    RT_DEVICE(c);
    const bool time_it = (flags & RT_TIMED) != 0u;
    flags &= ~RT_TIMED;
    if (c->multi) {
        hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->parts[0]->stream;
        if (time_it) RT_HIP(c, hipEventRecord(c->ev0, st));
        if (y1 > y0 && (rc = multi_enqueue(c, m, w, h, y0, y1, depth, flags, d_out, d_hit_idx, st)) != RT_OK) return rc;
        if (time_it) RT_HIP(c, hipEventRecord(c->ev_multi_end, st));
        c->ev_end = c->ev_multi_end;
        c->timed = time_it;
        return RT_OK;
    }
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    if (time_it) RT_HIP(c, hipEventRecord(c->ev0, st));
    rc = enqueue_rows(c, m, w, h, y0, y1, depth, flags, d_out, d_hit_idx, st, ScratchKind::kContext, time_it);
    if (rc != RT_OK) return rc;
    if (time_it) RT_HIP(c, hipEventRecord(c->ev_dev_end, st));
    c->ev_end = c->ev_dev_end;
    c->timed = time_it;
    return RT_OK;
}

int32_t rt_render_rows_device_noalloc(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                                      void* d_out, void* d_hit_idx, void* stream)
{
    int32_t rc = check_render_args(c, m, w, h, y0, y1, depth, flags);
    if (rc != RT_OK) return rc;
    if (y1 > y0 && !d_out) return fail(c, RT_ERR_INVALID, "output pointer is null");
    if (c->multi) return fail(c, RT_ERR_INVALID, "rt_render_rows_device_noalloc works on single-device contexts (a multi-device context stages bands in allocated buffers)");
    if (!c->noalloc_scratch) return fail(c, RT_ERR_HIP, "the scratch array of this device was not resolved at rt_ctx_create");
    RT_DEVICE(c);
    hipStream_t st = stream ? static_cast<hipStream_t>(stream) : c->stream;
    return enqueue_rows(c, m, w, h, y0, y1, depth, flags & ~RT_TIMED, d_out, d_hit_idx, st, ScratchKind::kStatic, false);
}

int32_t rt_render_rows(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1, uint32_t depth, uint32_t flags,
                       uint8_t* out_host, int32_t* hit_idx_host)
{
    int32_t rc = check_render_args(c, m, w, h, y0, y1, depth, flags);
    if (rc != RT_OK) return rc;
    if (y1 == y0) return RT_OK;
    if (!out_host) return fail(c, RT_ERR_INVALID, "output pointer is null");
    flags &= ~RT_TIMED;  // the host-output entry points always time their kernels (two events per blocking call)
    RT_DEVICE(c);
    const size_t npix = (size_t)(y1 - y0) * w;
    const size_t bytes = npix * ((flags & RT_OUT_RGB8) ? 3 : 4);
    if (c->multi) {
        // the frame is assembled on the root device (bands rendered on every device of the list, gathered over RCCL), then copied out
        rt_ctx* root = c->parts[0];
        const size_t total = (bytes + 255) / 256 * 256 + (hit_idx_host ? npix * 4 : 0);
        if (total > root->d_out_bytes) {
            RT_HIP(c, hipStreamSynchronize(root->stream));
            if (root->d_out) RT_HIP(c, hipFree(root->d_out));
            root->d_out = nullptr;
            root->d_out_bytes = 0;
            RT_HIP(c, hipMalloc(&root->d_out, total));
            root->d_out_bytes = total;
        }
        char* d_pix = static_cast<char*>(root->d_out);
        char* d_hit = hit_idx_host ? d_pix + (bytes + 255) / 256 * 256 : nullptr;
        RT_HIP(c, hipEventRecord(c->ev0, root->stream));
        rc = multi_enqueue(c, m, w, h, y0, y1, depth, flags, d_pix, d_hit, root->stream);
        if (rc != RT_OK) {
            for (rt_ctx* p : c->parts) {  // nothing of this call may still be in flight when the caller gets its buffers back
                DeviceGuard g(p->device);
                (void)hipDeviceSynchronize();
            }
            return rc;
        }
        RT_HIP(c, hipEventRecord(c->ev_multi_end, root->stream));
        c->ev_end = c->ev_multi_end;
        c->timed = true;
        RT_HIP(c, hipMemcpyAsync(out_host, d_pix, bytes, hipMemcpyDeviceToHost, root->stream));
        if (hit_idx_host) RT_HIP(c, hipMemcpyAsync(hit_idx_host, d_hit, npix * 4, hipMemcpyDeviceToHost, root->stream));
        RT_HIP(c, hipStreamSynchronize(root->stream));
        return RT_OK;
    }
    if (bytes > c->d_out_bytes) {
        if (c->d_out) RT_HIP(c, hipFree(c->d_out));
        c->d_out = nullptr;
        c->d_out_bytes = 0;
        RT_HIP(c, hipMalloc(&c->d_out, bytes));
        c->d_out_bytes = bytes;
    }
    if (hit_idx_host && npix * 4 > c->d_hit_bytes) {
        if (c->d_hit) RT_HIP(c, hipFree(c->d_hit));
        c->d_hit = nullptr;
        c->d_hit_bytes = 0;
        RT_HIP(c, hipMalloc(&c->d_hit, npix * 4));
        c->d_hit_bytes = npix * 4;
    }
    // A large band goes out in row chunks: the copy of chunk k to the host (copy stream) overlaps the render of the chunks after it.
    // The chunks are rendered on two alternating streams, so the drain of one launch overlaps the ramp of the next (a chunk alone on
    // a stream pays ~34 us of launch ramp and drain, profiles/r02_tuning.md).  Chunks are row bands: the image is the single-launch
    // image byte for byte.  General-path chunks are ordered by the library anyway (shared hit records).
    const size_t bpp = (flags & RT_OUT_RGB8) ? 3 : 4;
    const uint32_t rows = y1 - y0;
    const uint32_t n_chunks = npix >= (2u << 20) ? kHostChunks : 1u;
    // two render streams only pay when the copies are asynchronous, i.e. into page-locked memory (rt_host_alloc or the caller's own
    // hipHostMalloc / hipHostRegister); a pageable destination makes every copy block in the driver and measured 7 % slower with two
    bool two_streams = false;
    if (n_chunks > 1) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, out_host) == hipSuccess) two_streams = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();  // ordinary memory is "invalid value" to the runtime: not an error of this call
    }
    hipStream_t rs[2] = {c->stream, two_streams ? c->stream2 : c->stream};
    RT_HIP(c, hipEventRecord(c->ev0, c->stream));
    if (two_streams) RT_HIP(c, hipStreamWaitEvent(c->stream2, c->ev0, 0));
    auto drain = [&]() {
        (void)hipStreamSynchronize(c->copy_stream);
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamSynchronize(c->stream);
    };
    uint32_t last_of[2] = {n_chunks, n_chunks};
    for (uint32_t k = 0; k < n_chunks; ++k) {
        const uint32_t r0 = (uint32_t)((uint64_t)rows * k / n_chunks), r1 = (uint32_t)((uint64_t)rows * (k + 1) / n_chunks);
        if (r1 == r0) continue;
        const size_t off = (size_t)r0 * w;
        hipStream_t st = rs[k & 1u];
        rc = enqueue_rows(c, m, w, h, y0 + r0, y0 + r1, depth, flags, static_cast<char*>(c->d_out) + off * bpp,
                          hit_idx_host ? static_cast<char*>(c->d_hit) + off * 4 : nullptr, st);
        if (rc != RT_OK) {  // copies of earlier chunks may still be writing the caller's buffer: let them finish before handing it back
            drain();
            return rc;
        }
        hipError_t e;
        if ((e = hipEventRecord(c->ev_chunk[k], st)) != hipSuccess || (e = hipStreamWaitEvent(c->copy_stream, c->ev_chunk[k], 0)) != hipSuccess ||
            (e = hipMemcpyAsync(out_host + off * bpp, static_cast<char*>(c->d_out) + off * bpp, (size_t)(r1 - r0) * w * bpp, hipMemcpyDeviceToHost, c->copy_stream)) != hipSuccess ||
            (hit_idx_host && (e = hipMemcpyAsync(hit_idx_host + off, static_cast<char*>(c->d_hit) + off * 4, (size_t)(r1 - r0) * w * 4, hipMemcpyDeviceToHost, c->copy_stream)) != hipSuccess)) {
            drain();
            return fail_hip(c, e, "chunked copy to the host");
        }
        last_of[k & 1u] = k;
    }
    // end of the kernels of this call = both render streams have finished: the timing event sits on the ctx stream after a wait for the other
    if (two_streams && last_of[1] < n_chunks) RT_HIP(c, hipStreamWaitEvent(c->stream, c->ev_chunk[last_of[1]], 0));
    RT_HIP(c, hipEventRecord(c->ev_host_end, c->stream));
    c->ev_end = c->ev_host_end;
    c->timed = true;
    RT_HIP(c, hipStreamSynchronize(c->copy_stream));
    RT_HIP(c, hipStreamSynchronize(c->stream2));
    RT_HIP(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

int32_t rt_render(rt_ctx* c, const float m[16], uint32_t w, uint32_t h, uint32_t depth, uint32_t flags, uint8_t* out_host)
{
    return rt_render_rows(c, m, w, h, 0, h, depth, flags, out_host, nullptr);
}

int32_t rt_host_alloc(rt_ctx* c, size_t bytes, void** out)
{
    if (!c) return RT_ERR_INVALID;
    if (!out || bytes == 0) return fail(c, RT_ERR_INVALID, "rt_host_alloc: empty request");
    *out = nullptr;
    RT_DEVICE(c);  // (a multi-device context allocates with its root device current)
    RT_HIP(c, hipHostMalloc(out, bytes, hipHostMallocDefault));
    return RT_OK;
}

int32_t rt_host_free(rt_ctx* c, void* p)
{
    if (!c) return RT_ERR_INVALID;
    if (!p) return RT_OK;
    RT_DEVICE(c);
    if (c->multi) RT_HIP(c, hipStreamSynchronize(c->parts[0]->stream));
    else RT_HIP(c, hipStreamSynchronize(c->copy_stream));  // no copy of this ctx may still target the block
    RT_HIP(c, hipHostFree(p));
    return RT_OK;
}

int32_t rt_last_phase_ms(rt_ctx* c, float ms[3])
{
    if (!c || !ms) return RT_ERR_INVALID;
    if (!c->phases_valid) return fail(c, RT_ERR_INVALID, "the most recent launch of this context was not a timed general one (depth > 1 or RT_SHADOWS; rt_render_rows_device needs RT_TIMED)");
    RT_DEVICE(c);
    const PhaseEvents& p = c->phases;
    RT_HIP(c, hipEventSynchronize(p.ev[1 + 2 * p.iters]));
    float t = 0.0f;
    RT_HIP(c, hipEventElapsedTime(&t, p.ev[0], p.ev[1]));
    ms[0] = t; ms[1] = 0.0f; ms[2] = 0.0f;
    for (uint32_t it = 0; it < p.iters; ++it) {
        RT_HIP(c, hipEventElapsedTime(&t, p.ev[1 + 2 * it], p.ev[2 + 2 * it]));
        ms[1] += t;
        RT_HIP(c, hipEventElapsedTime(&t, p.ev[2 + 2 * it], p.ev[3 + 2 * it]));
        ms[2] += t;
    }
    return RT_OK;
}

int32_t rt_last_kernel_ms(rt_ctx* c, float* ms)
{
    if (!c || !ms) return RT_ERR_INVALID;
    if (!c->timed) return fail(c, RT_ERR_INVALID, "the most recent launch of this context was not timed (rt_render_rows_device needs RT_TIMED)");
    RT_DEVICE(c);
    RT_HIP(c, hipEventSynchronize(c->ev_end));
    RT_HIP(c, hipEventElapsedTime(ms, c->ev0, c->ev_end));
    return RT_OK;
}

}  // extern "C"
