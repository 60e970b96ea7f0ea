/*
 * rt_oracle.c — CPU ORACLE (test infrastructure, NOT product code). See rt_oracle.h.
 *
 * Build (oracle/Makefile): gcc -std=c11 -O2 -ffp-contract=off -fno-fast-math -msse2 -mfpmath=sse -shared -fPIC
 * Every arithmetic statement is one IEEE binary32 operation in the order the reference performs it;
 * -ffp-contract=off keeps gcc from fusing a*b+c.  Citations are file:line under /root/reference/src.
 */
#define _GNU_SOURCE
#include "rt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdatomic.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

typedef struct { float x, y, z; } v3;

/* math/vec3.rs:153-155 dot = (self.0 * rhs.0).reduce_add(); reduce_add (vec3.rs:118-150, SSE path) = (l0+l1)+(l2+l3), l3 == 0*0 */
static inline float dot3(v3 a, v3 b)
{
    float p0 = a.x * b.x;
    float p1 = a.y * b.y;
    float p2 = a.z * b.z;
    float p3 = 0.0f; /* lane 3 is 0 by construction (vec3.rs:28-30) */
    float lo = p0 + p1;
    float hi = p2 + p3;
    return lo + hi;
}
static inline v3 sub3(v3 a, v3 b) { return (v3){a.x - b.x, a.y - b.y, a.z - b.z}; }  /* vec3.rs:433-440 */
static inline v3 add3(v3 a, v3 b) { return (v3){a.x + b.x, a.y + b.y, a.z + b.z}; }  /* vec3.rs:417-424 */
static inline v3 scale3(float s, v3 a) { return (v3){s * a.x, s * a.y, s * a.z}; }    /* vec3.rs:467-474 */
static inline v3 mul3(v3 a, v3 b) { return (v3){a.x * b.x, a.y * b.y, a.z * b.z}; }   /* vec3.rs:90-92 wide_mul */
/* vec3.rs:190-197: norm = sqrt(sq_norm); unit = self / norm ; Div<f32> (vec3.rs:483-490) is a true per-lane division */
static inline v3 unit3(v3 a)
{
    float n = sqrtf(dot3(a, a));
    return (v3){a.x / n, a.y / n, a.z / n};
}

float rto_dot3(const float a[3], const float b[3]) { return dot3((v3){a[0], a[1], a[2]}, (v3){b[0], b[1], b[2]}); }
void rto_unit3(const float v[3], float out[3])
{
    v3 u = unit3((v3){v[0], v[1], v[2]});
    out[0] = u.x; out[1] = u.y; out[2] = u.z;
}

/* display.rs:121 — f32::to_radians(60.0) == 60.0 * (PI_f32 / 180.0) */
float rto_default_fov(void)
{
    const float pi = 3.14159265358979323846f;
    float rads_per_deg = pi / 180.0f;
    return 60.0f * rads_per_deg;
}

/* display.rs:30-41 */
void rto_camera_transform(float fov, float z_near, float z_far, float aspect, float m[16])
{
    float yy = 1.0f / tanf(fov / 2.0f); /* f32::tan(..).recip() */
    float zm = z_far - z_near;
    float zp = z_far + z_near;
    memset(m, 0, 16 * sizeof(float));
    m[0] = yy / aspect;
    m[5] = yy;
    m[10] = -zp / zm;
    m[11] = ((-2.0f * z_far) * z_near) / zm;
    m[14] = -1.0f;
}

/* quat.rs:201-226 — Hamilton product in the reference's lane order: (t0 - t3) + ((t1 + t2) * (-1, 1, 1, 1)), quaternions as (r, i, j, k) */
static void quat_mul(const float a[4], const float b[4], float out[4])
{
    const float t1[4] = {a[1] * b[1], a[1] * b[0], a[2] * b[0], a[3] * b[0]};
    const float t2[4] = {a[2] * b[2], a[2] * b[3], a[3] * b[1], a[1] * b[2]};
    const float t3[4] = {a[3] * b[3], a[3] * b[2], a[1] * b[3], a[2] * b[1]};
    const float sg[4] = {-1.0f, 1.0f, 1.0f, 1.0f};
    for (int k = 0; k < 4; ++k) {
        float t12 = t1[k] + t2[k];
        float t12m = t12 * sg[k];
        float t0 = a[0] * b[k];
        float t03 = t0 - t3[k];
        out[k] = t03 + t12m;
    }
}

/* math/mod.rs:40-43 Transform::apply = position + rotation.apply(v).wide_mul(scale); quat.rs:159-166 Versor::apply = (q * (0, v)) * conj(q) */
void rto_transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3])
{
    const float p[4] = {0.0f, v[0], v[1], v[2]};
    const float inv[4] = {versor[0] * 1.0f, versor[1] * -1.0f, versor[2] * -1.0f, versor[3] * -1.0f}; /* quat.rs:145-147 */
    float qp[4], res[4];
    quat_mul(versor, p, qp);
    quat_mul(qp, inv, res);
    for (int k = 0; k < 3; ++k) {
        float scaled = res[k + 1] * scale[k];
        out[k] = position[k] + scaled;
    }
}

/* vec4.rs:119-121 dot ; vec4.rs:84-116 reduce_add = (p0+p1)+(p2+p3) */
static inline float dot4(const float* r, const float* v)
{
    float p0 = r[0] * v[0], p1 = r[1] * v[1], p2 = r[2] * v[2], p3 = r[3] * v[3];
    float lo = p0 + p1;
    float hi = p2 + p3;
    return lo + hi;
}

/* mat4.rs:180-192 */
void rto_mat4_mul_vec4(const float m[16], const float v[4], float out[4])
{
    for (int r = 0; r < 4; ++r) out[r] = dot4(m + 4 * r, v);
}

/* display.rs:98-100: transform * (2. * Vec4(j, i, 1, 1).wide_div(size) - splat(1.)) ; vec3.rs:702-707 drops w */
static inline v3 pixel_position(const float m[16], uint32_t j, uint32_t i, uint32_t w, uint32_t h)
{
    float v[4];
    v[0] = 2.0f * ((float)j / (float)w) - 1.0f;
    v[1] = 2.0f * ((float)i / (float)h) - 1.0f;
    v[2] = 2.0f * (1.0f / 1.0f) - 1.0f;
    v[3] = 2.0f * (1.0f / 1.0f) - 1.0f;
    return (v3){dot4(m + 0, v), dot4(m + 4, v), dot4(m + 8, v)};
}
void rto_pixel_position(const float m[16], uint32_t j, uint32_t i, uint32_t w, uint32_t h, float pos[3])
{
    v3 p = pixel_position(m, j, i, w, h);
    pos[0] = p.x; pos[1] = p.y; pos[2] = p.z;
}

/* object/sphere.rs:26-44 */
static inline int sphere_is_hit_by(v3 o, v3 d, v3 c, float r, float* t)
{
    v3 dist = sub3(o, c);                                  /* :27 */
    float alpha = dot3(d, dist);                           /* :28 */
    float delta = (alpha * alpha) - (dot3(dist, dist) - (r * r)); /* :29 */
    alpha = -alpha;                                        /* :31 */
    float beta = sqrtf(delta);                             /* :32 (unconditional; NaN when delta < 0) */
    if (isnan(alpha) || isnan(beta)) return 0;             /* :34 partial_cmp(..)? => None */
    if (alpha == beta) { *t = 0.0f; return 1; }            /* :35 Equal */
    if (alpha > beta) { *t = alpha - beta; return 1; }     /* :36 Greater */
    float x = alpha + beta;                                /* :37 Less */
    if (x < 0.0f) return 0;                                /* :38 */
    *t = x;                                                /* :39 */
    return 1;
}
int rto_sphere_is_hit_by(const float o[3], const float d[3], const float c[3], float r, float* t)
{
    return sphere_is_hit_by((v3){o[0], o[1], o[2]}, (v3){d[0], d[1], d[2]}, (v3){c[0], c[1], c[2]}, r, t);
}

/* EXTENSION: plane primitive (semantics in rt_oracle.h), same operation discipline: one rounding per statement, dot3 order */
static inline int plane_is_hit_by(v3 o, v3 d, v3 p0, v3 n, float* t)
{
    float denom = dot3(d, n);
    float num = dot3(sub3(p0, o), n);
    float x = num / denom;
    if (isnan(x) || isinf(x)) return 0;
    if (x < 0.0f) return 0;
    *t = x;
    return 1;
}
int rto_plane_is_hit_by(const float o[3], const float d[3], const float p0[3], const float n[3], float* t)
{
    return plane_is_hit_by((v3){o[0], o[1], o[2]}, (v3){d[0], d[1], d[2]}, (v3){p0[0], p0[1], p0[2]}, (v3){n[0], n[1], n[2]}, t);
}

/* light/point.rs:25-30 */
static inline int point_hits(v3 lp, v3 lc, float intensity, v3 point, v3* out)
{
    float dist = sqrtf(dot3(sub3(point, lp), sub3(point, lp))); /* point.distance(self.point), vec3.rs:200-202 */
    float in = intensity / (dist * dist);
    if (in <= 1.1920929e-7f) return 0;                          /* f32::EPSILON ; NaN compares false => Some(NaN..) */
    *out = scale3(in, lc);
    return 1;
}
int rto_point_hits(const float lp[3], const float lc[3], float intensity, const float p[3], float out[3])
{
    v3 o;
    int r = point_hits((v3){lp[0], lp[1], lp[2]}, (v3){lc[0], lc[1], lc[2]}, intensity, (v3){p[0], p[1], p[2]}, &o);
    if (r) { out[0] = o.x; out[1] = o.y; out[2] = o.z; }
    return r;
}

/* renderer.rs:76 — Rust `f32 as u8`: NaN -> 0, saturate, truncate toward zero */
uint8_t rto_quantise(float c)
{
    float q = 255.0f * c; /* renderer.rs:75 */
    if (isnan(q)) return 0;
    if (q <= 0.0f) return 0;
    if (q >= 255.0f) return 255;
    return (uint8_t)q;
}

typedef struct {
    const rto_objects* objs;
    const rto_lights* lights;
    const float* m;
    uint32_t w, h, y0, y1, row_step, depth, flags, channels;
    uint8_t* out;
    int32_t* hit_idx;
    atomic_uint next_row;
} job_t;

static inline v3 obj_center(const rto_objects* o, int k) { return (v3){o->cx[k], o->cy[k], o->cz[k]}; }
static inline int obj_is_plane(const rto_objects* o, int k) { return o->kind && o->kind[k] == RTO_KIND_PLANE; }
static inline v3 obj_plane_normal(const rto_objects* o, int k) { return (v3){o->nx[k], o->ny[k], o->nz[k]}; }

/* Object::is_hit_by through the Box<dyn Object> of renderer.rs:48 (object/mod.rs:31-33) */
static inline int object_is_hit_by(const rto_objects* objs, int k, v3 o, v3 d, float* t)
{
    if (obj_is_plane(objs, k)) return plane_is_hit_by(o, d, obj_center(objs, k), obj_plane_normal(objs, k), t);
    return sphere_is_hit_by(o, d, obj_center(objs, k), objs->radius[k], t);
}

/* renderer.rs:47-54 — closest hit; strict `<` so the lowest index wins ties */
static inline int closest_hit(const rto_objects* objs, v3 o, v3 d, float* t_out)
{
    int best = -1;
    float best_t = 0.0f;
    for (int k = 0; k < objs->n; ++k) {
        float t;
        if (!object_is_hit_by(objs, k, o, d, &t)) continue;
        if (best < 0 || t < best_t) { best = k; best_t = t; }
    }
    *t_out = best_t;
    return best;
}

/* EXTENSION (no reference counterpart): is the segment p -> light blocked by any element? */
static inline int occluded(const rto_objects* objs, v3 p, v3 lp)
{
    v3 to_l = sub3(lp, p);
    float dist = sqrtf(dot3(to_l, to_l));
    v3 dir = (v3){to_l.x / dist, to_l.y / dist, to_l.z / dist};
    for (int k = 0; k < objs->n; ++k) {
        float t;
        if (object_is_hit_by(objs, k, p, dir, &t) && t > RTO_SHADOW_EPS && t < dist) return 1;
    }
    return 0;
}

/* renderer.rs:40-77 — the closure body for one pixel */
static inline void render_pixel(const job_t* jb, uint32_t i, uint32_t j, uint8_t* px, int32_t* hit)
{
    const rto_objects* objs = jb->objs;
    const rto_lights* lights = jb->lights;
    v3 pos = pixel_position(jb->m, j, i, jb->w, jb->h);
    v3 color = {0.0f, 0.0f, 0.0f};   /* :42 */
    v3 o = {0.0f, 0.0f, 0.0f};       /* :43 */
    v3 d = unit3(pos);               /* :43 pos.unit() */
    const uint32_t limit = jb->depth - 1; /* :36 */
    int first_hit = -1;

    for (uint32_t it = 0; it < jb->depth; ++it) { /* :46 */
        float t;
        int k = closest_hit(objs, o, d, &t);
        if (it == 0) first_hit = k;
        if (k < 0) break; /* a miss leaves prev_info untouched, so every later iteration misses identically */
        v3 new_origin = add3(o, scale3(t, d)); /* :57, object/mod.rs:61-63 */
        v3 c = color;                          /* :59 */
        for (int l = 0; l < lights->n; ++l) {  /* :60-64, list order */
            uint8_t kind = lights->kind ? lights->kind[l] : RTO_LIGHT_POINT;
            v3 lc = {lights->cr[l], lights->cg[l], lights->cb[l]};
            if (kind == RTO_LIGHT_AMBIENT) { /* light/ambient.rs:23-25 */
                c = add3(c, lc);
            } else {
                v3 lp = {lights->px[l], lights->py[l], lights->pz[l]};
                v3 contrib;
                if (!point_hits(lp, lc, lights->intensity[l], new_origin, &contrib)) continue;
                if ((jb->flags & RTO_FLAG_SHADOWS) && occluded(objs, new_origin, lp)) continue; /* extension */
                c = add3(c, contrib);
            }
        }
        v3 refl = {objs->refl_r[k], objs->refl_g[k], objs->refl_b[k]};
        color = mul3(c, refl); /* :66 */
        if (it < limit) {      /* :67 */
            /* :68 — normal at the OLD origin (reference quirk): sphere.rs:21-23; a plane's normal is the same everywhere */
            v3 normal = obj_is_plane(objs, k) ? obj_plane_normal(objs, k) : unit3(sub3(o, obj_center(objs, k)));
            float dn = dot3(d, normal);                        /* object/mod.rs:72 */
            float two_dn = 2.0f * dn;
            d = unit3(sub3(d, scale3(two_dn, normal)));        /* object/mod.rs:72 */
            o = new_origin;                                    /* :70 */
        }
    }
    px[0] = rto_quantise(color.x); /* :75-76 */
    px[1] = rto_quantise(color.y);
    px[2] = rto_quantise(color.z);
    if (jb->channels == 4) px[3] = 255;
    if (hit) *hit = first_hit;
}

/* display.rs:90-103 — one task per image row (rayon par_chunks_exact_mut(width)) */
static void* worker(void* arg)
{
    job_t* jb = (job_t*)arg;
    for (;;) {
        uint32_t r = atomic_fetch_add(&jb->next_row, 1u);
        uint64_t i64 = (uint64_t)jb->y0 + (uint64_t)r * jb->row_step;
        if (i64 >= jb->y1) break;
        uint32_t i = (uint32_t)i64;
        size_t row_off = (size_t)(i - jb->y0) * jb->w;
        for (uint32_t j = 0; j < jb->w; ++j)
            render_pixel(jb, i, j, jb->out + (row_off + j) * jb->channels, jb->hit_idx ? jb->hit_idx + row_off + j : NULL);
    }
    return NULL;
}

int rto_default_threads(void)
{
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) {
        int n = CPU_COUNT(&set);
        if (n > 0) return n;
    }
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n > 0 ? (int)n : 1;
}

int rto_render(const rto_objects* objs, const rto_lights* lights, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
               uint32_t row_step, uint32_t depth, uint32_t flags, uint32_t channels, uint8_t* out, int32_t* hit_idx, int nthreads)
{
    if (!objs || !lights || !m || !out || depth == 0 || w == 0 || h == 0 || y0 > y1 || y1 > h || row_step == 0 || (channels != 3 && channels != 4) || objs->n < 0 || lights->n < 0)
        return -1;
    for (int k = 0; k < objs->n; ++k) {
        if (!objs->kind || objs->kind[k] == RTO_KIND_SPHERE) continue;
        if (objs->kind[k] != RTO_KIND_PLANE || !objs->nx || !objs->ny || !objs->nz) return -1;
    }
    job_t jb = {objs, lights, m, w, h, y0, y1, row_step, depth, flags, channels, out, hit_idx, 0};
    if (nthreads <= 0) nthreads = rto_default_threads();
    if (nthreads > 256) nthreads = 256;
    uint32_t nrows = (y1 - y0 + row_step - 1) / row_step;
    if ((uint32_t)nthreads > nrows) nthreads = nrows ? (int)nrows : 1;
    if (nthreads == 1) { worker(&jb); return 0; }
    pthread_t th[256];
    int started = 0;
    for (int t = 0; t < nthreads - 1; ++t)
        if (pthread_create(&th[started], NULL, worker, &jb) == 0) ++started;
    worker(&jb);
    for (int t = 0; t < started; ++t) pthread_join(th[t], NULL);
    return 0;
}
