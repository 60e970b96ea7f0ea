/*
 * rt_oracle.h — CPU ORACLE (test infrastructure, NOT product code).
 *
 * A scalar IEEE-binary32 restatement of the per-pixel render path of Aandreba/raytracing,
 * following the reference's operation order exactly (no FMA contraction, IEEE div/sqrt).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (raytracing_amd/, include/rt_mi355x.h) never links or calls it.
 *
 * Parity pins (see tests/test_oracle_golden.py):
 *   - tests/golden/result_100x100.rgb  = pixels of the reference's own result.png  (bit-exact, 10000/10000)
 *   - known-answer tests object/mod.rs:83-92, object/sphere.rs:47-57 (t == 4.0), math/mat4.rs:288-291
 * Unpinned by any reference artefact (source reading only): depth>1 bounce, Point::hits EPSILON cut,
 * Equal/Less branches of is_hit_by, saturation, non-square aspect. RTO_SHADOWS and RTO_KIND_PLANE are extensions
 * with no counterpart in the reference (their semantics are DEFINED here).
 */
#ifndef RT_ORACLE_H
#define RT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTO_KIND_SPHERE 0u
#define RTO_KIND_PLANE 1u /* EXTENSION: the reference has one primitive (object/sphere.rs); see rto_plane_is_hit_by */
#define RTO_LIGHT_POINT 0u
#define RTO_LIGHT_AMBIENT 1u

#define RTO_FLAG_SHADOWS 1u /* extension: occlusion test toward every point light */

/* Shadow-ray acceptance window (extension): occluded iff RTO_SHADOW_EPS < t < dist(light). */
#define RTO_SHADOW_EPS 1.0e-3f

typedef struct {
    int32_t n;
    const uint8_t* kind; /* NULL => all spheres */
    const float *cx, *cy, *cz, *radius;
    const float *refl_r, *refl_g, *refl_b;
    const float *nx, *ny, *nz; /* RTO_KIND_PLANE: unit normal, (cx,cy,cz) = a point of the plane, radius unused; may be NULL without planes */
} rto_objects;

typedef struct {
    int32_t n;
    const uint8_t* kind; /* NULL => all point lights */
    const float *px, *py, *pz; /* ignored for ambient */
    const float *cr, *cg, *cb;
    const float* intensity; /* ignored for ambient */
} rto_lights;

/* display.rs:118-123 — Camera::default().fov == 60f32.to_radians() */
float rto_default_fov(void);
/* display.rs:30-41 — Camera::transform, row-major 4x4 */
void rto_camera_transform(float fov, float z_near, float z_far, float aspect, float m[16]);
/* math/mod.rs:40-43 (Transform::apply) over quat.rs:159-166 (Versor::apply) and quat.rs:201-226 (quaternion product); versor = (r, i, j, k).
 * Nothing in the reference attaches a Transform to an object: host-side pre-transform of sphere centres only. */
void rto_transform_apply(const float position[3], const float versor[4], const float scale[3], const float v[3], float out[3]);
/* mat4.rs:180-192 + vec4.rs:84-121 — Mat4 * Vec4 with the SSE reduce order (p0+p1)+(p2+p3) */
void rto_mat4_mul_vec4(const float m[16], const float v[4], float out[4]);
/* vec3.rs:118-155 — dot with (xx+yy)+(zz+0) order */
float rto_dot3(const float a[3], const float b[3]);
/* vec3.rs:195-197 — unit() */
void rto_unit3(const float v[3], float out[3]);
/* object/sphere.rs:26-44 — returns 1 and *t on Some(t), 0 on None */
int rto_sphere_is_hit_by(const float origin[3], const float dir[3], const float center[3], float radius, float* t);
/*
 * EXTENSION (no reference counterpart; an `impl Object` in the shape of object/sphere.rs:19-45): two-sided plane through `point`
 * with unit normal n.  is_hit_by: denom = dot3(d, n); num = dot3(point - origin, n); t = num / denom; Some(t) iff t is finite
 * and t >= 0 (denom == 0, i.e. a parallel ray, gives inf or NaN => None).  normal(at) = n.  Returns 1 and *t on Some(t).
 */
int rto_plane_is_hit_by(const float origin[3], const float dir[3], const float point[3], const float n[3], float* t);
/* light/point.rs:25-30 — returns 1 and out rgb on Some, 0 on None */
int rto_point_hits(const float light_pos[3], const float light_color[3], float intensity, const float point[3], float out[3]);
/* display.rs:98-100 — pixel (j = column, i = row) -> position (w dropped, vec3.rs:702-707) */
void rto_pixel_position(const float m[16], uint32_t j, uint32_t i, uint32_t w, uint32_t h, float pos[3]);
/* renderer.rs:75-76 — (255*c) as u8 (Rust saturating cast) */
uint8_t rto_quantise(float c);

/*
 * renderer.rs:32-82 + display.rs:74-105 — render rows [y0, y1) stepping by row_step (1 = every row) of a w x h
 * frame. `out` holds (y1-y0) rows of w pixels, `channels` (3 = RGB8 like Framebuffer.pixels, 4 = RGBA8, A=255)
 * bytes each; rows skipped by row_step are left untouched. hit_idx (nullable, same geometry, int32) receives the
 * index of the element hit by the PRIMARY ray or -1. nthreads<=0 => all online cores. Returns 0, or -1 on bad args
 * (depth == 0 is rejected: the reference underflows at renderer.rs:36).
 */
int rto_render(const rto_objects* objs, const rto_lights* lights, const float m[16], uint32_t w, uint32_t h, uint32_t y0, uint32_t y1,
               uint32_t row_step, uint32_t depth, uint32_t flags, uint32_t channels, uint8_t* out, int32_t* hit_idx, int nthreads);

/* number of worker threads rto_render would use for nthreads<=0 */
int rto_default_threads(void);

#ifdef __cplusplus
}
#endif
#endif
