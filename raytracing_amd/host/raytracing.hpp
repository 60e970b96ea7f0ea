// raytracing.hpp — C++ host-side mirror of the reference's scene / renderer interface, sitting on the C ABI
// (include/rt_mi355x.h).  The reference is Rust and no Rust toolchain exists in this image, so the host side above
// the boundary is C++ with the reference's names, argument meaning and error behaviour:
//
//   math/vec3.rs            -> society::Vec3
//   object/mod.rs, sphere.rs-> society::Object (trait), society::Sphere
//   light/mod.rs, point.rs, ambient.rs -> society::Light (trait), society::Point, society::Ambient
//   element.rs              -> society::Material, society::Element
//   display.rs              -> society::Camera, society::Framebuffer
//   renderer.rs             -> society::Renderer  (render(depth): pixel loop on the GPU, then display() and clear())
//
// ("society" is the reference's crate name, Cargo.toml:2.)  Header-only; link with librt_mi355x.so.
// Nothing here computes a pixel: the per-pixel path lives in the HIP kernels.  There is no CPU fallback.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "rt_mi355x.h"

namespace society {

// anyhow::Error stand-in: every fallible call of the reference returns anyhow::Result (display.rs:46,113; renderer.rs:32)
struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// math/vec3.rs:22-30
struct Vec3 {
    float x = 0.0f, y = 0.0f, z = 0.0f;
    constexpr Vec3() = default;
    constexpr Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static constexpr Vec3 splat(float v) { return Vec3(v, v, v); }  // vec3.rs:33-35
    static constexpr Vec3 ZERO() { return Vec3(0.0f, 0.0f, 0.0f); } // vec3.rs:25
};
inline Vec3 operator*(float s, Vec3 v) { return Vec3(s * v.x, s * v.y, s * v.z); }  // vec3.rs:467-474

// math/mod.rs:5-51 — Transform: position, rotation (a versor as r, i, j, k; quat.rs), scale.  The reference never attaches one to an
// object; the mirror applies it on the host before flattening (Sphere::transformed).  The arithmetic is the library's
// rt_transform_apply (same lane order as quat.rs:201-226).
struct Transform {
    Vec3 position;
    float rotation[4] = {1.0f, 0.0f, 0.0f, 0.0f};
    Vec3 scale = Vec3(1.0f, 1.0f, 1.0f);
    Vec3 apply(Vec3 v) const  // math/mod.rs:40-43
    {
        const float p[3] = {position.x, position.y, position.z}, s[3] = {scale.x, scale.y, scale.z}, in[3] = {v.x, v.y, v.z};
        float out[3];
        rt_transform_apply(p, rotation, s, in, out);
        return Vec3(out[0], out[1], out[2]);
    }
};

// What a scene object / light hands to the FFI (the Rust adapter's `flatten()`)
struct FlatObject {
    uint8_t kind;
    float cx, cy, cz, radius;
    float nx = 0.0f, ny = 0.0f, nz = 0.0f;  // RT_PLANE only: unit normal
};
struct FlatLight {
    uint8_t kind;
    float px, py, pz, cr, cg, cb, intensity;
};

// object/mod.rs:7-10 — trait Object; `flatten` is the one method the adapter adds (default: not flattenable)
struct Object {
    virtual ~Object() = default;
    virtual bool flatten(FlatObject&) const { return false; }
};
using DynObject = std::unique_ptr<Object>;  // object/mod.rs:5

// object/sphere.rs:6-16
struct Sphere final : Object {
    Vec3 center;
    float radius = 0.0f;
    Sphere() = default;
    Sphere(Vec3 c, float r) : center(c), radius(r) {}
    // The sphere after an object Transform: centre through Transform::apply, radius times the (uniform) scale.
    Sphere transformed(const Transform& t) const
    {
        if (!(t.scale.x == t.scale.y && t.scale.y == t.scale.z)) throw std::invalid_argument("a Sphere can only take a uniform scale");
        return Sphere(t.apply(center), radius * std::fabs(t.scale.x));
    }
    bool flatten(FlatObject& f) const override
    {
        f = FlatObject{(uint8_t)RT_SPHERE, center.x, center.y, center.z, radius, 0.0f, 0.0f, 0.0f};
        return true;
    }
};

// EXTENSION (no reference counterpart): two-sided infinite plane, an `impl Object` in the shape of object/sphere.rs:19-45.
// The normal is normalised once here like vec3.rs:195-197 (v / sqrt((xx + yy) + (zz + 0)), true divisions).
struct Plane final : Object {
    Vec3 point, normal;
    Plane() = default;
    Plane(Vec3 p, Vec3 n) : point(p)
    {
        const float xx = n.x * n.x, yy = n.y * n.y, zz = n.z * n.z;
        const float lo = xx + yy, hi = zz + 0.0f;
        const float len = std::sqrt(lo + hi);
        normal = Vec3(n.x / len, n.y / len, n.z / len);
    }
    bool flatten(FlatObject& f) const override
    {
        f = FlatObject{(uint8_t)RT_PLANE, point.x, point.y, point.z, 0.0f, normal.x, normal.y, normal.z};
        return true;
    }
};

// element.rs:15-19 — only `reflectiveness` is read by the renderer (renderer.rs:66)
struct Material {
    Vec3 color, reflectiveness;
    Material(Vec3 c, Vec3 r) : color(c), reflectiveness(r) {}
};

// element.rs:10-13, Element<DynObject>
struct Element {
    DynObject object;
    Material material;
    // element.rs:38-40
    template <class T>
    static Element new_unzise(T object, Material material) { return Element{std::make_unique<T>(std::move(object)), material}; }
    // HEAD's main.rs:39-46 passes a bare Vec3 as the material (does not type-check); result.png shows it acts as reflectiveness.
    template <class T>
    static Element new_unzise(T object, Vec3 reflectiveness) { return new_unzise(std::move(object), Material(reflectiveness, reflectiveness)); }
};

// light/mod.rs:8-10 — trait Light
struct Light {
    virtual ~Light() = default;
    virtual bool flatten(FlatLight&) const { return false; }
};
using DynLight = std::unique_ptr<Light>;  // light/mod.rs:6

// light/point.rs:5-21
struct Point final : Light {
    Vec3 point, color;
    float intensity = 0.0f;
    Point(Vec3 p, Vec3 c, float i) : point(p), color(c), intensity(i) {}
    static DynLight new_unsize(Vec3 p, Vec3 c, float i) { return std::make_unique<Point>(p, c, i); }
    bool flatten(FlatLight& f) const override
    {
        f = FlatLight{(uint8_t)RT_POINT, point.x, point.y, point.z, color.x, color.y, color.z, intensity};
        return true;
    }
};

// light/ambient.rs:5-19
struct Ambient final : Light {
    Vec3 color;
    explicit Ambient(Vec3 c) : color(c) {}
    static DynLight new_unsize(Vec3 c) { return std::make_unique<Ambient>(c); }
    bool flatten(FlatLight& f) const override
    {
        f = FlatLight{(uint8_t)RT_AMBIENT, 0.0f, 0.0f, 0.0f, color.x, color.y, color.z, 0.0f};
        return true;
    }
};

// display.rs:8-13, 22-42, 118-123
struct Camera {
    float fov, z_near, z_far;
    Camera() { rt_camera_default(&fov, &z_near, &z_far); }  // Default: 60 degrees, 0.01, 1000
    Camera(float fov_, float zn, float zf) : fov(fov_), z_near(zn), z_far(zf) {}
    void transform(float aspect_ratio, float m[16]) const { rt_camera_transform(fov, z_near, z_far, aspect_ratio, m); }
};

namespace detail {
inline uint32_t crc32_update(uint32_t c, const uint8_t* p, size_t n)
{
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t r = i;
            for (int k = 0; k < 8; ++k) r = (r & 1u) ? 0xEDB88320u ^ (r >> 1) : r >> 1;
            table[i] = r;
        }
        init = true;
    }
    for (size_t i = 0; i < n; ++i) c = table[(c ^ p[i]) & 0xFFu] ^ (c >> 8);
    return c;
}
inline void put_be32(std::vector<uint8_t>& v, uint32_t x)
{
    v.push_back(uint8_t(x >> 24));
    v.push_back(uint8_t(x >> 16));
    v.push_back(uint8_t(x >> 8));
    v.push_back(uint8_t(x));
}
inline void png_chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data)
{
    put_be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    uint32_t c = crc32_update(0xFFFFFFFFu, out.data() + start, out.size() - start) ^ 0xFFFFFFFFu;
    put_be32(out, c);
}
// Minimal PNG encoder (8-bit RGB, zlib "stored" blocks): enough for Framebuffer::display; host I/O after the path.
inline std::vector<uint8_t> encode_png_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h)
{
    std::vector<uint8_t> raw;
    raw.reserve((size_t)h * (3 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; ++y) {
        raw.push_back(0);  // filter: none
        raw.insert(raw.end(), rgb + (size_t)y * w * 3, rgb + (size_t)(y + 1) * w * 3);
    }
    std::vector<uint8_t> z = {0x78, 0x01};
    uint32_t a = 1, b = 0;
    for (uint8_t v : raw) {
        a = (a + v) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t pos = 0;
    do {
        size_t n = raw.size() - pos < 65535 ? raw.size() - pos : 65535;
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back(uint8_t(n));
        z.push_back(uint8_t(n >> 8));
        z.push_back(uint8_t(~n));
        z.push_back(uint8_t((~n) >> 8));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        pos += n;
    } while (pos < raw.size());
    put_be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, w);
    put_be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 2, 0, 0, 0});
    png_chunk(out, "IHDR", ihdr);
    png_chunk(out, "IDAT", z);
    png_chunk(out, "IEND", {});
    return out;
}
}  // namespace detail

// display.rs:16-20, 44-116
class Framebuffer {
public:
    Camera camera;
    Framebuffer(uint32_t width, uint32_t height, Camera cam)  // display.rs:46-61
        : camera(cam), width_(width), height_(height), aspect_ratio_((float)width / (float)height), pixels_((size_t)3 * width * height, 0)
    {
    }
    uint32_t width() const { return width_; }
    uint32_t height() const { return height_; }
    float aspect_ratio() const { return aspect_ratio_; }
    void clear() { std::fill(pixels_.begin(), pixels_.end(), 0); }  // display.rs:108-110
    // display.rs:113-115 — always "result.png" in the CWD
    void display(const char* path = "result.png") const
    {
        std::vector<uint8_t> png = detail::encode_png_rgb8(pixels_.data(), width_, height_);
        FILE* f = std::fopen(path, "wb");
        if (!f) throw Error(std::string("cannot open ") + path);
        size_t n = std::fwrite(png.data(), 1, png.size(), f);
        std::fclose(f);
        if (n != png.size()) throw Error(std::string("short write to ") + path);
    }
    std::vector<uint8_t>& pixels() { return pixels_; }  // RGB8, row-major, row 0 = top (display.rs:16-20)
    const std::vector<uint8_t>& pixels() const { return pixels_; }

private:
    uint32_t width_, height_;
    float aspect_ratio_;
    std::vector<uint8_t> pixels_;
};

// renderer.rs:12-83
class Renderer {
public:
    std::vector<Element> elements;  // pub elements: E
    std::vector<DynLight> lights;   // pub lights: L
    std::vector<uint8_t> last_frame;  // copy of the pixels before clear() (the reference only keeps result.png)

    Renderer(Framebuffer frame, std::vector<Element> elements_, std::vector<DynLight> lights_, int32_t device_id = 0)
        : elements(std::move(elements_)), lights(std::move(lights_)), frame_(std::move(frame)), devices_(1, device_id)
    {
    }
    // One row band per entry of `device_ids` (display.rs:90-93's row split across GPUs), gathered to device_ids[0] inside the library.
    Renderer(Framebuffer frame, std::vector<Element> elements_, std::vector<DynLight> lights_, std::vector<int32_t> device_ids)
        : elements(std::move(elements_)), lights(std::move(lights_)), frame_(std::move(frame)), devices_(std::move(device_ids))
    {
    }
    ~Renderer() { rt_ctx_destroy(ctx_); }
    Renderer(const Renderer&) = delete;
    Renderer& operator=(const Renderer&) = delete;

    Framebuffer& frame() { return frame_; }

    // renderer.rs:32-82.  flags: 0 = reference semantics; RT_SHADOWS = extension.
    void render(size_t depth, uint32_t flags = 0, const char* display_path = "result.png")
    {
        if (depth == 0) throw Error("attempt to subtract with overflow: depth - 1 (renderer.rs:36)");
        if (depth > 0xFFFFFFFFull) throw Error("depth too large");
        if (!ctx_) check(devices_.size() == 1 ? rt_ctx_create(&ctx_, devices_[0]) : rt_ctx_create_multi(&ctx_, devices_.data(), (int32_t)devices_.size()), nullptr);
        upload();
        float m[16];
        frame_.camera.transform(frame_.aspect_ratio(), m);
        check(rt_render(ctx_, m, frame_.width(), frame_.height(), (uint32_t)depth, flags | RT_OUT_RGB8, frame_.pixels().data()), ctx_);
        last_frame = frame_.pixels();
        if (display_path) frame_.display(display_path);  // renderer.rs:79
        frame_.clear();                                  // renderer.rs:80
    }

private:
    void upload()
    {
        const size_t n = elements.size(), nl = lights.size();
        std::vector<uint8_t> kind(n), lkind(nl);
        std::vector<float> a[10], b[7];
        for (auto& v : a) v.resize(n);
        for (auto& v : b) v.resize(nl);
        for (size_t i = 0; i < n; ++i) {
            FlatObject f;
            if (!elements[i].object || !elements[i].object->flatten(f)) throw Error("element cannot be flattened for the GPU path");
            kind[i] = f.kind;
            a[0][i] = f.cx; a[1][i] = f.cy; a[2][i] = f.cz; a[3][i] = f.radius;
            const Vec3 r = elements[i].material.reflectiveness;
            a[4][i] = r.x; a[5][i] = r.y; a[6][i] = r.z;
            a[7][i] = f.nx; a[8][i] = f.ny; a[9][i] = f.nz;
        }
        for (size_t i = 0; i < nl; ++i) {
            FlatLight f;
            if (!lights[i] || !lights[i]->flatten(f)) throw Error("light cannot be flattened for the GPU path");
            lkind[i] = f.kind;
            b[0][i] = f.px; b[1][i] = f.py; b[2][i] = f.pz; b[3][i] = f.cr; b[4][i] = f.cg; b[5][i] = f.cb; b[6][i] = f.intensity;
        }
        rt_objects o{(int32_t)n, kind.data(), a[0].data(), a[1].data(), a[2].data(), a[3].data(), a[4].data(), a[5].data(), a[6].data(),
                     a[7].data(), a[8].data(), a[9].data()};
        rt_lights l{(int32_t)nl, lkind.data(), b[0].data(), b[1].data(), b[2].data(), b[3].data(), b[4].data(), b[5].data(), b[6].data()};
        check(rt_scene_upload(ctx_, &o, &l), ctx_);
    }
    static void check(int32_t rc, const rt_ctx* ctx)
    {
        if (rc != RT_OK) throw Error(std::string("rt_mi355x error ") + std::to_string(rc) + ": " + rt_last_error(ctx));
    }

    Framebuffer frame_;
    std::vector<int32_t> devices_;
    rt_ctx* ctx_ = nullptr;
};

}  // namespace society
