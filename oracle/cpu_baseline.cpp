// cpu_baseline.cpp -- the reference's CPU path, timed on the host cores (a REPORTED baseline, not a target).
//
// TEST/BENCH INFRASTRUCTURE (lives under oracle/; never linked into the product).
//
// C++17 restatement of RT_Weekend's tracer -- no Swift toolchain exists here:
//   color() recursion, depth < 50              RT_Weekend/Tracer/main.swift:3-17
//   HittableList linear scan                   RT_Weekend/Tracer/Hittable.swift:22-35
//   Sphere.hitTest                             RT_Weekend/Tracer/Sphere.swift:15-43
//   Lambertian / Metal / Dielectric scatter    RT_Weekend/Tracer/Material.swift:33-118
//   thin-lens Camera.cast                      RT_Weekend/Tracer/Camera.swift:24-60
//   row bands over processorCount workers      RT_Weekend/Tracer/main.swift:77-114
//   gamma 2 + 8-bit quantisation               RT_Weekend/Tracer/main.swift:100-105
// BASELINE config 1 ("Cornell box 400x400x16") needs rectangles, boxes and emitters, which first appear
// in RT_Nextweek; those pieces are restated from
//   Rect / Box                                 RT_Nextweek/Tracer/Rect.swift:7-69, Box.swift:3-33
//   NormalFlipped / Translate / Rotate(.y)     RT_Nextweek/Tracer/Hittable.swift:58-268
//   DiffuseLight, color1 (emission)            RT_Nextweek/Tracer/Material.swift:86-101, Render.swift:327-340
//   cornellBox scene, camera1                  RT_Nextweek/Tracer/Render.swift:171-192,59-81
// randomFloat() = Float(arc4random())/Float(UInt32.max) (Random.swift:3-6) becomes a per-thread PCG32
// (the reference's own pcg_basic.c algorithm) with fixed seeds, one stream per image row, so that the image and the ray
// count are reproducible whatever the number of worker threads (tests/test_cpu_baseline.py).
//
// A "ray" here = one world.hitTest call (primary + scattered), the same unit as the GPU's Scene::hit count.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <string>
#include <thread>
#include <vector>

namespace {

struct Rng {
    uint64_t state = 0, inc = 1;
    void seed(uint64_t initstate, uint64_t initseq) {
        state = 0; inc = (initseq << 1u) | 1u; next(); state += initstate; next();
    }
    uint32_t next() {
        uint64_t old = state;
        state = old * 6364136223846793005ULL + inc;
        uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
        return (xs >> rot) | (xs << ((0u - rot) & 31));
    }
    float uniform() { return (float)next() / (float)UINT32_MAX; }
};
thread_local Rng* t_rng = nullptr;
thread_local uint64_t t_rays = 0;
inline float randomFloat() { return t_rng->uniform(); }

struct Vec3 {
    float x = 0, y = 0, z = 0;
    Vec3() {}
    Vec3(float s) : x(s), y(s), z(s) {}
    Vec3(float a, float b, float c) : x(a), y(b), z(c) {}
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
    float dot(const Vec3& o) const { return x * o.x + y * o.y + z * o.z; }
    float length() const { return std::sqrt(dot(*this)); }
    Vec3 normalize() const { float l = length(); return Vec3(x / l, y / l, z / l); }
    Vec3 cross(const Vec3& o) const { return Vec3(y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x); }
};
inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator*(Vec3 a, float s) { return Vec3(s * a.x, s * a.y, s * a.z); }
inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }

struct Ray {
    Vec3 origin, direction;
    Ray() {}
    Ray(Vec3 o, Vec3 d) : origin(o), direction(d) {}
    Vec3 pointAt(float t) const { return origin + t * direction; }
};

Vec3 randomInUnitSphere() {     // Random.swift:8-16
    Vec3 p;
    do { p = 2.0f * Vec3(randomFloat(), randomFloat(), randomFloat()) - Vec3(1, 1, 1); } while (p.dot(p) >= 1.0f);
    return p;
}
Vec3 randomInUnitDisk() {       // Random.swift:18-26
    Vec3 p;
    do { p = 2.0f * Vec3(randomFloat(), randomFloat(), 0) - Vec3(1, 1, 0); } while (p.dot(p) >= 1.0f);
    return p;
}

struct Material;
struct HitRecord { float t = 0; Vec3 p, n; const Material* m = nullptr; };

struct Material {
    virtual ~Material() {}
    virtual bool scatter(const Ray& ray, const HitRecord& rec, Ray& scattered, Vec3& attenuation) const = 0;
    virtual Vec3 emitted() const { return Vec3(0); }
};
inline Vec3 reflect(Vec3 v, Vec3 n) { return v - 2 * v.dot(n) * n; }
inline bool refract(Vec3 v, Vec3 n, float ni_over_nt, Vec3& refracted) {
    Vec3 i = v.normalize();
    float idn = i.dot(n);
    float discriminant = 1 - ni_over_nt * ni_over_nt * (1 - idn * idn);
    if (discriminant > 0) { refracted = ni_over_nt * (i - n * idn) - n * std::sqrt(discriminant); return true; }
    return false;
}
inline float schlick(float cosine, float ref_idx) {
    float r0 = (1 - ref_idx) / (1 + ref_idx);
    float R0 = r0 * r0;
    return R0 + (1 - R0) * std::pow(1 - cosine, 5.0f);
}
struct Lambertian : Material {
    Vec3 albedo;
    explicit Lambertian(Vec3 a) : albedo(a) {}
    bool scatter(const Ray&, const HitRecord& rec, Ray& scattered, Vec3& attenuation) const override {
        Vec3 target = rec.p + rec.n + randomInUnitSphere();
        scattered = Ray(rec.p, target - rec.p);
        attenuation = albedo;
        return true;
    }
};
struct Metal : Material {
    Vec3 albedo; float fuzz;
    Metal(Vec3 a, float f) : albedo(a), fuzz(std::min(f, 1.0f)) {}
    bool scatter(const Ray& ray, const HitRecord& rec, Ray& scattered, Vec3& attenuation) const override {
        Vec3 reflected = reflect(ray.direction.normalize(), rec.n);
        scattered = Ray(rec.p, reflected + fuzz * randomInUnitSphere());
        attenuation = albedo;
        return scattered.direction.dot(rec.n) > 0;
    }
};
struct Dielectric : Material {
    float ref_idx;
    explicit Dielectric(float r) : ref_idx(r) {}
    bool scatter(const Ray& ray, const HitRecord& rec, Ray& scattered, Vec3& attenuation) const override {
        Vec3 reflected = reflect(ray.direction, rec.n);
        Vec3 outNormal; float ni_over_nt, cosine;
        float idn = ray.direction.dot(rec.n);
        if (idn > 0) {
            outNormal = Vec3() - rec.n; ni_over_nt = ref_idx;
            float tmp = idn / ray.direction.length();
            cosine = std::sqrt(1 - ref_idx * ref_idx * (1 - tmp * tmp));
        } else {
            outNormal = rec.n; ni_over_nt = 1.0f / ref_idx;
            cosine = -idn / ray.direction.length();
        }
        attenuation = Vec3(1.0f);
        Vec3 refracted; float reflect_prob = 1.0f;
        bool can = refract(ray.direction, outNormal, ni_over_nt, refracted);
        if (can) reflect_prob = schlick(cosine, ref_idx);
        if (randomFloat() < reflect_prob) { scattered = Ray(rec.p, reflected); return true; }
        if (can) { scattered = Ray(rec.p, refracted); return true; }
        return false;
    }
};
struct DiffuseLight : Material {
    Vec3 le;
    explicit DiffuseLight(Vec3 e) : le(e) {}
    bool scatter(const Ray&, const HitRecord&, Ray&, Vec3&) const override { return false; }
    Vec3 emitted() const override { return le; }
};

struct Hittable {
    virtual ~Hittable() {}
    virtual bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const = 0;
};
struct Sphere : Hittable {
    Vec3 center; float radius; const Material* material;
    Sphere(Vec3 c, float r, const Material* m) : center(c), radius(r), material(m) {}
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        Vec3 oc = ray.origin - center;
        float a = ray.direction.dot(ray.direction), b = oc.dot(ray.direction), c = oc.dot(oc) - radius * radius;
        float discriminant = b * b - a * c;
        if (discriminant <= 0) return false;
        float tmp = (-b - std::sqrt(discriminant)) / a;
        if (!(tmp < t_max && tmp > t_min)) {
            tmp = (-b + std::sqrt(discriminant)) / a;
            if (!(tmp < t_max && tmp > t_min)) return false;
        }
        rec.t = tmp; rec.p = ray.pointAt(tmp); rec.n = (rec.p - center) / radius; rec.m = material;
        return true;
    }
};
struct HittableList : Hittable {
    std::vector<std::unique_ptr<Hittable>> list;
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        bool any = false; float closest = t_max; HitRecord tmp;
        for (const auto& h : list)
            if (h->hitTest(ray, t_min, closest, tmp)) { any = true; closest = tmp.t; rec = tmp; }
        return any;
    }
};
struct Rect : Hittable {        // RT_Nextweek Rect.swift
    int a, b, c; float SA, EA, SB, EB, k; const Material* material;
    Rect(int a_, float sa, float ea, int b_, float sb, float eb, float k_, const Material* m)
        : a(a_), b(b_), c(3 - a_ - b_), SA(sa), EA(ea), SB(sb), EB(eb), k(k_), material(m) {}
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        float t = (k - ray.origin[c]) / ray.direction[c];
        if (t < t_min || t > t_max) return false;
        float _a = ray.origin[a] + t * ray.direction[a], _b = ray.origin[b] + t * ray.direction[b];
        if (_a < SA || _a > EA || _b < SB || _b > EB) return false;
        rec.t = t; rec.p = ray.pointAt(t);
        rec.n = Vec3(c == 0 ? 1.f : 0.f, c == 1 ? 1.f : 0.f, c == 2 ? 1.f : 0.f);
        rec.m = material;
        return true;
    }
};
struct NormalFlipped : Hittable {
    std::unique_ptr<Hittable> h;
    explicit NormalFlipped(Hittable* p) : h(p) {}
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        if (!h->hitTest(ray, t_min, t_max, rec)) return false;
        rec.n = -rec.n;
        return true;
    }
};
struct Box : Hittable {         // RT_Nextweek Box.swift
    HittableList sides;
    Box(Vec3 ps, Vec3 pe, const Material* m) {
        sides.list.emplace_back(new Rect(0, ps.x, pe.x, 1, ps.y, pe.y, pe.z, m));
        sides.list.emplace_back(new NormalFlipped(new Rect(0, ps.x, pe.x, 1, ps.y, pe.y, ps.z, m)));
        sides.list.emplace_back(new Rect(0, ps.x, pe.x, 2, ps.z, pe.z, pe.y, m));
        sides.list.emplace_back(new NormalFlipped(new Rect(0, ps.x, pe.x, 2, ps.z, pe.z, ps.y, m)));
        sides.list.emplace_back(new Rect(1, ps.y, pe.y, 2, ps.z, pe.z, pe.x, m));
        sides.list.emplace_back(new NormalFlipped(new Rect(1, ps.y, pe.y, 2, ps.z, pe.z, ps.x, m)));
    }
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        return sides.hitTest(ray, t_min, t_max, rec);
    }
};
struct Translate : Hittable {
    std::unique_ptr<Hittable> h; Vec3 offset;
    Translate(Hittable* p, Vec3 o) : h(p), offset(o) {}
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        Ray moved(ray.origin - offset, ray.direction);
        if (!h->hitTest(moved, t_min, t_max, rec)) return false;
        rec.p = rec.p + offset;
        return true;
    }
};
struct RotateY : Hittable {     // Hittable.swift Rotate, axis .y
    std::unique_ptr<Hittable> h; float sinTheta, cosTheta;
    RotateY(Hittable* p, float angle) : h(p) {
        float radians = (3.14159265358979323846f / 180) * angle;
        sinTheta = std::sin(radians); cosTheta = std::cos(radians);
    }
    bool hitTest(const Ray& ray, float t_min, float t_max, HitRecord& rec) const override {
        Vec3 o(cosTheta * ray.origin.x - sinTheta * ray.origin.z, ray.origin.y, sinTheta * ray.origin.x + cosTheta * ray.origin.z);
        Vec3 d(cosTheta * ray.direction.x - sinTheta * ray.direction.z, ray.direction.y, sinTheta * ray.direction.x + cosTheta * ray.direction.z);
        if (!h->hitTest(Ray(o, d), t_min, t_max, rec)) return false;
        Vec3 p = rec.p, n = rec.n;
        rec.p = Vec3(cosTheta * p.x + sinTheta * p.z, p.y, -sinTheta * p.x + cosTheta * p.z);
        rec.n = Vec3(cosTheta * n.x + sinTheta * n.z, n.y, -sinTheta * n.x + cosTheta * n.z);
        return true;
    }
};

struct Camera {                 // RT_Weekend Camera.swift
    Vec3 lookFrom, u, v, w, vertical, horizontal, cornerLowLeft; float lenRadius;
    Camera(Vec3 from, Vec3 at, Vec3 up, float vfov, float aspect, float aperture, float focus_dist) {
        lookFrom = from; lenRadius = aperture / 2;
        float theta = vfov * 3.14159265358979323846f / 180;
        float halfHeight = std::tan(theta / 2), halfWidth = aspect * halfHeight;
        w = (from - at).normalize(); u = up.cross(w).normalize(); v = w.cross(u);
        cornerLowLeft = from - halfWidth * focus_dist * u - halfHeight * focus_dist * v - focus_dist * w;
        vertical = 2 * halfHeight * focus_dist * v; horizontal = 2 * halfWidth * focus_dist * u;
    }
    Ray cast(float s, float t) const {
        Vec3 rd = lenRadius * randomInUnitDisk();
        Vec3 offset = u * rd.x + v * rd.y;
        Vec3 origin = lookFrom + offset;
        Vec3 sample = cornerLowLeft + s * horizontal + t * vertical;
        return Ray(origin, sample - origin);
    }
};

const float kFloatMax = std::numeric_limits<float>::max();

// main.swift:3-17 (sky background, no emission)
Vec3 color0(const Ray& ray, const Hittable& world, int depth) {
    HitRecord rec;
    t_rays++;
    if (!world.hitTest(ray, 0.001f, kFloatMax, rec)) {
        Vec3 d = ray.direction.normalize();
        float t = 0.5f * (d.y + 1.0f);
        return (1.0f - t) * Vec3(1.0f) + t * Vec3(0.5f, 0.7f, 1.0f);
    }
    if (depth < 50) {
        Ray scattered; Vec3 attenuation;
        if (rec.m->scatter(ray, rec, scattered, attenuation)) return attenuation * color0(scattered, world, depth + 1);
    }
    return Vec3();
}
// RT_Nextweek Render.swift:327-340 (black background, emission)
Vec3 color1(const Ray& ray, const Hittable& world, int depth) {
    HitRecord rec;
    t_rays++;
    if (!world.hitTest(ray, 0.001f, kFloatMax, rec)) return Vec3();
    Vec3 emitted = rec.m->emitted();
    if (depth < 50) {
        Ray scattered; Vec3 attenuation;
        if (rec.m->scatter(ray, rec, scattered, attenuation)) return emitted + attenuation * color1(scattered, world, depth + 1);
    }
    return emitted;
}

struct Scene {
    std::vector<std::unique_ptr<Material>> materials;
    HittableList world;
    std::unique_ptr<Camera> camera;
    bool emissive = false;
    const Material* add(Material* m) { materials.emplace_back(m); return m; }
};

void build_random_scene(Scene& s, float aspect) {      // main.swift:19-57,69-75
    Rng rng; rng.seed(2024, 7); t_rng = &rng;
    s.world.list.emplace_back(new Sphere(Vec3(0, -1000, 0), 1000, s.add(new Lambertian(Vec3(0.5f)))));
    s.world.list.emplace_back(new Sphere(Vec3(0, 1, 0), 1.0f, s.add(new Dielectric(1.5f))));
    s.world.list.emplace_back(new Sphere(Vec3(-4, 1, 0), 1.0f, s.add(new Lambertian(Vec3(0.4f, 0.2f, 0.1f)))));
    s.world.list.emplace_back(new Sphere(Vec3(4, 1, 0), 1.0f, s.add(new Metal(Vec3(0.7f, 0.6f, 0.5f), 0.0f))));
    for (int a = -11; a <= 11; ++a)
        for (int b = -11; b <= 11; ++b) {
            float mat = randomFloat();
            Vec3 center((float)a + 0.9f * randomFloat(), 0.2f, (float)b + 0.9f * randomFloat());
            if ((center - Vec3(4, 0.2f, 0)).length() > 0.9f) {
                const Material* m;
                if (mat < 0.8f) m = s.add(new Lambertian(Vec3(randomFloat() * randomFloat(), randomFloat() * randomFloat(), randomFloat() * randomFloat())));
                else if (mat < 0.95f) m = s.add(new Metal(Vec3(0.5f * (1 + randomFloat()), 0.5f * (1 + randomFloat()), 0.5f * (1 + randomFloat())), 0.5f * (1 + randomFloat())));
                else m = s.add(new Dielectric(1.5f));
                s.world.list.emplace_back(new Sphere(center, 0.2f, m));
            }
        }
    s.camera.reset(new Camera(Vec3(13, 2, 3), Vec3(), Vec3(0, 1, 0), 20, aspect, 0.1f, 10.0f));
    s.emissive = false;
    t_rng = nullptr;
}

void build_cornell(Scene& s, float aspect) {            // RT_Nextweek Render.swift:171-192, camera1 :59-81
    const Material* red = s.add(new Lambertian(Vec3(0.65f, 0.05f, 0.05f)));
    const Material* white = s.add(new Lambertian(Vec3(0.73f)));
    const Material* green = s.add(new Lambertian(Vec3(0.12f, 0.45f, 0.15f)));
    const Material* light = s.add(new DiffuseLight(Vec3(15)));
    auto& L = s.world.list;
    L.emplace_back(new NormalFlipped(new Rect(1, 0, 555, 2, 0, 555, 555, green)));
    L.emplace_back(new Rect(1, 0, 555, 2, 0, 555, 0, red));
    L.emplace_back(new Rect(0, 213, 343, 2, 227, 332, 554, light));
    L.emplace_back(new NormalFlipped(new Rect(0, 0, 555, 2, 0, 555, 555, white)));
    L.emplace_back(new Rect(0, 0, 555, 2, 0, 555, 0, white));
    L.emplace_back(new NormalFlipped(new Rect(0, 0, 555, 1, 0, 555, 555, white)));
    L.emplace_back(new Translate(new RotateY(new Box(Vec3(), Vec3(165), white), -18), Vec3(130, 0, 65)));
    L.emplace_back(new Translate(new RotateY(new Box(Vec3(), Vec3(165, 330, 165), white), 15), Vec3(265, 0, 295)));
    s.camera.reset(new Camera(Vec3(278, 278, -800), Vec3(278, 278, 0), Vec3(0, 1, 0), 40, aspect, 0.0f, 10.0f));
    s.emissive = true;
}

}  // namespace

int main(int argc, char** argv) {
    std::string scene_name = "cornell";
    int nx = 400, ny = 400, ns = 16, threads = (int)std::thread::hardware_concurrency();
    std::string ppm;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto next = [&](int& v) { if (i + 1 < argc) v = std::atoi(argv[++i]); };
        if (a == "--scene" && i + 1 < argc) scene_name = argv[++i];
        else if (a == "--width") next(nx);
        else if (a == "--height") next(ny);
        else if (a == "--spp") next(ns);
        else if (a == "--threads") next(threads);
        else if (a == "--ppm" && i + 1 < argc) ppm = argv[++i];
    }
    if (threads < 1) threads = 1;
    threads = std::min(threads, ny);
    Scene scene;
    if (scene_name == "random") build_random_scene(scene, (float)nx / (float)ny);
    else build_cornell(scene, (float)nx / (float)ny);

    std::vector<uint8_t> image((size_t)nx * ny * 3);
    std::vector<uint64_t> rays(threads, 0);
    const int dataUnit = ny / threads, remain = ny % threads;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int index = 0; index < threads; ++index)
        pool.emplace_back([&, index] {
            Rng rng; t_rng = &rng; t_rays = 0;
            int upper = dataUnit;
            if (index == threads - 1 && remain != 0) upper += remain;     // the last band takes the remainder
            for (int value = 0; value < upper; ++value) {
                int j = value + index * dataUnit;
                rng.seed(0x5EED0000ull, (uint64_t)j);     // one stream per ROW: image and ray count do not depend on --threads
                for (int i = 0; i < nx; ++i) {
                    Vec3 col;
                    for (int sidx = 0; sidx < ns; ++sidx) {
                        float u = ((float)i + randomFloat()) / (float)nx;
                        float v = ((float)j + randomFloat()) / (float)ny;
                        Ray ray = scene.camera->cast(u, v);
                        col = col + (scene.emissive ? color1(ray, scene.world, 0) : color0(ray, scene.world, 0));
                    }
                    col = col / (float)ns;
                    col = Vec3(std::sqrt(col.x), std::sqrt(col.y), std::sqrt(col.z));
                    uint8_t* px = &image[((size_t)(ny - 1 - j) * nx + i) * 3];
                    px[0] = (uint8_t)std::min(255.0f, 255.99f * col.x);
                    px[1] = (uint8_t)std::min(255.0f, 255.99f * col.y);
                    px[2] = (uint8_t)std::min(255.0f, 255.99f * col.z);
                }
            }
            rays[index] = t_rays;
        });
    for (auto& th : pool) th.join();
    double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    uint64_t total_rays = 0;
    for (uint64_t r : rays) total_rays += r;
    double mean = 0;
    for (uint8_t b : image) mean += b;
    mean /= (double)image.size();
    uint64_t hash = 1469598103934665603ull;                // FNV-1a over the 8-bit image (tests/test_cpu_baseline.py)
    for (uint8_t b : image) { hash ^= b; hash *= 1099511628211ull; }
    if (!ppm.empty()) {
        FILE* f = std::fopen(ppm.c_str(), "wb");
        if (f) { std::fprintf(f, "P6\n%d %d\n255\n", nx, ny); std::fwrite(image.data(), 1, image.size(), f); std::fclose(f); }
    }
    std::printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp\": %d, \"threads\": %d, \"seconds\": %.4f, "
                "\"rays\": %llu, \"mrays_per_s\": %.3f, \"mpaths_per_s\": %.3f, \"mean_pixel\": %.2f, \"image_fnv1a\": \"%016llx\"}\n",
                scene_name.c_str(), nx, ny, ns, threads, seconds, (unsigned long long)total_rays,
                total_rays / seconds / 1e6, (double)nx * ny * ns / seconds / 1e6, mean, (unsigned long long)hash);
    return 0;
}
