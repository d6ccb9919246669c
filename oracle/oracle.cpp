// oracle.cpp -- CPU oracle: scalar restatement of RT_Metal's path-tracing hot path.
//
// TEST INFRASTRUCTURE ONLY (see oracle.h).  PARITY PIN STATUS: unpinned beyond PCG32 and the
// Sobol' tables (tests/test_sobol.py) -- the reference has no tests/goldens for this path and cannot be built here.
//
// Conventions that resolve what Metal leaves unspecified (the HIP path uses the same ones):
//   * all arithmetic is IEEE-754 binary32, no FMA contraction (-ffp-contract=off);
//     unsuffixed literals are float in MSL, so every "1.0"/"0.5" below is float;
//   * dot(a,b) = a.x*b.x + a.y*b.y + a.z*b.z (left to right); length = sqrt(dot);
//     normalize(v) = v * (1 / length(v)); distance(a,b) = length(a-b);
//   * M * (v,0) = (c0*v.x + c1*v.y) + c2*v.z;  M * (p,1) = that + c3;
//   * min/max ignore NaN (fminf/fmaxf), Appendix B-10 of SURVEY.md;
//   * sin/cos/exp/log/pow/asin/acos/atan2 come from include/trc_detmath.h
//     (or glibc with -DORACLE_USE_LIBM, used to bound the statistical effect);
//   * constructor/function arguments are evaluated left to right (clang behaviour);
//   * uninitialised reference variables (bxPDF, HitRecord fields) start at 0 (B-3).
#include "oracle.h"

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "trc_detmath.h"
#include "trc_sobol.h"

namespace {

// ---------------------------------------------------------------- elementary functions
#ifdef ORACLE_USE_LIBM
inline float m_sin(float x) { return sinf(x); }
inline float m_cos(float x) { return cosf(x); }
inline float m_exp(float x) { return expf(x); }
inline float m_log(float x) { return logf(x); }
inline float m_pow(float x, float y) { return powf(x, y); }
inline float m_asin(float x) { return asinf(x); }
inline float m_acos(float x) { return acosf(x); }
inline float m_atan2(float y, float x) { return atan2f(y, x); }
#else
inline float m_sin(float x) { return dm_sinf(x); }
inline float m_cos(float x) { return dm_cosf(x); }
inline float m_exp(float x) { return dm_expf(x); }
inline float m_log(float x) { return dm_logf(x); }
inline float m_pow(float x, float y) { return dm_powf(x, y); }
inline float m_asin(float x) { return dm_asinf(x); }
inline float m_acos(float x) { return dm_acosf(x); }
inline float m_atan2(float y, float x) { return dm_atan2f(y, x); }
#endif

const float PI_F = 3.14159265358979323846f;   // M_PI_F
const float PI_2_F = 1.57079632679489661923f; // M_PI_2_F

// ---------------------------------------------------------------- vectors
struct V2 { float x, y; float& operator[](int i) { return i == 0 ? x : y; } float operator[](int i) const { return i == 0 ? x : y; } };
struct V3 {
    float x, y, z;
    float& operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 v3(float s) { return V3{s, s, s}; }
inline V3 v3(const trc_float3& f) { return V3{f.x, f.y, f.z}; }
inline V3 v3a(const float* f) { return V3{f[0], f[1], f[2]}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, V3 b) { return v3(a.x / b.x, a.y / b.y, a.z / b.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
inline float length(V3 a) { return sqrtf(dot(a, a)); }
inline V3 normalize(V3 a) { float inv = 1.0f / length(a); return a * inv; }
inline V3 vabs(V3 a) { return v3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
inline float fmin3(float a, float b, float c) { return fminf(fminf(a, b), c); }
inline float fmax3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

// column-major 4x4 times direction / point
inline V3 mul_dir(const trc_float4x4& m, V3 v) {
    V3 c0 = v3(m.columns[0].x, m.columns[0].y, m.columns[0].z);
    V3 c1 = v3(m.columns[1].x, m.columns[1].y, m.columns[1].z);
    V3 c2 = v3(m.columns[2].x, m.columns[2].y, m.columns[2].z);
    return (c0 * v.x + c1 * v.y) + c2 * v.z;
}
inline V3 mul_point(const trc_float4x4& m, V3 p) {
    V3 c3 = v3(m.columns[3].x, m.columns[3].y, m.columns[3].z);
    return mul_dir(m, p) + c3;
}

// ---------------------------------------------------------------- PCG32 / sampler
// Random.metal:3-26 (== RT_Metal/Tracer/pcg_basic.c:42-72)
struct pcg32_t { uint64_t state, inc; };
inline uint32_t pcg32_random_r(pcg32_t* rng) {
    uint64_t oldstate = rng->state;
    rng->state = oldstate * 6364136223846793005ULL + rng->inc;
    uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
    uint32_t rot = (uint32_t)(oldstate >> 59u);
    return (xorshifted >> rot) | (xorshifted << ((0u - rot) & 31));
}
inline void pcg32_srandom_r(pcg32_t* rng, uint64_t initstate, uint64_t initseq) {
    rng->state = 0U;
    rng->inc = (initseq << 1u) | 1u;
    pcg32_random_r(rng);
    rng->state += initstate;
    pcg32_random_r(rng);
}
// Random.metal:21-26: ldexp(float(u32), -32); float(u32) rounds to nearest, so 1.0f is reachable (B-4)
inline float randomF(pcg32_t* rng) { uint32_t i = pcg32_random_r(rng); return ldexpf((float)i, -32); }

// RandomSampler.hh:6-46
struct RandomSampler {
    pcg32_t* rng;
    float random() { return randomF(rng); }
    float sample1D() { return randomF(rng); }
    V2 sample2D() { float a = sample1D(); float b = sample1D(); return V2{a, b}; }
    V2 sampleUnitInDisk() {
        V2 p;
        do {
            float a = sample1D(); float b = sample1D();
            p = V2{2.0f * a - 1.0f, 2.0f * b - 1.0f};
        } while (p.x * p.x + p.y * p.y >= 1.0f);
        float inv = 1.0f / sqrtf(p.x * p.x + p.y * p.y);     // normalize(p): on the unit circle (B-2)
        return V2{p.x * inv, p.y * inv};
    }
};

// ---------------------------------------------------------------- SobolSampler.hh
// pbrt::SobolSampler as the reference declares it (SobolSampler.hh:26-167); its tables come from
// include/trc_sobol.h (generated, pinned to the reference's by tests/test_sobol.py).
struct SobolTables {
    uint32_t m32[TRC_SOBOL_DIMS * TRC_SOBOL_MATRIX_SIZE];
    uint64_t vdc[TRC_SOBOL_MAX_LOG2RES][TRC_SOBOL_MATRIX_SIZE], inv[TRC_SOBOL_MAX_LOG2RES][TRC_SOBOL_MATRIX_SIZE];
    SobolTables() {
        trc_sobol_matrices32(m32);
        for (uint32_t m = 1; m <= TRC_SOBOL_MAX_LOG2RES; ++m) trc_sobol_interval_tables(m, vdc[m - 1], inv[m - 1]);
    }
};
inline const SobolTables& sobol_tables() { static const SobolTables t; return t; }

inline uint32_t RoundUpPow2(uint32_t v) {                            // Math.hh:102-112
    v--;
    v |= v >> 1; v |= v >> 2; v |= v >> 4; v |= v >> 8; v |= v >> 16;
    return v + 1;
}
inline int PBRT_Log2Int(uint32_t v) { return 31 - __builtin_clz(v); }   // Math.hh:81-83 (v != 0)

// SobolSampler.hh:126-148
inline uint64_t SobolIntervalToIndex(const uint32_t m, uint64_t sampleIndex, uint32_t px, uint32_t py) {
    if (m == 0) return 0;
    const SobolTables& T = sobol_tables();
    const uint32_t m2 = m << 1;
    uint64_t index = sampleIndex << m2;
    uint64_t delta = 0;
    for (int c = 0; sampleIndex; sampleIndex >>= 1, ++c)
        if (sampleIndex & 1) delta ^= T.vdc[m - 1][c];
    uint64_t b = (((uint64_t)px << m) | (uint32_t)py) ^ delta;
    for (int c = 0; b; b >>= 1, ++c)
        if (b & 1) index ^= T.inv[m - 1][c];
    return index;
}
// SobolSampler.hh:154-164 (scramble 0).  The reference walks on into the next dimension's columns when the index
// has more than 52 bits; indices here stay below 2^52 (frame < 2^(52 - 2m)).
inline float SobolSampleFloat(uint64_t index, uint32_t dimension) {
    const SobolTables& T = sobol_tables();
    uint32_t v = 0;
    for (uint32_t i = dimension * TRC_SOBOL_MATRIX_SIZE; index != 0; index >>= 1, i++)
        if (index & 1) v ^= T.m32[i];
    return fminf((float)v * 2.3283064365386963e-10f, 1.0f - FLT_EPSILON);
}

struct SobolSampler {
    pcg32_t rng;                                   // a COPY of the pixel's stream (SobolSampler.hh:30,50)
    uint32_t xy[2];
    uint32_t resolution, log2Resolution;
    uint64_t mSampleIndex, mSobolIndex;
    uint32_t mDimension;
    SobolSampler(const pcg32_t& r, uint32_t frame, uint32_t x, uint32_t y, uint32_t w, uint32_t h)   // :50-61
        : rng(r), xy{x, y}, mSampleIndex(frame), mDimension(0) {
        resolution = RoundUpPow2(w > h ? w : h);
        log2Resolution = (uint32_t)PBRT_Log2Int(resolution);
        mSobolIndex = SobolIntervalToIndex(log2Resolution, mSampleIndex, xy[0], xy[1]);
    }
    float random() { return randomF(&rng); }                                     // :46-48
    float SampleDimension(uint64_t index, uint32_t dimension) const {            // :152-163
        if (dimension >= TRC_SOBOL_DIMS) return 0;     // NumSobolDimensions: 1024 there, 40 generated here
        float s = SobolSampleFloat(index, dimension);
        if (dimension <= 1) {
            s = s * (float)resolution + 0.0f;
            s = fminf(fmaxf(s - (float)xy[dimension], 0.0f), 1.0f - FLT_EPSILON);   // clamp
        }
        return s;
    }
    float sample1D() { return SampleDimension(mSobolIndex, mDimension++); }      // :63-65
    V2 sample2D() { float a = sample1D(); float b = sample1D(); return V2{a, b}; }
};

// ---------------------------------------------------------------- Math.hh
inline int32_t FloatToInt(float f) { int32_t i; memcpy(&i, &f, 4); return i; }
inline float IntToFloat(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

// Math.hh:51-55 (MachineEpsilon is the unparenthesised macro FLT_EPSILON * 0.5)
inline float gamma_n(int n) { return (n * FLT_EPSILON * 0.5f) / (1 - n * FLT_EPSILON * 0.5f); }

// Math.hh:57-74 (Ray Tracing Gems ch. 6)
inline V3 offset_ray(const V3 p, const V3 n) {
    const float origin = 1.0f / 32.0f, float_scale = 1.0f / 65536.0f, int_scale = 256.0f;
    int32_t of_x = (int32_t)(int_scale * n.x), of_y = (int32_t)(int_scale * n.y), of_z = (int32_t)(int_scale * n.z);
    V3 p_i = v3(IntToFloat(FloatToInt(p.x) + ((p.x < 0) ? -of_x : of_x)),
                IntToFloat(FloatToInt(p.y) + ((p.y < 0) ? -of_y : of_y)),
                IntToFloat(FloatToInt(p.z) + ((p.z < 0) ? -of_z : of_z)));
    return v3(fabsf(p.x) < origin ? p.x + float_scale * n.x : p_i.x,
              fabsf(p.y) < origin ? p.y + float_scale * n.y : p_i.y,
              fabsf(p.z) < origin ? p.z + float_scale * n.z : p_i.z);
}

// Math.hh:118-146
inline float ErfInv(float x) {
    float w, p;
    x = clampf(x, -.99999f, .99999f);
    w = -m_log((1 - x) * (1 + x));
    if (w < 5) {
        w = w - 2.5f;
        p = 2.81022636e-08f;
        p = 3.43273939e-07f + p * w;
        p = -3.5233877e-06f + p * w;
        p = -4.39150654e-06f + p * w;
        p = 0.00021858087f + p * w;
        p = -0.00125372503f + p * w;
        p = -0.00417768164f + p * w;
        p = 0.246640727f + p * w;
        p = 1.50140941f + p * w;
    } else {
        w = sqrtf(w) - 3;
        p = -0.000200214257f;
        p = 0.000100950558f + p * w;
        p = 0.00134934322f + p * w;
        p = -0.00367342844f + p * w;
        p = 0.00573950773f + p * w;
        p = -0.0076224613f + p * w;
        p = 0.00943887047f + p * w;
        p = 1.00167406f + p * w;
        p = 2.83297682f + p * w;
    }
    return p * x;
}

// Math.hh:148-167
inline float Erf(float x) {
    float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
    float p = 0.3275911f;
    int sign = 1;
    if (x < 0) sign = -1;
    x = fabsf(x);
    float t = 1 / (1 + p * x);
    float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * m_exp(-x * x);
    return sign * y;
}

// ---------------------------------------------------------------- Ray / HitRecord
// Ray.hh:10-33
struct Ray {
    V3 origin, direction;
    float eta = 1.0f;                       // Ray.hh:16
    int medium = TRC_MEDIUM_NIL;            // Ray.hh:18 (traceVolume only)
    Ray() : origin(v3(0)), direction(v3(0)) {}
    Ray(V3 o, V3 d) : origin(o) { direction = normalize(d); }
    void update(V3 o, V3 d) { origin = o; direction = normalize(d); }
    V3 pointAt(float t) const { return origin + direction * t; }
};

// HitRecord.hh:9-30 (+ pType/pIndex so tests can compare hit identity)
struct HitRecord {
    float t = 0;
    V3 p = v3(0);
    bool f = false;
    V3 gn = v3(0), sn = v3(0);
    V2 uv = V2{0, 0};
    uint32_t material = 0;
    float PDF = 0;
    int32_t pType = -1;
    uint32_t pIndex = 0;
    // HitRecord.hh:22-23, written by Cube::hit_test only (Cube.hh:30-31,39) and consumed by GridDensityMedium::Sample;
    // uninitialised in the reference, zero / null here
    Ray _r; float _t = 0; const trc_float4x4* modelMatrix = nullptr;
    void checkFace(const Ray& ray) { f = dot(ray.direction, gn) <= 0; sn = f ? gn : -gn; }
};

struct Counters {
    uint64_t rays = 0, shaded = 0, n_descend = 0, n_return = 0;
    uint64_t n_leaf[4] = {0, 0, 0, 0};
    uint64_t n_hit_triangle = 0, n_hit_cube = 0;
};

// ---------------------------------------------------------------- AABB.hh (device branch)
// AABB.hh:73-90
inline bool aabb_hit(const trc_AABB& b, const Ray& ray, V2 range_t) {
    V3 inverse = v3(1.0f) / ray.direction;
    V3 ts = (v3(b.mini) - ray.origin) * inverse;
    V3 te = (v3(b.maxi) - ray.origin) * inverse;
    V3 a = v3(fminf(ts.x, te.x), fminf(ts.y, te.y), fminf(ts.z, te.z));
    V3 bb = v3(fmaxf(ts.x, te.x), fmaxf(ts.y, te.y), fmaxf(ts.z, te.z));
    float tmin = fmax3(a.x, a.y, a.z);
    float tmax = fmin3(bb.x, bb.y, bb.z);
    tmin = fmaxf(tmin, range_t.x);
    tmax = fminf(tmax, range_t.y);
    return !(tmax < tmin || tmax < 0);
}
// AABB.hh:92-112
inline bool aabb_hit_t(const trc_AABB& b, const Ray& ray, V2 range_t, float& t) {
    V3 inverse = v3(1.0f) / ray.direction;
    V3 ts = (v3(b.mini) - ray.origin) * inverse;
    V3 te = (v3(b.maxi) - ray.origin) * inverse;
    V3 a = v3(fminf(ts.x, te.x), fminf(ts.y, te.y), fminf(ts.z, te.z));
    V3 bb = v3(fmaxf(ts.x, te.x), fmaxf(ts.y, te.y), fmaxf(ts.z, te.z));
    float tmin = fmax3(a.x, a.y, a.z);
    float tmax = fmin3(bb.x, bb.y, bb.z);
    tmin = fmaxf(tmin, range_t.x);
    tmax = fminf(tmax, range_t.y);
    if (tmax < tmin || tmax < 0) return false;
    t = (tmin < 0) ? tmax : tmin;   // maybe internal
    return true;
}
// AABB.hh:114-209 -- box test that also fills a record (used by Cube)
inline bool aabb_hit_record(const trc_AABB& b, const Ray& ray, V2 range_t, HitRecord& record) {
    const V3 mini = v3(b.mini), maxi = v3(b.maxi);
    float tmin = -FLT_MAX;
    float tmax = range_t.y;
    uint32_t axisPick = 0;
    V3 ddd = ray.origin - mini;
    V3 bbb = ray.origin - maxi;
    const float pad = 1 + 2 * gamma_n(3);

    if ((ddd.x > 0 && ddd.y > 0 && ddd.z > 0) && (bbb.x < 0 && bbb.y < 0 && bbb.z < 0)) {   // internal hit
        for (int i = 0; i < 3; ++i) {
            float min_bound = (mini[i] - ray.origin[i]) / ray.direction[i];
            float max_bound = (maxi[i] - ray.origin[i]) / ray.direction[i];
            float ts = fminf(max_bound, min_bound);
            float te = fmaxf(max_bound, min_bound);
            te *= pad;
            tmin = fmaxf(ts, tmin);
            if (te < tmax) { tmax = te; axisPick = i; }
            if (tmax < tmin || tmax < 0) return false;
        }
        record.t = tmax;
        record.gn = v3(0);
        record.gn[axisPick] = ray.direction[axisPick] > 0 ? 1.0f : -1.0f;
        V3 hitPoint = ray.pointAt(record.t);
        record.p = hitPoint;
        record.p[axisPick] = ray.direction[axisPick] > 0 ? maxi[axisPick] : mini[axisPick];
        record.uv = V2{hitPoint[(1 + axisPick) % 3], hitPoint[(2 + axisPick) % 3]};
        return true;
    }
    for (int i = 0; i < 3; ++i) {
        float min_bound = (mini[i] - ray.origin[i]) / ray.direction[i];
        float max_bound = (maxi[i] - ray.origin[i]) / ray.direction[i];
        float ts = fminf(max_bound, min_bound);
        float te = fmaxf(max_bound, min_bound);
        te *= pad;
        tmax = fminf(te, tmax);
        if (ts > tmin) { tmin = ts; axisPick = i; }
        if (tmax < tmin || tmax < 0) return false;
    }
    record.t = tmin;   // external
    record.gn = v3(0);
    record.gn[axisPick] = ray.direction[axisPick] > 0 ? -1.0f : 1.0f;
    V3 hitPoint = ray.pointAt(record.t);
    record.p = hitPoint;
    record.p[axisPick] = ray.direction[axisPick] > 0 ? mini[axisPick] : maxi[axisPick];
    record.uv = V2{hitPoint[(1 + axisPick) % 3], hitPoint[(2 + axisPick) % 3]};
    return true;
}

// ---------------------------------------------------------------- primitives
// Sphere.hh:19-78
inline void sphereUV(const V3& p, V2& uv) {
    float phi = m_atan2(p.z, p.x);
    float theta = m_asin(p.y);
    uv.x = 1 - (phi + PI_F) / (2 * PI_F);
    uv.y = (theta + PI_2_F) / PI_F;
}
inline bool sphere_hit_test(const trc_Sphere& s, const Ray& ray, V2& range_t, HitRecord& rec) {
    const V3 center = v3(s.center);
    V3 oc = ray.origin - center;
    float a = dot(ray.direction, ray.direction);
    float half_b = dot(oc, ray.direction);
    float c = dot(oc, oc) - s.radius * s.radius;
    float discriminant = half_b * half_b - a * c;
    if (discriminant <= 0) return false;
    float t_min = range_t.x, t_max = range_t.y;
    float root = sqrtf(discriminant);
    float temp = (-half_b - root) / a;
    if (temp < t_max && temp > t_min) {
        rec.t = temp;
        rec.p = ray.pointAt(rec.t);
        rec.gn = (rec.p - center) / s.radius;
        rec.checkFace(ray);
        sphereUV(rec.gn, rec.uv);
        rec.material = s.material;
        range_t.y = rec.t;
        return true;
    }
    temp = (-half_b + root) / a;
    if (temp < t_max && temp > t_min) {
        rec.t = temp;
        rec.p = ray.pointAt(rec.t);
        rec.gn = (rec.p - center) / s.radius;
        rec.checkFace(ray);
        sphereUV(rec.gn, rec.uv);
        rec.material = s.material;
        range_t.y = rec.t;
        return true;
    }
    return false;
}

// Square.hh:31-38 (area is 2*i*j, B-7)
inline float square_area(const trc_Square& q) {
    float i = q.range_i.y - q.range_i.x;
    float j = q.range_j.y - q.range_j.x;
    return 2 * i * j;
}
inline float square_areaPDF(const trc_Square& q) { return 1 / square_area(q); }
// Square.hh:60-113
inline bool square_hit_test(const trc_Square& q, const Ray& ray, V2& range_t, HitRecord& rec) {
    float t = (q.value_k - ray.origin[q.axis_k]) / ray.direction[q.axis_k];
    if (std::isinf(t) || std::isnan(t)) return false;
    if (t < range_t.x || t > range_t.y) return false;
    float a = ray.origin[q.axis_i] + t * ray.direction[q.axis_i];
    if (a < q.range_i.x || a > q.range_i.y) return false;
    float b = ray.origin[q.axis_j] + t * ray.direction[q.axis_j];
    if (b < q.range_j.x || b > q.range_j.y) return false;
    rec.uv.x = (a - q.range_i.x) / (q.range_i.y - q.range_i.x);
    rec.uv.y = (b - q.range_j.x) / (q.range_j.y - q.range_j.x);
    rec.t = t;
    rec.gn = v3(0);
    rec.gn[q.axis_k] = 1;
    rec.checkFace(ray);
    rec.gn = rec.sn;
    rec.p[q.axis_k] = q.value_k;
    rec.p[q.axis_i] = a;
    rec.p[q.axis_j] = b;
    range_t.y = t;
    rec.PDF = square_areaPDF(q);
    rec.material = q.material;
    return true;
}

// Sampling.hh:6-11
struct LightSampleRecord { V3 p = v3(0), n = v3(0); float areaPDF = 0; uint32_t material = 0; };
// Square.hh:40-58
inline void square_sample(const trc_Square& q, V2 u, V3 pos, LightSampleRecord& lsr) {
    lsr.p[q.axis_k] = q.value_k;
    lsr.p[q.axis_i] = q.range_i.x + u[0] * (q.range_i.y - q.range_i.x);
    lsr.p[q.axis_j] = q.range_j.x + u[1] * (q.range_j.y - q.range_j.x);
    lsr.n = v3(0);
    lsr.n[q.axis_k] = 1;
    V3 w = normalize(pos - lsr.p);
    float v = 1.0f;
    lsr.n[q.axis_k] = copysignf(v, dot(w, lsr.n));
    lsr.p = offset_ray(lsr.p, lsr.n);
    lsr.areaPDF = square_areaPDF(q);
    lsr.material = q.material;
}

// Cube.hh:17-47 (world range_t handed to an object-space slab test, B-8)
inline bool cube_hit_test(const trc_Cube& cube, const Ray& ray, V2& range_t, HitRecord& rec, Counters* cnt) {
    V3 origin = mul_point(cube.inverse_matrix, ray.origin);
    V3 direction = mul_dir(cube.inverse_matrix, ray.direction);
    HitRecord _record;
    _record.PDF = rec.PDF;          // the reference leaves PDF uninitialised; keep the previous value
    Ray _ray(origin, direction);
    if (!aabb_hit_record(cube.box, _ray, range_t, _record)) return false;
    if (cnt) cnt->n_hit_cube++;
    _record.p = mul_point(cube.model_matrix, _record.p);
    _record._r = _ray;                               // Cube.hh:30-31
    _record._t = _record.t;
    _record.t = length(ray.origin - _record.p);     // distance(ray.origin, p)
    if (_record.t >= range_t.y) return false;
    range_t.y = _record.t;
    _record.material = cube.material;
    _record.modelMatrix = &cube.model_matrix;        // Cube.hh:39
    _record.gn = normalize(mul_dir(cube.normal_matrix, _record.gn));
    _record.checkFace(ray);
    rec = _record;
    return true;
}

// Triangle.hh:31-85 (two-sided; unnormalised interpolated normal; material 19 hard-coded, B-5)
inline bool triangle_hit_test(const trc_TriangleVertex& A, const trc_TriangleVertex& B, const trc_TriangleVertex& C,
                              const Ray& ray, V2& range, HitRecord& rec, Counters* cnt) {
    const V3 ori = ray.origin, dir = ray.direction;
    const V3 v0 = v3a(A.v), v1 = v3a(B.v), v2 = v3a(C.v);
    V3 v0v1 = v1 - v0;
    V3 v0v2 = v2 - v0;
    V3 pvec = cross(dir, v0v2);
    float det = dot(v0v1, pvec);
    if (fabsf(det) < FLT_EPSILON) return false;
    float invDet = 1 / det;
    V3 tvec = ori - v0;
    float u = dot(tvec, pvec) * invDet;
    if (u < 0 || u > 1) return false;
    V3 qvec = cross(tvec, v0v1);
    float v = dot(dir, qvec) * invDet;
    if (v < 0 || (u + v) > 1) return false;
    float w = 1.0f - u - v;
    float t = dot(v0v2, qvec) * invDet;
    if (t > range.y || t < range.x) return false;
    if (cnt) cnt->n_hit_triangle++;
    rec.p = u * v1 + v * v2 + w * v0;
    range.y = t;
    rec.t = t;
    rec.gn = u * v3a(B.n) + v * v3a(C.n) + w * v3a(A.n);
    rec.uv = V2{u * B.uv[0] + v * C.uv[0] + w * A.uv[0], u * B.uv[1] + v * C.uv[1] + w * A.uv[1]};
    rec.checkFace(ray);
    rec.material = 19;
    return true;
}

// ---------------------------------------------------------------- Scene::hit, Render.hh:135-252
struct Scene {
    const trc_scene& prims;
    Counters* cnt;

    bool leaf_test(uint32_t selected, const Ray& ray, V2& range_t, HitRecord& rec) {
        const trc_BVH& node = prims.bvhList[selected];
        const uint32_t pIndex = node.pIndex;
        bool ok = false;
        switch (node.pType) {
            case TRC_PRIM_SPHERE:
                if (cnt) cnt->n_leaf[0]++;
                ok = sphere_hit_test(prims.sphereList[pIndex], ray, range_t, rec); break;
            case TRC_PRIM_SQUARE:
                if (cnt) cnt->n_leaf[1]++;
                ok = square_hit_test(prims.squareList[pIndex], ray, range_t, rec); break;
            case TRC_PRIM_CUBE:
                if (cnt) cnt->n_leaf[2]++;
                ok = cube_hit_test(prims.cubeList[pIndex], ray, range_t, rec, cnt); break;
            case TRC_PRIM_TRIANGLE: {
                if (cnt) cnt->n_leaf[3]++;
                uint32_t r = pIndex * 3;
                ok = triangle_hit_test(prims.triList[prims.idxList[r]], prims.triList[prims.idxList[r + 1]],
                                       prims.triList[prims.idxList[r + 2]], ray, range_t, rec, cnt);
                break;
            }
            default: break;
        }
        if (ok) { rec.pType = node.pType; rec.pIndex = pIndex; }
        return ok;
    }

    bool hit(const Ray& ray, HitRecord& hitRecord, const float test_t, bool any = false,
             uint32_t* out_descend = nullptr, uint32_t* out_return = nullptr, uint32_t* out_leaf = nullptr) {
        if (cnt) cnt->rays++;
        const trc_BVH* bvhList = prims.bvhList;
        uint32_t the_index = 0;
        uint32_t tested_index = UINT_MAX;
        uint64_t stack_mark = 0;          // 32 bits in the reference (B-9); 64 here, depth asserted <= 64
        uint32_t stack_level = 0;
        uint32_t nd = 0, nr = 0, nl = 0;
        V2 range_t = V2{FLT_MIN, test_t};
        bool result_early = false;
        // A ray with a NaN / infinite component (randomF == 1.0 makes BeckmannSample11 / TrowbridgeReitzSample11
        // produce inf -> NaN directions, SURVEY B-4) MISSES the scene.  Metal's fast-math min/max on NaN are
        // unspecified (B-10); with NaN-ignoring min/max such a ray would "hit" every box and walk the whole
        // tree (1.3 s for one lane on a 1 M-triangle scene).  The sample's radiance is NaN -> 0 either way.
        const float finite_probe = fabsf(ray.origin.x) + fabsf(ray.origin.y) + fabsf(ray.origin.z) +
                                   fabsf(ray.direction.x) + fabsf(ray.direction.y) + fabsf(ray.direction.z);
        const bool ray_ok = finite_probe < INFINITY;

        if (ray_ok && aabb_hit(bvhList[the_index].bBOX, ray, range_t)) {
            do {   // travel in bvh
                uint32_t selected_index = UINT_MAX;
                uint32_t left_index = bvhList[the_index].left;
                uint32_t right_index = bvhList[the_index].right;
                uint32_t parent_index = bvhList[the_index].parent;

                if (tested_index != left_index && tested_index != right_index) {   // came from parent
                    nd++;
                    float t_left = range_t.y, t_right = range_t.y;
                    bool left_test = aabb_hit_t(bvhList[left_index].bBOX, ray, range_t, t_left);
                    bool right_test = aabb_hit_t(bvhList[right_index].bBOX, ray, range_t, t_right);
                    if (!left_test && !right_test) {
                        tested_index = the_index;
                        the_index = parent_index;
                        stack_level -= 1;   // pop stack
                        continue;
                    }
                    bool needTestAnother = left_test && right_test;
                    if (needTestAnother) stack_mark |= 1ULL << (stack_level & 63);
                    selected_index = (t_left < t_right) ? left_index : right_index;
                } else {   // came from child
                    nr++;
                    uint64_t needCheckChild = (stack_mark >> (stack_level & 63)) & 1ULL;
                    stack_mark &= ~(1ULL << (stack_level & 63));
                    if (0 == needCheckChild) {   // go up
                        tested_index = the_index;
                        the_index = parent_index;
                        stack_level -= 1;
                        continue;
                    }
                    selected_index = (tested_index == left_index) ? right_index : left_index;
                }

                if (bvhList[selected_index].pType == TRC_PRIM_BVH) {
                    the_index = selected_index;
                    stack_level += 1;
                    continue;
                }
                nl++;
                leaf_test(selected_index, ray, range_t, hitRecord);
                if (any && range_t.y < test_t) { result_early = true; break; }
                tested_index = selected_index;
            } while (tested_index != 0);
        }
        if (cnt) { cnt->n_descend += nd; cnt->n_return += nr; }
        if (out_descend) *out_descend = nd;
        if (out_return) *out_return = nr;
        if (out_leaf) *out_leaf = nl;
        if (result_early) return true;
        return range_t.y < test_t;
    }
};

// ---------------------------------------------------------------- Sampling.hh
// Sampling.hh:18-34
inline void CoordinateSystem(const V3& a, V3& b, V3& c) {
    if (fabsf(a.x) > fabsf(a.y)) b = v3(-a.z, 0, a.x);
    else b = v3(0, a.z, -a.y);
    b = normalize(b);
    c = cross(a, b);
}
// Sampling.hh:79-99
inline V2 ConcentricSampleDisk(const V2& u) {
    V2 uOffset = V2{2.f * u.x - 1, 2.f * u.y - 1};
    if (uOffset.x == 0 && uOffset.y == 0) return V2{0, 0};
    const float PiOver2 = PI_F / 2.0f, PiOver4 = PI_F / 4.0f;
    float theta, r;
    if (fabsf(uOffset.x) > fabsf(uOffset.y)) {
        r = uOffset.x;
        theta = PiOver4 * (uOffset.y / uOffset.x);
    } else {
        r = uOffset.y;
        theta = PiOver2 - PiOver4 * (uOffset.x / uOffset.y);
    }
    return V2{r * m_cos(theta), r * m_sin(theta)};
}
// Sampling.hh:125-129
inline V3 CosineSampleHemisphere(const V2& u) {
    V2 d = ConcentricSampleDisk(u);
    float z = sqrtf(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return v3(d.x, d.y, z);
}
// Sampling.hh:137-140
inline float PowerHeuristic(int nf, float fPdf, int ng, float gPdf) {
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
// Sampling.hh:148-203
inline float CosTheta(const V3& w) { return w.z; }
inline float Cos2Theta(const V3& w) { return w.z * w.z; }
inline float AbsCosTheta(const V3& w) { return fabsf(w.z); }
inline float Sin2Theta(const V3& w) { return fmaxf(0.0f, 1.0f - Cos2Theta(w)); }
inline float SinTheta(const V3& w) { return sqrtf(Sin2Theta(w)); }
inline float TanTheta(const V3& vec) {
    float temp = 1 - vec.z * vec.z;
    if (temp <= 0.0f || vec.z == 0.0f) return 0.0f;
    return sqrtf(temp) / vec.z;
}
inline float Tan2Theta(const V3& vec) {
    float zz = vec.z * vec.z;
    float temp = 1 - zz;
    if (temp <= 0.0f || zz == 0.0f) return 0.0f;
    return temp / zz;
}
inline float CosPhi(const V3& w) { float s = SinTheta(w); return (s == 0) ? 1 : clampf(w.x / s, -1.0f, 1.0f); }
inline float SinPhi(const V3& w) { float s = SinTheta(w); return (s == 0) ? 0 : clampf(w.y / s, -1.0f, 1.0f); }
inline float Cos2Phi(const V3& w) { float r = CosPhi(w); return r * r; }
inline float Sin2Phi(const V3& w) { float r = SinPhi(w); return r * r; }
inline float Sqr(float v) { return v * v; }
inline V3 Faceforward(const V3& n, const V3& v) { return (dot(n, v) < 0.f) ? -n : n; }

// ---------------------------------------------------------------- BXDF.hh / BXDF.metal
// BXDF.hh:24-41 (cos(theta_i) from wo.z, not dot(wo,n), B-6)
inline V3 Reflect(const V3& wo, const V3& n) { return -wo + 2 * dot(wo, n) * n; }
inline bool Refract(const V3& wo, const V3& n, float eta, V3& wi) {
    float cosThetaI = wo.z;
    float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrtf(1 - sin2ThetaT);
    wi = eta * -wo + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
// BXDF.metal:3-22 (scalar, broadcast to rgb by the callers)
inline float FrDielectric(float cosi, float eta) {
    cosi = clampf(cosi, -1.0f, 1.0f);
    bool entering = cosi > 0.f;
    if (!entering) { eta = 1 / eta; cosi = -cosi; }
    float sin2Theta_i = 1 - cosi * cosi;
    float sin2Theta_t = sin2Theta_i / Sqr(eta);
    if (sin2Theta_t >= 1) return 1.f;
    float cosTheta_t = sqrtf(fmaxf(0.0f, 1 - sin2Theta_t));
    float r_parl = (eta * cosi - cosTheta_t) / (eta * cosi + cosTheta_t);
    float r_perp = (cosi - eta * cosTheta_t) / (cosi + eta * cosTheta_t);
    return (r_parl * r_parl + r_perp * r_perp) / 2;
}
// BXDF.metal:24-34
inline V3 FrConductor(float cosi, const V3& eta, const V3& k) {
    V3 tmp = (eta * eta + k * k) * cosi * cosi;
    V3 Rparl2 = (tmp - (2.f * eta * cosi) + v3(1)) / (tmp + (2.f * eta * cosi) + v3(1));
    V3 tmp_f = eta * eta + k * k;
    V3 Rperp2 = (tmp_f - (2.f * eta * cosi) + v3(cosi * cosi)) / (tmp_f + (2.f * eta * cosi) + v3(cosi * cosi));
    return 0.5f * (Rparl2 + Rperp2);
}
struct FresnelConductor {   // BXDF.hh:59-70
    V3 eta, k;
    V3 Evaluate(float cosThetaI) const { return FrConductor(fabsf(cosThetaI), eta, k); }
};
struct FresnelDielectric {  // BXDF.hh:72-81
    float eta;
    V3 Evaluate(float cosThetaI) const { return v3(FrDielectric(cosThetaI, eta)); }
};

// ---------------------------------------------------------------- MatteBXDF.hh:6-22
struct Lambertian {
    float F(const V3&, const V3& wi, const V2&) const { return wi.z / PI_F; }
    float PDF(const V3& wo, const V3& wi, const V2&) const { return wo.z * wi.z > 0 ? fabsf(wi.z) / PI_F : 0; }
    float S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const {
        wi = CosineSampleHemisphere(uu);
        pdf = PDF(wo, wi, uu);
        return wi.z / PI_F;
    }
};

// ---------------------------------------------------------------- MicrofacetBXDF.h
// Beckmann, MicrofacetBXDF.h:137-290
struct Beckmann {
    float alphax, alphay;
    Beckmann(float ax, float ay) : alphax(fmaxf(0.001f, ax)), alphay(fmaxf(0.001f, ay)) {}

    float Lambda(const V3& w) const {
        float absTanTheta = fabsf(TanTheta(w));
        if (std::isinf(absTanTheta)) return 0.;
        float alpha = sqrtf(Cos2Phi(w) * alphax * alphax + Sin2Phi(w) * alphay * alphay);
        float a = 1 / (alpha * absTanTheta);
        if (a >= 1.6f) return 0;
        return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
    }
    float D(const V3& wh) const {
        float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        return m_exp(-tan2Theta * (Cos2Phi(wh) / (alphax * alphax) + Sin2Phi(wh) / (alphay * alphay))) /
               (PI_F * alphax * alphay * cos4Theta);
    }
    float G1(const V3& w) const { return 1 / (1 + Lambda(w)); }
    float G(const V3& wo, const V3& wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    float PDF(const V3& wo, const V3& wh) const { return D(wh) * G1(wo) * fabsf(dot(wo, wh)) / AbsCosTheta(wo); }

    static void BeckmannSample11(float cosThetaI, float U1, float U2, float* slope_x, float* slope_y) {
        if (cosThetaI > .9999f) {   // normal incidence
            float r = sqrtf(-m_log(1.0f - U1));
            float sinPhi = m_sin(2 * PI_F * U2);
            float cosPhi = m_cos(2 * PI_F * U2);
            *slope_x = r * cosPhi;
            *slope_y = r * sinPhi;
            return;
        }
        float sinThetaI = sqrtf(fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI));
        float tanThetaI = sinThetaI / cosThetaI;
        float cotThetaI = 1 / tanThetaI;
        float a = -1, c = Erf(cotThetaI);
        float sample_x = fmaxf(U1, 1e-6f);
        float thetaI = m_acos(cosThetaI);
        float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        float b = c - (1 + c) * m_pow(1 - sample_x, fit);
        const float SQRT_PI_INV = 1.f / sqrtf(PI_F);
        float normalization = 1 / (1 + c + SQRT_PI_INV * tanThetaI * m_exp(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            float invErf = ErfInv(b);
            float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * m_exp(-invErf * invErf)) - sample_x;
            float derivative = normalization * (1 - invErf * tanThetaI);
            if (fabsf(value) < 1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        *slope_x = ErfInv(b);
        *slope_y = ErfInv(2.0f * fmaxf(U2, 1e-6f) - 1.0f);
    }
    static V3 BeckmannSample(const V3& wi, float alpha_x, float alpha_y, float U1, float U2) {
        V3 wiStretched = normalize(v3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
        float slope_x, slope_y;
        BeckmannSample11(CosTheta(wiStretched), U1, U2, &slope_x, &slope_y);
        float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alpha_x * slope_x;
        slope_y = alpha_y * slope_y;
        return normalize(v3(-slope_x, -slope_y, 1.f));
    }
    V3 sample_wh(const V3& wo, const V2& u) const {
        bool flip = wo.z < 0;
        V3 wh = BeckmannSample(flip ? -wo : wo, alphax, alphay, u[0], u[1]);
        if (flip) wh = -wh;
        return wh;
    }
};

// TrowbridgeReitz, MicrofacetBXDF.h:292-434
struct TrowbridgeReitz {
    float alpha_x, alpha_y;
    TrowbridgeReitz(float ax, float ay) : alpha_x(fmaxf(0.001f, ax)), alpha_y(fmaxf(0.001f, ay)) {}

    float D(const V3& wh) const {
        float tan2Theta = Tan2Theta(wh);
        if (std::isinf(tan2Theta)) return 0.;
        const float cos4Theta = Cos2Theta(wh) * Cos2Theta(wh);
        if (cos4Theta < 1e-16f) return 0;
        float e = (Cos2Phi(wh) / Sqr(alpha_x) + Sin2Phi(wh) / Sqr(alpha_y)) * tan2Theta;
        return 1 / (PI_F * alpha_x * alpha_y * Sqr(1 + e) * cos4Theta);
    }
    float Lambda(const V3& w) const {
        float tan2Theta = Tan2Theta(w);
        if (std::isinf(tan2Theta)) return 0.;
        float alpha2 = Sqr(CosPhi(w) * alpha_x) + Sqr(SinPhi(w) * alpha_y);
        return 0.5f * (sqrtf(1 + alpha2 * tan2Theta) - 1);
    }
    float G1(const V3& w) const { return 1 / (1 + Lambda(w)); }
    float G(const V3& wo, const V3& wi) const { return 1 / (1 + Lambda(wo) + Lambda(wi)); }
    float PDF(const V3& wo, const V3& wh) const { return D(wh) * G1(wo) * fabsf(dot(wo, wh) / CosTheta(wo)); }

    static void TrowbridgeReitzSample11(float cosTheta, float U1, float U2, float* slope_x, float* slope_y) {
        if (cosTheta > .9999f) {
            float r = sqrtf(U1 / (1 - U1));
            float phi = 6.28318530718f * U2;
            *slope_x = r * m_cos(phi);
            *slope_y = r * m_sin(phi);
            return;
        }
        float sinTheta = sqrtf(fmaxf(0.0f, 1.0f - cosTheta * cosTheta));
        float tanTheta = sinTheta / cosTheta;
        float a = 1 / tanTheta;
        float G1 = 2 / (1 + sqrtf(1.f + 1.f / (a * a)));
        float A = 2 * U1 / G1 - 1;
        float tmp = 1.f / (A * A - 1.f);
        if (tmp > 1e10f) tmp = 1e10f;
        float B = tanTheta;
        float D = sqrtf(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
        float slope_x_1 = B * tmp - D;
        float slope_x_2 = B * tmp + D;
        *slope_x = (A < 0 || slope_x_2 > 1.f / tanTheta) ? slope_x_1 : slope_x_2;
        float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        *slope_y = S * z * sqrtf(1.f + *slope_x * *slope_x);
    }
    static V3 TrowbridgeReitzSample(const V3& wi, float alpha_x, float alpha_y, float U1, float U2) {
        V3 wiStretched = normalize(v3(alpha_x * wi.x, alpha_y * wi.y, wi.z));
        float slope_x, slope_y;
        TrowbridgeReitzSample11(CosTheta(wiStretched), U1, U2, &slope_x, &slope_y);
        float tmp = CosPhi(wiStretched) * slope_x - SinPhi(wiStretched) * slope_y;
        slope_y = SinPhi(wiStretched) * slope_x + CosPhi(wiStretched) * slope_y;
        slope_x = tmp;
        slope_x = alpha_x * slope_x;
        slope_y = alpha_y * slope_y;
        return normalize(v3(-slope_x, -slope_y, 1.f));
    }
    V3 sample_wh(const V3& wo, const V2& u) const {
        bool flip = wo.z < 0;
        V3 wh = TrowbridgeReitzSample(flip ? -wo : wo, alpha_x, alpha_y, u[0], u[1]);
        if (flip) wh = -wh;
        return wh;
    }
};

// MicrofacetBXDF.h:7-62
template <typename DistType, typename FrType>
struct MicrofacetReflection {
    V3 R; FrType fresnel; DistType distribution;
    V3 F(const V3& wo, const V3& wi, const V2&) const {
        float cosThetaO = AbsCosTheta(wo), cosThetaI = AbsCosTheta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return v3(0);
        V3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return v3(0);
        wh = normalize(wh);
        V3 Fr = fresnel.Evaluate(dot(wi, Faceforward(wh, v3(0, 0, 1))));
        return R * distribution.D(wh) * distribution.G(wo, wi) * Fr / (4 * cosThetaI * cosThetaO);
    }
    float PDF(const V3& wo, const V3& wi, const V2&) const {
        if (wo.z * wi.z <= 0) return 0;
        V3 wh = normalize(wo + wi);
        return distribution.PDF(wo, wh) / (4 * dot(wo, wh));
    }
    V3 S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const {
        if (wo.z == 0) return v3(0);
        V3 wh = distribution.sample_wh(wo, uu);
        if (dot(wo, wh) <= 0) return v3(0);
        wi = Reflect(wo, wh);
        if (wo.z * wi.z <= 0) return v3(0);
        pdf = distribution.PDF(wo, wh) / (4 * dot(wo, wh));
        return F(wo, wi, uu);
    }
};

// MicrofacetBXDF.h:64-135 (its Fresnel is FresnelDielectric(etaA), :81)
enum class TransportMode { Radiance, Importance };
template <typename DistType>
struct MicrofacetTransmission {
    V3 T; DistType dist; float etaA, etaB; TransportMode mode; FresnelDielectric fresnel;
    MicrofacetTransmission(V3 T, DistType d, float etaA, float etaB, TransportMode mode)
        : T(T), dist(d), etaA(etaA), etaB(etaB), mode(mode), fresnel{etaA} {}
    V3 F(const V3& wo, const V3& wi, const V2&) const {
        if (wo.z * wi.z > 0) return v3(0);
        float cosThetaO = CosTheta(wo), cosThetaI = CosTheta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return v3(0);
        float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        V3 wh = normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0) return v3(0);
        V3 Fr = fresnel.Evaluate(dot(wo, wh));
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = (mode == TransportMode::Radiance) ? (1 / eta) : 1;
        return (v3(1.0f) - Fr) * T *
               fabsf(dist.D(wh) * dist.G(wo, wi) * eta * eta * fabsf(dot(wi, wh)) * fabsf(dot(wo, wh)) * factor * factor /
                     (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    float PDF(const V3& wo, const V3& wi, const V2&) const {
        if (wo.z * wi.z > 0) return 0;
        float eta = CosTheta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        V3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return dist.PDF(wo, wh) * dwh_dwi;
    }
    V3 S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const {
        if (wo.z == 0) return v3(0);
        V3 wh = dist.sample_wh(wo, uu);
        if (dot(wo, wh) < 0) return v3(0);
        float eta = CosTheta(wo) > 0 ? (etaA / etaB) : (etaB / etaA);
        if (!Refract(wo, wh, eta, wi)) return v3(0);
        pdf = PDF(wo, wi, uu);
        return F(wo, wi, uu);
    }
};

// MicrofacetBXDF.h:436-455
typedef MicrofacetReflection<TrowbridgeReitz, FresnelConductor> MetalMaterial;
inline MetalMaterial createMetalMaterial() {
    return MetalMaterial{v3(1.0f), FresnelConductor{v3(0.18f, 0.15f, 0.81f), v3(1)}, TrowbridgeReitz(0.01f, 0.02f)};
}
// MicrofacetBXDF.h:457-530
typedef MicrofacetReflection<Beckmann, FresnelDielectric> PlasticX;
struct PlasticMaterial {
    V3 ks = v3(0.2f);
    V3 kd = v3(0.35f, 0.12f, 0.48f);
    Lambertian matte;
    PlasticX micro;
    explicit PlasticMaterial(const PlasticX& m) : micro(m) {}
    V3 F(const V3& wo, const V3& wi, const V2& uu) const {
        if (uu[0] < 0.5f) { V2 _uu = uu; _uu[0] *= 2; return kd * matte.F(wo, wi, _uu); }
        return ks * micro.F(wo, wi, uu);
    }
    float PDF(const V3& wo, const V3& wi, const V2& uu) const {
        if (uu[0] < 0.5f) return matte.PDF(wo, wi, uu);
        return micro.PDF(wo, wi, uu);
    }
    V3 S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const {
        V2 _uu_ = uu;
        if (_uu_[0] < 0.5f) { _uu_[0] *= 2; return kd * matte.S_F(wo, wi, _uu_, pdf); }
        _uu_[0] -= 0.5f; _uu_[0] *= 2.0f;
        return ks * micro.S_F(wo, wi, _uu_, pdf);
    }
};
inline PlasticMaterial createPlasticMaterial() {
    return PlasticMaterial(PlasticX{v3(1.0f), FresnelDielectric{1.5f}, Beckmann(0.01f, 0.1f)});
}
// MicrofacetBXDF.h:532-586
struct GlassMaterial {
    V3 kr = v3(0.98f), kt = v3(0.98f);
    MicrofacetReflection<Beckmann, FresnelDielectric> _mr;
    MicrofacetTransmission<Beckmann> _mt;
    float ratio = 0.25f;
    GlassMaterial(const FresnelDielectric& fr, const Beckmann& dist)
        : _mr{v3(0.98f), fr, dist}, _mt(v3(0.98f), dist, 1.0f, fr.eta, TransportMode::Importance) {}
    V3 F(const V3& wo, const V3& wi, const V2& uu) const {
        if (uu[0] < ratio) { V2 _uu = uu; _uu[0] = uu[0] / ratio; return _mr.F(wo, wi, _uu); }
        V2 _uu = uu; _uu[0] = (uu[0] - ratio) / (1.0f - ratio); return _mt.F(wo, wi, _uu);
    }
    float PDF(const V3& wo, const V3& wi, const V2& uu) const {
        if (uu[0] < ratio) { V2 _uu = uu; _uu[0] = uu[0] / ratio; return ratio * _mr.PDF(wo, wi, _uu); }
        V2 _uu = uu; _uu[0] = (uu[0] - ratio) / (1.0f - ratio); return (1 - ratio) * _mt.PDF(wo, wi, _uu);
    }
    V3 S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const {
        if (uu[0] < ratio) { V2 _uu = uu; _uu[0] = uu[0] / ratio; return _mr.S_F(wo, wi, _uu, pdf); }
        V2 _uu = uu; _uu[0] = (uu[0] - ratio) / (1.0f - ratio); return _mt.S_F(wo, wi, _uu, pdf);
    }
};
inline GlassMaterial createGlass() { return GlassMaterial(FresnelDielectric{1.5f}, Beckmann(0.01f, 0.01f)); }

// ---------------------------------------------------------------- Texture.hh:17-43, Material.hh:44-146
// Noise / Image need Metal's sampler and noise(); no material on the path uses them:
// Image with a null texture returns albedo (Texture.hh:29-31); Noise is resolved to albedo too.
inline V3 texture_value(const trc_TextureInfo& ti, V2 uv) {
    const V3 albedo = v3(ti.albedo);
    switch (ti.type) {
        case TRC_TEX_CONSTANT: return albedo;
        case TRC_TEX_CHECKER: {
            float sines = m_sin(8 * PI_F * uv.x) * m_cos(PI_F / 2 + 4 * PI_F * uv.y);
            return albedo * (0.5f * (sines < 0 ? 0.0f : 1.0f) + 0.5f);   // step(0, sines)
        }
        case TRC_TEX_IMAGE: return albedo;
        case TRC_TEX_NOISE: return albedo;
        default: return v3(1.0f);
    }
}
template <typename Bx> inline V3 mat_F(const trc_Material& m, const Bx& bx, const V3& wo, const V3& wi, const V2& uv, float& pdf, const V2& uu) {
    V3 color = texture_value(m.textureInfo, uv);
    pdf = bx.PDF(wo, wi, uu);
    return color * bx.F(wo, wi, uu);
}
template <typename Bx> inline V3 mat_S_F(const trc_Material& m, const Bx& bx, const V3& wo, V3& wi, const V2& uv, const V2& uu, float& pdf) {
    V3 color = texture_value(m.textureInfo, uv);
    return color * bx.S_F(wo, wi, uu, pdf);
}
// the Lambertian returns a scalar; broadcast it like Metal does (float3 * float)
struct LambertianRGB {
    Lambertian l;
    V3 F(const V3& wo, const V3& wi, const V2& uu) const { return v3(l.F(wo, wi, uu)); }
    float PDF(const V3& wo, const V3& wi, const V2& uu) const { return l.PDF(wo, wi, uu); }
    V3 S_F(const V3& wo, V3& wi, const V2& uu, float& pdf) const { return v3(l.S_F(wo, wi, uu, pdf)); }
};
inline V3 Material_F(const trc_Material& m, const V3& wo, const V3& wi, const V2& uv, float& pdf, const V2& uu) {
    switch (m.type) {
        case TRC_MAT_LAMBERT: return mat_F(m, LambertianRGB{}, wo, wi, uv, pdf, uu);
        case TRC_MAT_METAL: return mat_F(m, createMetalMaterial(), wo, wi, uv, pdf, uu);
        case TRC_MAT_PLASTIC: return mat_F(m, createPlasticMaterial(), wo, wi, uv, pdf, uu);
        case TRC_MAT_GLASS: return mat_F(m, createGlass(), wo, wi, uv, pdf, uu);
        default: return v3(0);
    }
}
inline float Material_PDF(const trc_Material& m, const V3& wo, const V3& wi, const V2& uu) {
    switch (m.type) {
        case TRC_MAT_LAMBERT: return LambertianRGB{}.PDF(wo, wi, uu);
        case TRC_MAT_METAL: return createMetalMaterial().PDF(wo, wi, uu);
        case TRC_MAT_PLASTIC: return createPlasticMaterial().PDF(wo, wi, uu);
        case TRC_MAT_GLASS: return createGlass().PDF(wo, wi, uu);
        default: return 0;
    }
}
inline V3 Material_S_F(const trc_Material& m, const V3& wo, V3& wi, const V2& uv, const V2& uu, float& pdf) {
    switch (m.type) {
        case TRC_MAT_LAMBERT: return mat_S_F(m, LambertianRGB{}, wo, wi, uv, uu, pdf);
        case TRC_MAT_METAL: return mat_S_F(m, createMetalMaterial(), wo, wi, uv, uu, pdf);
        case TRC_MAT_PLASTIC: return mat_S_F(m, createPlasticMaterial(), wo, wi, uv, uu, pdf);
        case TRC_MAT_GLASS: return mat_S_F(m, createGlass(), wo, wi, uv, uu, pdf);
        default: return v3(0);
    }
}

// Spectrum.hh:186-190 (only Y is used)
inline float RGBToY(const V3& rgb) { return 0.212671f * rgb[0] + 0.715160f * rgb[1] + 0.072169f * rgb[2]; }

// ---------------------------------------------------------------- Camera.hh:59-69
inline Ray castRay(const trc_Camera* camera, float s, float t, RandomSampler* xsampler) {
    V2 disk = xsampler->sampleUnitInDisk();
    V2 rd = V2{camera->lenRadius * disk.x, camera->lenRadius * disk.y};
    V3 offset = v3(camera->u) * rd.x + v3(camera->v) * rd.y;
    V3 origin = v3(camera->lookFrom) + offset;
    V3 sample = v3(camera->cornerLowLeft) + v3(camera->horizontal) * s + v3(camera->vertical) * t;
    return Ray(origin, sample - origin);
}

// ---------------------------------------------------------------- integrators
struct Env {
    const trc_Material* materials; V3 ambient;
    const trc_GridDensityInfo* densityInfo = nullptr; const float* densityArray = nullptr;
    const float* envmap = nullptr; uint32_t env_w = 0, env_h = 0;      // equirectangular RGB float image (texHDR)
};
// Render.hh:42-48 + `texHDR.sample(textureSampler, uv)` (Render.metal:100-105,301-304,435-438, Photon.metal:29-33):
// linear filter, clamp-to-edge (Common.hh:11).  Metal's filtering weights are implementation-defined; this is the
// textbook bilinear lookup at texel centres, evaluated with Lerp's operation order.  Without a map: env.ambient.
inline V3 env_radiance(const Env& env, const V3& direction) {
    if (!env.envmap) return env.ambient;
    const V3 v = normalize(direction);
    float u = m_atan2(v.z, v.x) * 0.1591f + 0.5f;
    float w = m_asin(v.y) * 0.3183f + 0.5f;
    const float x = u * (float)env.env_w - 0.5f, y = w * (float)env.env_h - 0.5f;
    const float fx0 = floorf(x), fy0 = floorf(y);
    const float fx = x - fx0, fy = y - fy0;
    auto clampi = [](float f, uint32_t n) { return f < 0.0f ? 0u : (f > (float)(n - 1) ? n - 1 : (uint32_t)f); };
    const uint32_t x0 = clampi(fx0, env.env_w), x1 = clampi(fx0 + 1.0f, env.env_w);
    const uint32_t y0 = clampi(fy0, env.env_h), y1 = clampi(fy0 + 1.0f, env.env_h);
    auto texel = [&](uint32_t xx, uint32_t yy) { const float* t = env.envmap + 3 * ((size_t)yy * env.env_w + xx); return v3(t[0], t[1], t[2]); };
    const V3 top = (1 - fx) * texel(x0, y0) + fx * texel(x1, y0);
    const V3 bot = (1 - fx) * texel(x0, y1) + fx * texel(x1, y1);
    return (1 - fy) * top + fy * bot;
}

// Render.metal:411-492
template <class XSampler>
V3 tracePath(int depth, Ray& ray, XSampler& xsampler, const Env& env, Scene& scene, Counters* cnt) {
    HitRecord hitRecord;
    V3 ratio = v3(1.0f);
    V3 color = v3(0.0f);
    bool hitted = scene.hit(ray, hitRecord, FLT_MAX);
    do {
        if (!hitted) { color = color + ratio * env_radiance(env, ray.direction); break; }
        const trc_Material& mat = env.materials[hitRecord.material];
        if (mat.type == TRC_MAT_DIFFUSE) {
            V3 le = v3(mat.textureInfo.albedo);
            float w = dot(-ray.direction, -hitRecord.gn);
            return ratio * le * fabsf(w);
        }
        V2 uu = xsampler.sample2D();
        const V3 hit_origin = hitRecord.p;
        V3 _origin = offset_ray(hitRecord.p, hitRecord.sn);
        V3 nx, ny;
        CoordinateSystem(hitRecord.sn, nx, ny);
        // stw = {nx, ny, sn} (columns); wts = transpose(stw)
        V3 minus_d = -ray.direction;
        V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
        V3 wi = v3(0);
        float bxPDF = 0;   // uninitialised in the reference (B-3)
        if (cnt) cnt->shaded++;
        V3 attenuation = Material_S_F(mat, wo, wi, hitRecord.uv, uu, bxPDF);
        if (bxPDF <= 0) break;
        if (wi.z < 0) {   // transmission
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(offset_ray(hit_origin, -hitRecord.sn), wiw);
        } else {
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(_origin, wiw);
        }
        ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
        {   // Russian roulette
            float p = RGBToY(ratio);
            if (xsampler.random() > p) break;
            ratio = ratio * (1.0f / p);
        }
        hitted = scene.hit(ray, hitRecord, FLT_MAX);
    } while ((--depth) > 0);
    return color;
}

// Render.metal:277-409
template <class XSampler>
V3 traceMIS(int depth, Ray& ray, XSampler& xsampler, const Env& env, Scene& scene, Counters* cnt) {
    HitRecord hitRecord;
    V3 scat_attenuation = v3(0); float scat_bxPDF = 1.0f;
    V3 ratio = v3(1.0f);
    V3 color = v3(0.0f);
    const trc_scene& prims = scene.prims;
    bool hitted = scene.hit(ray, hitRecord, FLT_MAX);
    do {
        if (!hitted) { color = color + ratio * env_radiance(env, ray.direction); break; }
        const trc_Material& mat = env.materials[hitRecord.material];
        if (mat.type == TRC_MAT_DIFFUSE) {
            V3 le = v3(mat.textureInfo.albedo);
            float w = dot(-ray.direction, -hitRecord.gn);
            return ratio * le * fabsf(w);
        }
        LightSampleRecord lsr;
        V2 uu = xsampler.sample2D();
        const V3 hit_origin = hitRecord.p;
        V3 _origin = offset_ray(hitRecord.p, hitRecord.sn);
        if (xsampler.random() < 0.5f) square_sample(prims.squareList[5], uu, _origin, lsr);
        else square_sample(prims.squareList[6], uu, _origin, lsr);
        V3 _dir = lsr.p - _origin;
        V3 _nor = normalize(_dir);
        V3 nx, ny;
        CoordinateSystem(hitRecord.sn, nx, ny);
        const float _tr = 1.0f;
        const float _dis = length(_dir);
        const Ray _ray(_origin, _nor);
        HitRecord shr;
        const bool blocked = scene.hit(_ray, shr, _dis, true);
        V3 minus_d = -ray.direction;
        if (!blocked) {   // light sampling
            V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
            V3 wi = v3(dot(nx, _ray.direction), dot(ny, _ray.direction), dot(hitRecord.sn, _ray.direction));
            float bxPDF = 0;
            if (cnt) cnt->shaded++;
            V3 weight = Material_F(mat, wo, wi, hitRecord.uv, bxPDF, uu);
            float cosOnLight = fabsf(dot(lsr.n, -_nor));
            V3 Li = v3(env.materials[lsr.material].textureInfo.albedo);
            weight = weight * (Li * cosOnLight);
            float dist2 = _dis * _dis;
            float liPDF = dist2 * lsr.areaPDF / cosOnLight;
            weight = weight * PowerHeuristic(1, liPDF, 1, bxPDF);
            color = color + _tr * ratio * weight / liPDF;
        }
        // BXDF sampling
        V3 wi = v3(0);
        float bxPDF = 0;
        V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
        if (cnt) cnt->shaded++;
        scat_attenuation = Material_S_F(mat, wo, wi, hitRecord.uv, uu, bxPDF);
        scat_bxPDF = bxPDF;
        if (bxPDF <= 0) break;
        if (wi.z < 0) {
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(offset_ray(hit_origin, -hitRecord.sn), wiw);
        } else {
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(_origin, wiw);
        }
        ratio = ratio * (scat_attenuation / scat_bxPDF);
        {
            float p = RGBToY(ratio);
            if (xsampler.random() > p) break;
            ratio = ratio * (1.0f / p);
        }
        hitted = scene.hit(ray, hitRecord, FLT_MAX);
        if (hitted && env.materials[hitRecord.material].type == TRC_MAT_DIFFUSE) {
            V3 Li = v3(env.materials[hitRecord.material].textureInfo.albedo);
            float cosOnLight = dot(-ray.direction, hitRecord.sn);
            V3 weight = scat_attenuation * Li * cosOnLight;
            V3 d = hitRecord.p - ray.origin;
            float dist2 = dot(d, d);
            float lightPDF = hitRecord.PDF * dist2 / cosOnLight;
            weight = weight * PowerHeuristic(1, scat_bxPDF, 1, lightPDF);
            color = color + ratio * weight / scat_bxPDF;
            break;
        }
    } while ((--depth) > 0);
    return color;
}

// ---------------------------------------------------------------- participating media (traceVolume only)
// HitRecord.hh:45-79
inline float PhaseHG(float cosTheta, float g) {
    float gg = g * g;
    float denom = 1 + gg + 2 * g * cosTheta;
    return (0.25f / PI_F) * (1 - gg) / (denom * sqrtf(denom));
}
// Sampling.hh:40-43
inline V3 SphericalDirection(float sinTheta, float cosTheta, float phi, const V3& x, const V3& y, const V3& z) {
    return (sinTheta * m_cos(phi) * x + sinTheta * m_sin(phi) * y) + cosTheta * z;
}
inline float HG_Sample_p(float g, const V3& wo, V3& wi, const V2& uu) {     // HitRecord.hh:58-77
    float cosTheta;
    if (fabsf(g) < 1e-3f) cosTheta = 1 - 2 * uu[0];
    else {
        float gg = g * g;
        float sqrTerm = (1 - gg) / (1 + g - 2 * g * uu[0]);
        cosTheta = -(1 + gg - sqrTerm * sqrTerm) / (2 * g);
    }
    float sinTheta = sqrtf(fmaxf(0.0f, 1 - cosTheta * cosTheta));
    float phi = 2 * PI_F * uu[1];
    V3 v1, v2;
    CoordinateSystem(wo, v1, v2);
    wi = SphericalDirection(sinTheta, cosTheta, phi, v1, v2, wo);
    return PhaseHG(cosTheta, g);
}
struct MediumInteraction { V3 p = v3(0); float phaseG = 0; bool sampled = false; };   // Medium.hh:14-22
// HomogeneousMedium::Sample, Medium.hh:38-74, as constructed at Render.metal:118: (0.02, 0.08, 0.5)
inline V3 homogeneous_sample(const Ray& ray, const HitRecord& hitRecord, MediumInteraction& mi, RandomSampler& xsampler) {
    const V3 sigma_a = v3(0.02f), sigma_s = v3(0.08f), sigma_t = sigma_s + sigma_a;
    const float g = 0.5f;
    const int nSamples = 3;
    int channel = (int)(xsampler.sample1D() * nSamples);
    if (channel > nSamples - 1) channel = nSamples - 1;
    float dist = -m_log(1 - xsampler.sample1D()) / sigma_t[channel];
    float t = fminf(dist, hitRecord.t);
    bool sampledMedium = t < hitRecord.t;
    const float tt = fminf(t, FLT_MAX);
    V3 Tr = v3(m_exp(-sigma_t.x * tt), m_exp(-sigma_t.y * tt), m_exp(-sigma_t.z * tt));
    V3 density = Tr, result = Tr;
    if (sampledMedium) {
        mi.p = ray.pointAt(t);
        mi.phaseG = g;
        mi.sampled = true;
        density = density * sigma_t;
        result = result * sigma_s;
    }
    float pdf = dot(v3(1.0f), density);
    if (0.0f >= pdf) pdf = 1.0f; else pdf = pdf / nSamples;
    return result / pdf;
}
// GridDensityMedium, Medium.hh:111-199
inline float grid_D(const trc_GridDensityInfo& info, const float* density, int x, int y, int z) {
    const int nx = (int)info.nx, ny = (int)info.ny, nz = (int)info.nz;
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0;
    return density[((size_t)z * ny + y) * nx + x];
}
inline float Lerp(float t, float s1, float s2) { return (1 - t) * s1 + t * s2; }       // Sampling.hh:13-16
inline float grid_Density(const trc_GridDensityInfo& info, const float* density, const V3& p) {
    const float nx = (float)info.nx, ny = (float)info.ny, nz = (float)info.nz;
    V3 pSamples = v3(p.x * nx - 0.5f, p.y * ny - 0.5f, p.z * nz - 0.5f);
    const float fx = floorf(pSamples.x), fy = floorf(pSamples.y), fz = floorf(pSamples.z);
    // (int3) floor(...): out-of-int-range / NaN coordinates are clamped far outside the grid (Metal's conversion
    // saturates; a plain C cast would be undefined)
    auto to_int = [](float f) { return !(f > -2.0e9f) ? (int)-2000000000 : (f > 2.0e9f ? (int)2000000000 : (int)f); };
    const int ix = to_int(fx), iy = to_int(fy), iz = to_int(fz);
    V3 d = v3(pSamples.x - (float)ix, pSamples.y - (float)iy, pSamples.z - (float)iz);
    float d00 = Lerp(d.x, grid_D(info, density, ix, iy, iz), grid_D(info, density, ix + 1, iy, iz));
    float d10 = Lerp(d.x, grid_D(info, density, ix, iy + 1, iz), grid_D(info, density, ix + 1, iy + 1, iz));
    float d01 = Lerp(d.x, grid_D(info, density, ix, iy, iz + 1), grid_D(info, density, ix + 1, iy, iz + 1));
    float d11 = Lerp(d.x, grid_D(info, density, ix, iy + 1, iz + 1), grid_D(info, density, ix + 1, iy + 1, iz + 1));
    float d0 = Lerp(d.y, d00, d10);
    float d1 = Lerp(d.y, d01, d11);
    return Lerp(d.z, d0, d1);
}
constexpr int kGridSampleMaxSteps = 1 << 16;   // the reference's `while (true)` (Medium.hh:179) is unbounded; a stale
                                               // hitRecord._t or a zero majorant would spin forever on a GPU
inline float grid_sample(const Env& env, const HitRecord& hitRecord, MediumInteraction& mi, RandomSampler& sampler) {
    if (!env.densityInfo || !env.densityArray) return 1.0f;
    const trc_GridDensityInfo& info = *env.densityInfo;
    const Ray& ray = hitRecord._r;
    float tMax = hitRecord._t;
    float t = 0;
    for (int step = 0; step < kGridSampleMaxSteps; ++step) {
        t -= m_log(1 - sampler.sample1D()) * info.invMaxDensity / info.sigma_t;
        if (t >= tMax) break;
        V3 p = ray.pointAt(t);
        if (grid_Density(info, env.densityArray, p) * info.invMaxDensity > sampler.sample1D()) {
            mi.p = hitRecord.modelMatrix ? mul_point(*hitRecord.modelMatrix, p) : p;
            mi.phaseG = info.g;
            mi.sampled = true;
            return info.sigma_s / info.sigma_t;
        }
    }
    return 1.0f;
}

// Render.metal:78-275
V3 traceVolume(int depth, Ray& ray, RandomSampler& xsampler, const Env& env, Scene& scene, Counters* cnt) {
    HitRecord hitRecord;
    V3 scat_attenuation = v3(0); float scat_bxPDF = 1.0f;
    V3 ratio = v3(1.0f);
    V3 color = v3(0.0f);
    const trc_scene& prims = scene.prims;
    bool hitted = scene.hit(ray, hitRecord, FLT_MAX);
    do {
        if (!hitted) { color = color + ratio * env_radiance(env, ray.direction); break; }
        if (env.materials[hitRecord.material].type == TRC_MAT_DIFFUSE) {
            V3 le = v3(env.materials[hitRecord.material].textureInfo.albedo);
            float w = dot(-ray.direction, -hitRecord.gn);
            return ratio * le * fabsf(w);
        }
        MediumInteraction mi;
        if (ray.medium == TRC_MEDIUM_HOMOGENEOUS) ratio = ratio * homogeneous_sample(ray, hitRecord, mi, xsampler);
        else if (ray.medium == TRC_MEDIUM_GRIDDENSITY) ratio = ratio * v3(grid_sample(env, hitRecord, mi, xsampler));
        bool need_test = false, need_bsdf = false;
        if (mi.sampled) {
            V3 wi, wo = -ray.direction;
            HG_Sample_p(mi.phaseG, wo, wi, xsampler.sample2D());
            ray.update(mi.p, wi);
            ray.medium = env.materials[hitRecord.material].medium;
            need_test = true;
        } else {
            if (env.materials[hitRecord.material].type == TRC_MAT_NIL) {
                if (dot(ray.direction, hitRecord.gn) < 0) {                     // enter
                    ray = Ray(offset_ray(hitRecord.p, -hitRecord.gn), ray.direction);
                    ray.medium = env.materials[hitRecord.material].medium;
                } else {                                                        // depart
                    ray = Ray(offset_ray(hitRecord.p, hitRecord.gn), ray.direction);
                    ray.medium = TRC_MEDIUM_NIL;
                }
                need_test = true;
            } else {
                need_test = false;
                need_bsdf = true;
            }
        }
        if (need_test) hitted = scene.hit(ray, hitRecord, FLT_MAX);
        if (!need_bsdf) continue;

        const trc_Material& mat = env.materials[hitRecord.material];
        LightSampleRecord lsr;
        V2 uu = xsampler.sample2D();
        const V3 hit_origin = hitRecord.p;
        V3 _origin = offset_ray(hitRecord.p, hitRecord.sn);
        if (xsampler.random() < 0.5f) square_sample(prims.squareList[5], uu, _origin, lsr);
        else square_sample(prims.squareList[6], uu, _origin, lsr);
        V3 _dir = lsr.p - _origin;
        V3 _nor = normalize(_dir);
        V3 nx, ny;
        CoordinateSystem(hitRecord.sn, nx, ny);
        const float _tr = 1.0f;
        const float _dis = length(_dir);
        const Ray _ray(_origin, _nor);
        HitRecord shr;
        const bool blocked = scene.hit(_ray, shr, _dis, true);
        V3 minus_d = -ray.direction;
        if (!blocked) {   // light sampling
            V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
            V3 wi = v3(dot(nx, _ray.direction), dot(ny, _ray.direction), dot(hitRecord.sn, _ray.direction));
            float bxPDF = 0;
            if (cnt) cnt->shaded++;
            V3 weight = Material_F(mat, wo, wi, hitRecord.uv, bxPDF, uu);
            float cosOnLight = fabsf(dot(lsr.n, -_nor));
            V3 Li = v3(env.materials[lsr.material].textureInfo.albedo);
            weight = weight * (Li * cosOnLight);
            float dist2 = _dis * _dis;
            float liPDF = dist2 * lsr.areaPDF / cosOnLight;
            weight = weight * PowerHeuristic(1, liPDF, 1, bxPDF);
            color = color + _tr * ratio * weight / liPDF;
        }
        // BXDF sampling
        V3 wi = v3(0);
        float bxPDF = 0;
        V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
        if (cnt) cnt->shaded++;
        scat_attenuation = Material_S_F(mat, wo, wi, hitRecord.uv, uu, bxPDF);
        scat_bxPDF = bxPDF;
        if (bxPDF <= 0) break;
        if (wi.z < 0) {   // transmission
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(offset_ray(hit_origin, -hitRecord.sn), wiw);
            if (dot(wiw, hitRecord.gn) < 0) ray.medium = mat.medium;            // enter
            else ray.medium = TRC_MEDIUM_NIL;                                   // depart
        } else {
            V3 wiw = (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z;
            ray.update(_origin, wiw);
        }
        ratio = ratio * (scat_attenuation / scat_bxPDF);
        {
            float p = RGBToY(ratio);
            if (xsampler.random() > p) break;
            ratio = ratio * (1.0f / p);
        }
        hitted = scene.hit(ray, hitRecord, FLT_MAX);
        if (hitted && env.materials[hitRecord.material].type == TRC_MAT_DIFFUSE) {
            V3 Li = v3(env.materials[hitRecord.material].textureInfo.albedo);
            float cosOnLight = dot(-ray.direction, hitRecord.sn);
            V3 weight = scat_attenuation * Li * cosOnLight;
            V3 d = hitRecord.p - ray.origin;
            float dist2 = dot(d, d);
            float lightPDF = hitRecord.PDF * dist2 / cosOnLight;
            weight = weight * PowerHeuristic(1, scat_bxPDF, 1, lightPDF);
            color = color + ratio * weight / scat_bxPDF;
            break;
        }
    } while ((--depth) > 0);
    return color;
}

// density grid of the GridDensity medium for orc_render (the oracle keeps one, like PackageEnv ids 3/4)
static const float* g_envmap = nullptr;
static uint32_t g_env_w = 0, g_env_h = 0;
static const trc_GridDensityInfo* g_density_info = nullptr;
static const float* g_density_array = nullptr;
static trc_GridDensityInfo g_density_info_copy;

// one pixel of kernelPathTracing, Render.metal:495-558, for `spp` successive frames
void render_pixel(const trc_scene& prims, const trc_Camera* camera, const Env& env, uint32_t W, uint32_t H,
                  uint32_t x, uint32_t y, uint32_t* rng_rgba, float* accum_rgba, const trc_params& prm, Counters* cnt) {
    uint32_t* texel = rng_rgba + 4 * ((size_t)y * W + x);
    float* px = accum_rgba + 4 * ((size_t)y * W + x);
    uint32_t rr = texel[0], gg = texel[1], bb = texel[2], aa = texel[3];
    V3 cached = v3(px[0], px[1], px[2]);
    Scene scene{prims, cnt};
    for (uint32_t s = 0; s < prm.spp; ++s) {
        uint64_t rng_state = ((uint64_t)rr << 32) | gg;
        uint64_t rng_inc = ((uint64_t)bb << 32) | aa;
        pcg32_t rng = {rng_inc, rng_state};     // aggregate order {state, inc}: the words trade roles (B-1)
        uint32_t frame = prm.frame0 + s;
        float u = (float)x / (float)W;          // no sub-pixel jitter (B-2)
        const uint32_t vh = (prm.view_height != 0 && prm.view_height < H) ? prm.view_height : H;   // stacked views
        float v = (float)(y % vh) / (float)vh;
        RandomSampler rs{&rng};
        Ray ray = castRay(camera, u, v, &rs);
        V3 color;
        if ((prm.flags & TRC_FLAG_SOBOL) && prm.integrator != TRC_INTEGRATOR_VOLUME) {
            // Render.metal:529-530 (commented out there): built AFTER castRay, from a copy of rng; tracePath(8, ray, ss, ...)
            SobolSampler ss(rng, frame, x, y % vh, W, vh);
            color = (prm.integrator == TRC_INTEGRATOR_MIS) ? traceMIS((int)prm.max_depth, ray, ss, env, scene, cnt)
                                                           : tracePath((int)prm.max_depth, ray, ss, env, scene, cnt);
        } else
        color = (prm.integrator == TRC_INTEGRATOR_VOLUME) ? traceVolume((int)prm.max_depth, ray, rs, env, scene, cnt)
                   : (prm.integrator == TRC_INTEGRATOR_MIS)  ? traceMIS((int)prm.max_depth, ray, rs, env, scene, cnt)
                                                             : tracePath((int)prm.max_depth, ray, rs, env, scene, cnt);
        bool bad = std::isinf(color.x) || std::isnan(color.x) || std::isinf(color.y) || std::isnan(color.y) ||
                   std::isinf(color.z) || std::isnan(color.z);
        if (bad) color = v3(0);
        V3 result = (cached * (float)frame + color) / (float)(frame + 1);
        cached = result;
        gg = (uint32_t)rng.state; rr = (uint32_t)(rng.state >> 32);
        aa = (uint32_t)rng.inc;   bb = (uint32_t)(rng.inc >> 32);
    }
    px[0] = cached.x; px[1] = cached.y; px[2] = cached.z; px[3] = 1.0f;
    texel[0] = rr; texel[1] = gg; texel[2] = bb; texel[3] = aa;
}

// ================================================================ SPPM (Photon.metal, Photon.hh)
// Photon.hh:57-69
inline float ph_mod(float x, float y) { return x - y * floorf(x / y); }
// Photon.hh:71-89: float-only Lehmer-style hash of a cell index into [0, N*N)
inline float ph_hash(const V3 idx, const float HashScale, const float BufInfo) {
    const float HashNum = BufInfo * BufInfo;
    float n[4] = {idx.x, idx.y, idx.z, idx.x + idx.y - idx.z};
    const float q[4] = {1225.0f, 1585.0f, 2457.0f, 2098.0f};
    const float r[4] = {1112.0f, 367.0f, 92.0f, 265.0f};
    const float a[4] = {3423.0f, 2646.0f, 1707.0f, 1999.0f};
    const float m[4] = {4194287.0f, 4194277.0f, 4194191.0f, 4194167.0f};
    float nm[4];
    for (int k = 0; k < 4; ++k) {
        float nk = n[k] * 4194304.0f / HashScale;
        float beta = floorf(nk / q[k]);
        float pk = a[k] * (nk - beta * q[k]) - beta * r[k];
        float sgn = (-pk > 0.0f) ? 1.0f : ((-pk < 0.0f) ? -1.0f : 0.0f);      // sign(-p)
        beta = (sgn + 1.0f) * 0.5f * m[k];
        nk = pk + beta;
        nm[k] = nk / m[k];
    }
    float d = ((nm[0] * 1.0f + nm[1] * -1.0f) + nm[2] * 1.0f) + nm[3] * -1.0f;   // dot(n/m, (1,-1,1,-1))
    float fr = d - floorf(d);                                                       // fract
    return floorf(fr * HashNum);
}
// Sampling.hh:55-60
inline V3 UniformSampleHemisphere(const V2& u) {
    float z = u[0];
    float r = sqrtf(fmaxf(0.0f, 1.0f - z * z));
    float phi = 2 * PI_F * u[1];
    return v3(r * m_cos(phi), r * m_sin(phi), z);
}

struct CamRec {     // Photon.hh:30-53
    V3 ratio = v3(1), position = v3(0), direction = v3(0);
    bool valid = false;
    V3 alternative = v3(0), flux = v3(0);
    float radius = 0;
    uint32_t photonCount = 0;
    void reset() { ratio = v3(1); position = v3(0); direction = v3(0); valid = false; flux = v3(0); radius = 0; photonCount = 0; }
};
inline float canon_nan(float x) { uint32_t q = 0x7FC00000u; float n; std::memcpy(&n, &q, 4); return x != x ? n : x; }
inline V3 canon_nan(V3 v) { return v3(canon_nan(v.x), canon_nan(v.y), canon_nan(v.z)); }
struct PhoRec {     // Photon.hh:12-28
    V3 flux = v3(1), normal = v3(0), position = v3(0), direction = v3(0);
    uint8_t step = 0;
    bool active = false;
    void reset() { flux = v3(1); step = 0; active = false; }
};

// Photon.metal:3-92
bool traceCameraRecord(int depth, Ray& ray, RandomSampler& xsampler, CamRec& cr, const Env& env, Scene& scene) {
    HitRecord hitRecord;
    cr.valid = false;
    V3 ratio = v3(1.0f);
    bool hitted = scene.hit(ray, hitRecord, FLT_MAX);
    do {
        if (!hitted) { cr.alternative = ratio * env_radiance(env, ray.direction); return false; }
        const trc_Material& material = env.materials[hitRecord.material];
        if (material.type == TRC_MAT_DIFFUSE) {
            V3 le = v3(material.textureInfo.albedo);
            float w = dot(-ray.direction, -hitRecord.gn);
            cr.alternative = ratio * le * fabsf(w);
            return false;
        }
        if (!material.specular) {
            cr.valid = true; cr.ratio = ratio; cr.position = hitRecord.p; cr.direction = ray.direction;
            return true;
        }
        V3 nx, ny;
        CoordinateSystem(hitRecord.sn, nx, ny);
        V3 wi = v3(0); float bxPDF = 0;
        V3 minus_d = -ray.direction;
        V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
        V2 uu = xsampler.sample2D();
        V3 attenuation = Material_S_F(material, wo, wi, hitRecord.uv, uu, bxPDF);
        if (bxPDF <= 0) break;
        V3 pn = hitRecord.sn * copysignf(1.0f, wi.z);
        V3 _origin = offset_ray(hitRecord.p, pn);
        ray.update(_origin, (nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z);
        ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
        if (std::isinf(ratio.x) || std::isinf(ratio.y) || std::isinf(ratio.z) ||
            std::isnan(ratio.x) || std::isnan(ratio.y) || std::isnan(ratio.z)) ratio = v3(1.0f);
        hitted = scene.hit(ray, hitRecord, FLT_MAX);
    } while ((--depth) > 0);
    cr.alternative = v3(0);
    return false;
}

// Photon.metal:220-285
void tracePhotonRecord(Ray& ray, RandomSampler& xsampler, PhoRec& pr, const Env& env, Scene& scene) {
    HitRecord hitRecord;
    V3 ratio = v3(1.0f);
    bool hitted = scene.hit(ray, hitRecord, FLT_MAX);
    const trc_Material& material = env.materials[hitRecord.material];
    if (!hitted || material.type == TRC_MAT_DIFFUSE) { pr.reset(); return; }
    V3 nx, ny;
    CoordinateSystem(hitRecord.sn, nx, ny);
    V3 wi = v3(0); float bxPDF = 0;
    V3 minus_d = -ray.direction;
    V3 wo = v3(dot(nx, minus_d), dot(ny, minus_d), dot(hitRecord.sn, minus_d));
    V2 uu = xsampler.sample2D();
    V3 attenuation = Material_S_F(material, wo, wi, hitRecord.uv, uu, bxPDF);
    if (bxPDF <= 0) { pr.reset(); return; }
    ratio = ratio * (attenuation / fmaxf(FLT_EPSILON, bxPDF));
    {
        float p = RGBToY(ratio);
        if (xsampler.random() > p) { pr.reset(); return; }
        ratio = ratio * (1.0f / p);
    }
    // A degenerate BSDF sample leaves wi = NaN (bxPDF NaN passes `<= 0`, Photon.metal:268-272), and the SIGN of a NaN is nobody's to
    // define: x86 generates negative ones, gfx950 positive ones, Metal whatever the GPU does, and each propagates operand signs its own
    // way.  The record exposes it twice -- copysign(1, wi.z) picks the side of the surface, and direction / flux are stored raw -- so
    // both restatements fix it the same way: a NaN counts as positive, and a stored NaN is the canonical 0x7FC00000 (declared: DESIGN 6).
    V3 pn = hitRecord.sn * (std::isnan(wi.z) ? 1.0f : copysignf(1.0f, wi.z));
    pr.position = offset_ray(hitRecord.p, pn);
    pr.normal = pn;
    pr.direction = canon_nan((nx * wi.x + ny * wi.y) + hitRecord.sn * wi.z);
    pr.flux = canon_nan(pr.flux * ratio);
    pr.step += 1;
    pr.active = !material.specular;
}

struct SppmState {
    uint32_t W = 0, H = 0;
    std::vector<CamRec> cam;
    std::vector<PhoRec> pho;
    std::vector<uint32_t> photon_rng;        // 512*512*4
    std::vector<int32_t> mark;               // winning photon index per cell, -1 = empty
    std::vector<uint32_t> count;
    // Complex (Camera.hh:27-55)
    uint32_t frame_count = 0;
    V3 box_min = v3(0), box_max = v3(0), box_size = v3(0);
    float initial_radius = 0, hash_scale = 0, total_photon_sum = 0;
    uint32_t frame_photon_sum = 0;
};
const uint32_t kHashN = TRC_PHOTON_HASHN;

inline pcg32_t toRNG(const uint32_t* t) {          // Render.hh:96-107 (same word swap as the path kernel, B-1)
    uint64_t rng_state = ((uint64_t)t[0] << 32) | t[1];
    uint64_t rng_inc = ((uint64_t)t[2] << 32) | t[3];
    return pcg32_t{rng_inc, rng_state};
}
inline void exRNG(const pcg32_t& rng, uint32_t* t) {   // Render.hh:109-120
    t[0] = (uint32_t)(rng.state >> 32); t[1] = (uint32_t)rng.state;
    t[2] = (uint32_t)(rng.inc >> 32);   t[3] = (uint32_t)rng.inc;
}

// The per-pixel and per-photon passes touch only their own record / texel / pixel, so they run over contiguous index
// bands on all host cores; every item computes what the serial loop computed (results do not depend on the thread
// count).  ORC_THREADS caps the pool.
template <typename F>
void sppm_parallel(size_t n, F&& body) {
    unsigned T = std::max(1u, std::thread::hardware_concurrency());
    if (const char* e = std::getenv("ORC_THREADS")) T = std::max(1, std::atoi(e));
    T = (unsigned)std::min<size_t>(T, std::max<size_t>(1, n / 256));
    if (T <= 1) { body((size_t)0, n); return; }
    std::vector<std::thread> pool;
    const size_t unit = (n + T - 1) / T;
    for (unsigned t = 0; t < T; ++t) {
        const size_t b = std::min(n, t * unit), e = std::min(n, b + unit);
        if (b < e) pool.emplace_back([&body, b, e]() { body(b, e); });
    }
    for (auto& th : pool) th.join();
}

// kernelCameraRecording, Photon.metal:96-167
void sppm_camera_pass(SppmState& st, const trc_scene& prims, const trc_Camera* camera, const Env& env, uint32_t* canvas_rng) {
    Scene scene{prims, nullptr};
    sppm_parallel(st.H, [&](size_t y0, size_t y1) {
    for (uint32_t y = (uint32_t)y0; y < (uint32_t)y1; ++y)
        for (uint32_t x = 0; x < st.W; ++x) {
            uint32_t* texel = canvas_rng + 4 * ((size_t)y * st.W + x);
            pcg32_t rng = toRNG(texel);
            float u = (float)x / (float)st.W, v = (float)y / (float)st.H;
            RandomSampler rs{&rng};
            Ray ray = castRay(camera, u, v, &rs);
            CamRec& slot = st.cam[(size_t)y * st.W + x];
            CamRec cr = slot;
            int depth = 8;
            if (st.frame_count == 0) { cr.reset(); depth = 3; }
            traceCameraRecord(depth, ray, rs, cr, env, scene);
            slot = cr;
            exRNG(rng, texel);
        }
    });
}

// kernelCameraReducing + kernelPhotonParams + kernelPhotonRadius, Photon.metal:169-218,357-384
void sppm_prepare_params(SppmState& st) {
    V3 lo = v3(FLT_MAX), hi = v3(-FLT_MAX);        // AABB over the valid camera records (min/max are exact)
    for (const CamRec& c : st.cam)
        if (c.valid) {
            lo = v3(fminf(lo.x, c.position.x), fminf(lo.y, c.position.y), fminf(lo.z, c.position.z));
            hi = v3(fmaxf(hi.x, c.position.x), fmaxf(hi.y, c.position.y), fmaxf(hi.z, c.position.z));
        }
    st.box_min = lo; st.box_max = hi;
    st.box_size = hi - lo;
    st.initial_radius = dot(st.box_size, v3(1.0f / 3.0f));
    st.initial_radius *= 2.5f / (1 << 12);
    st.box_min = st.box_min - v3(st.initial_radius);
    st.box_max = st.box_max + v3(st.initial_radius);
    st.hash_scale = 1.0f / (st.initial_radius * 1.5f);
    for (CamRec& c : st.cam) c.radius = st.initial_radius;
}

// kernelPhotonRecording, Photon.metal:286-355
void sppm_photon_pass(SppmState& st, const trc_scene& prims, const Env& env) {
    Scene scene{prims, nullptr};
    sppm_parallel((size_t)kHashN * kHashN, [&](size_t i0, size_t i1) {
    for (uint32_t idx = (uint32_t)i0; idx < (uint32_t)i1; ++idx) {
        PhoRec pc = st.pho[idx];
        uint32_t* texel = &st.photon_rng[4 * (size_t)idx];
        pcg32_t rng = toRNG(texel);
        RandomSampler rs{&rng};
        Ray ray;
        bool check = (st.frame_count == 0) || (pc.step == 0) || (pc.step == 8);
        if (check) {   // ray from the light source
            pc.reset();
            LightSampleRecord lsr;
            V2 uu = rs.sample2D();
            V3 _origin = v3(450, 250, 250);
            if (rs.random() < 1) square_sample(prims.squareList[5], uu, _origin, lsr);
            else square_sample(prims.squareList[6], uu, _origin, lsr);
            pc.flux = v3(env.materials[lsr.material].textureInfo.albedo) * 100000.0f;
            V3 nx, ny;
            CoordinateSystem(lsr.n, nx, ny);
            uu = rs.sample2D();
            V3 h = UniformSampleHemisphere(uu);
            ray = Ray(lsr.p, (nx * h.x + ny * h.y) + lsr.n * h.z);
        } else {
            ray = Ray(pc.position, pc.direction);
        }
        tracePhotonRecord(ray, rs, pc, env, scene);
        st.pho[idx] = pc;
        exRNG(rng, texel);
    }
    });
}

// kernelPhotonHashing + the point raster PhotonMarkVS/FS + kernelPhotonSumming, Photon.metal:386-496.
// Raster semantics: a point at the integer corner (hx, hy) lands in texel (hx-1, hy-1) -- the texel
// kernelPhotonRefine reads back (:556-558); the last primitive in API order wins the mark (= the
// highest photon index), the count blends additively; inactive photons are clipped (z = -1).
void sppm_hash_pass(SppmState& st) {
    std::fill(st.mark.begin(), st.mark.end(), -1);
    std::fill(st.count.begin(), st.count.end(), 0u);
    for (uint32_t idx = 0; idx < kHashN * kHashN; ++idx) {
        const PhoRec& p = st.pho[idx];
        V3 HashIndex = (p.position - st.box_min) * st.hash_scale;
        HashIndex = v3(floorf(HashIndex.x), floorf(HashIndex.y), floorf(HashIndex.z));
        float hashed = ph_hash(HashIndex, st.hash_scale, (float)kHashN);
        float hx = ph_mod(hashed, (float)kHashN), hy = floorf(hashed / (float)kHashN);
        if (!p.active) continue;
        float tx = hx - 1.0f, ty = hy - 1.0f;
        if (!(tx >= 0.0f && tx < (float)kHashN && ty >= 0.0f && ty < (float)kHashN)) continue;
        uint32_t cell = (uint32_t)ty * kHashN + (uint32_t)tx;
        st.mark[cell] = std::max(st.mark[cell], (int32_t)idx);
        st.count[cell] += 1;
    }
    uint32_t sum = 0;
    for (uint32_t c : st.count) if (c > 0) sum += std::max(1u, c);
    st.frame_photon_sum += sum;
}

// kernelPhotonRefine, Photon.metal:498-623
void sppm_refine_pass(SppmState& st, float* accum) {
    const float fN = (float)kHashN;
    sppm_parallel((size_t)st.W * st.H, [&](size_t p0, size_t p1) {
    for (size_t i = p0; i < p1; ++i) {
        CamRec& c = st.cam[i];
        float* px = accum + 4 * i;
        const float frame = (float)st.frame_count, frame1 = (float)(st.frame_count + 1);
        V3 cache = v3(px[0], px[1], px[2]);
        if (!c.valid) {
            V3 result = (cache * frame + c.alternative) / frame1;
            px[0] = result.x; px[1] = result.y; px[2] = result.z; px[3] = 1.0f;
            continue;
        }
        V3 QueryPosition = c.position, QueryDirection = c.direction, QueryFlux = c.flux, QueryReflectance = c.ratio;
        float QueryRadius = c.radius;
        uint32_t QueryPhotonCount = c.photonCount;
        V3 BBoxMin = st.box_min;
        float HashScale = st.hash_scale;
        V3 RangeMin = vabs(QueryPosition - v3(QueryRadius) - BBoxMin) * HashScale;
        V3 RangeMax = vabs(QueryPosition + v3(QueryRadius) - BBoxMin) * HashScale;
        V3 _Flux = v3(0); uint32_t _PhotonCount = 0;
        for (int iz = (int)RangeMin.z; iz <= (int)RangeMax.z; iz++)
            for (int iy = (int)RangeMin.y; iy <= (int)RangeMax.y; iy++)
                for (int ix = (int)RangeMin.x; ix <= (int)RangeMax.x; ix++) {
                    V3 hashIndex = v3((float)ix, (float)iy, (float)iz);
                    float hashed = ph_hash(hashIndex, HashScale, fN);
                    float hx = ph_mod(hashed, fN) - 1.0f, hy = floorf(hashed / fN) - 1.0f;
                    // _marksHashGrid.read((uint2)(hx, hy)): out-of-range reads return 0 -> photon (0,0), count 0
                    int32_t winner; float Correction;
                    if (hx >= 0.0f && hx < fN && hy >= 0.0f && hy < fN) {
                        uint32_t cell = (uint32_t)hy * kHashN + (uint32_t)hx;
                        winner = st.mark[cell];
                        Correction = (float)st.count[cell];
                        if (winner < 0) continue;                                    // PhotonIndex2D.x < 0
                    } else { winner = 0; Correction = 0.0f; }
                    const PhoRec& ph = st.pho[(size_t)winner];
                    V3 _RangeMin = hashIndex / HashScale + BBoxMin;
                    V3 _RangeMax = (hashIndex + v3(1.0f)) / HashScale + BBoxMin;
                    if ((_RangeMin.x < ph.position.x) && (ph.position.x < _RangeMax.x) &&
                        (_RangeMin.y < ph.position.y) && (ph.position.y < _RangeMax.y) &&
                        (_RangeMin.z < ph.position.z) && (ph.position.z < _RangeMax.z)) {
                        float d = length(ph.position - QueryPosition);
                        if ((d < QueryRadius) && (-dot(QueryDirection, ph.direction) > 0.001f)) {
                            _Flux = _Flux + ph.flux * Correction;
                            _PhotonCount = (uint32_t)((float)_PhotonCount + Correction);
                        }
                    }
                }
        _Flux = _Flux * (QueryReflectance / 3.141592f);
        const float alpha = 0.8f;
        float g = fminf(((float)QueryPhotonCount + (float)_PhotonCount * alpha) / (float)(QueryPhotonCount + _PhotonCount), 1.0f);
        QueryRadius = QueryRadius * sqrtf(g);
        QueryPhotonCount = (uint32_t)((float)QueryPhotonCount + (float)_PhotonCount * alpha);
        QueryFlux = (QueryFlux + _Flux) * g;
        c.flux = QueryFlux; c.radius = QueryRadius; c.photonCount = QueryPhotonCount;
        float TotalPhotonNum = st.total_photon_sum;
        TotalPhotonNum += (float)st.frame_photon_sum;
        V3 color = QueryFlux / (QueryRadius * QueryRadius * 3.141592f * TotalPhotonNum);
        V3 result = (cache * frame + color) / frame1;
        if (std::isnan(result.x) || std::isnan(result.y) || std::isnan(result.z)) result = v3(0);
        px[0] = result.x; px[1] = result.y; px[2] = result.z; px[3] = 1.0f;
    }
    });
}

inline void fill_hit(trc_hit& o, bool hit, const HitRecord& rec, float tmax_after, uint32_t nd, uint32_t nr, uint32_t nl) {
    memset(&o, 0, sizeof o);
    o.hit = hit ? 1 : 0;
    o.pType = hit ? rec.pType : -1;
    (void)tmax_after;
    if (hit) {
        o.pIndex = rec.pIndex; o.t = rec.t;
        o.p[0] = rec.p.x; o.p[1] = rec.p.y; o.p[2] = rec.p.z;
        o.gn[0] = rec.gn.x; o.gn[1] = rec.gn.y; o.gn[2] = rec.gn.z;
        o.sn[0] = rec.sn.x; o.sn[1] = rec.sn.y; o.sn[2] = rec.sn.z;
        o.uv[0] = rec.uv.x; o.uv[1] = rec.uv.y;
        o.material = rec.material; o.PDF = rec.PDF;
    }
    o.n_descend = nd; o.n_return = nr; o.n_leaf = nl;
}

}  // namespace

// ================================================================ C API
extern "C" {

int orc_uses_libm(void) {
#ifdef ORACLE_USE_LIBM
    return 1;
#else
    return 0;
#endif
}

void orc_pcg32_srandom(uint64_t* state, uint64_t* inc, uint64_t initstate, uint64_t initseq) {
    pcg32_t r; pcg32_srandom_r(&r, initstate, initseq); *state = r.state; *inc = r.inc;
}
uint32_t orc_pcg32_random(uint64_t* state, uint64_t inc) {
    pcg32_t r{*state, inc}; uint32_t v = pcg32_random_r(&r); *state = r.state; return v;
}
float orc_randomF(uint64_t* state, uint64_t inc) {
    pcg32_t r{*state, inc}; float v = randomF(&r); *state = r.state; return v;
}

void orc_trace_rays(const trc_scene* scene, const trc_ray* rays, size_t n, trc_hit* out, int any_hit) {
    Scene sc{*scene, nullptr};
    for (size_t i = 0; i < n; ++i) {
        Ray ray(v3a(rays[i].origin), v3a(rays[i].direction));
        HitRecord rec;
        uint32_t nd = 0, nr = 0, nl = 0;
        bool h = sc.hit(ray, rec, rays[i].tmax, any_hit != 0, &nd, &nr, &nl);
        fill_hit(out[i], h, rec, 0, nd, nr, nl);
    }
}

void orc_trace_rays_brute(const trc_scene* scene, const trc_ray* rays, size_t n, trc_hit* out) {
    Scene sc{*scene, nullptr};
    for (size_t i = 0; i < n; ++i) {
        Ray ray(v3a(rays[i].origin), v3a(rays[i].direction));
        HitRecord rec;
        V2 range_t = V2{FLT_MIN, rays[i].tmax};
        uint32_t nl = 0;
        for (uint32_t k = 0; k < scene->n_bvh; ++k) {
            if (scene->bvhList[k].pType == TRC_PRIM_BVH) continue;
            nl++;
            sc.leaf_test(k, ray, range_t, rec);
        }
        fill_hit(out[i], range_t.y < rays[i].tmax, rec, 0, 0, 0, nl);
    }
}

// fragmentShader, Render.metal:29-75
void orc_tonemap(const float* accum, uint32_t W, uint32_t H, uint8_t* rgba8, float* exposure_out) {
    const size_t n = (size_t)W * H;
    uint64_t sum[3] = {0, 0, 0};
    for (size_t i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) {
            float v = accum[4 * i + c];
            if (!(v > 0.0f)) v = 0.0f;                       // negative / NaN radiance does not expose
            if (v > 1048576.0f) v = 1048576.0f;
            sum[c] += (uint64_t)(v * 65536.0f + 0.5f);
        }
    float mean[3];
    for (int c = 0; c < 3; ++c) mean[c] = (float)((double)sum[c] / 65536.0 / (double)n);
    const float luma = (mean[0] * 0.2126f + mean[1] * 0.7152f) + mean[2] * 0.0722f;
    float mapped = 1 - m_exp(-1.0f * luma);                  // CETone(luma, 1.)
    mapped = fminf(fmaxf(mapped, 0.0f), 0.9999f);
    const float expose = 1.0f - mapped;
    if (exposure_out) *exposure_out = expose;
    const float A = 2.51f, B = 0.03f, Cc = 2.43f, D = 0.59f, E = 0.14f;
    for (uint32_t y = 0; y < H; ++y)
        for (uint32_t x = 0; x < W; ++x) {
            const float* px = accum + 4 * ((size_t)(H - 1 - y) * W + x);
            uint8_t* o = rgba8 + 4 * ((size_t)y * W + x);
            for (int c = 0; c < 3; ++c) {
                float col = px[c] * expose;
                float t = (col * (A * col + B)) / (col * (Cc * col + D) + E);      // ACESTone
                if (!(t > 0.0f)) t = 0.0f;
                if (t > 1.0f) t = 1.0f;
                o[c] = (uint8_t)(t * 255.0f + 0.5f);
            }
            o[3] = 255;
        }
}

float orc_hg_sample(float g, const float wo[3], const float uu[2], float wi_out[3]) {      // HitRecord.hh:58-77
    V3 wi;
    const float pdf = HG_Sample_p(g, v3a(wo), wi, V2{uu[0], uu[1]});
    wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z;
    return pdf;
}
float orc_phase_hg(float cosTheta, float g) { return PhaseHG(cosTheta, g); }                  // HitRecord.hh:45-49
float orc_grid_density(const trc_GridDensityInfo* info, const float* density, const float p[3]) {   // Medium.hh:129-145
    return grid_Density(*info, density, v3a(p));
}

void orc_set_environment_map(uint32_t w, uint32_t h, const float* rgb) {
    if (rgb && w && h) { g_envmap = rgb; g_env_w = w; g_env_h = h; } else { g_envmap = nullptr; g_env_w = g_env_h = 0; }
}

void orc_set_density(const trc_GridDensityInfo* info, const float* density) {
    if (info && density) { g_density_info_copy = *info; g_density_info = &g_density_info_copy; g_density_array = density; }
    else { g_density_info = nullptr; g_density_array = nullptr; }
}

void orc_render(const trc_scene* scene, const trc_Camera* camera, const float env_rgb[3], uint32_t W, uint32_t H,
                uint32_t* rng_rgba, float* accum_rgba, const trc_params* params, trc_stats* stats, int n_threads) {
    Env env{scene->materials, v3(env_rgb[0], env_rgb[1], env_rgb[2])};
    env.densityInfo = g_density_info; env.densityArray = g_density_array;
    env.envmap = g_envmap; env.env_w = g_env_w; env.env_h = g_env_h;
    const uint32_t nranks = params->tile_nranks ? params->tile_nranks : 1;
    unsigned T = n_threads > 0 ? (unsigned)n_threads : std::max(1u, std::thread::hardware_concurrency());
    T = std::min<unsigned>(T, H ? H : 1);
    std::vector<Counters> counters(T);
    std::vector<uint64_t> paths(T, 0);
    auto work = [&](unsigned tid) {
        // contiguous row bands, the last one takes the remainder (RT_Weekend main.swift:77-87 shape)
        uint32_t unit = H / T, y0 = tid * unit, y1 = (tid == T - 1) ? H : y0 + unit;
        for (uint32_t y = y0; y < y1; ++y)
            for (uint32_t x = 0; x < W; ++x) {
                uint32_t tx = x / TRC_TILE, ty = y / TRC_TILE;
                if ((tx + ty) % nranks != params->tile_rank) continue;
                render_pixel(*scene, camera, env, W, H, x, y, rng_rgba, accum_rgba, *params, &counters[tid]);
                paths[tid] += params->spp;
            }
    };
    if (T == 1) work(0);
    else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < T; ++t) pool.emplace_back(work, t);
        for (auto& th : pool) th.join();
    }
    if (stats) {
        for (unsigned t = 0; t < T; ++t) {
            const Counters& c = counters[t];
            stats->paths += paths[t];
            stats->rays += c.rays; stats->shaded += c.shaded;
            stats->n_descend += c.n_descend; stats->n_return += c.n_return;
            stats->n_leaf_sphere += c.n_leaf[0]; stats->n_leaf_square += c.n_leaf[1];
            stats->n_leaf_cube += c.n_leaf[2]; stats->n_leaf_triangle += c.n_leaf[3];
            stats->n_hit_triangle += c.n_hit_triangle; stats->n_hit_cube += c.n_hit_cube;
        }
        stats->launches += 1;
    }
}

void orc_material_S_F(const trc_Material* m, const float wo[3], const float uv[2], const float uu[2],
                      float wi_out[3], float f_out[3], float* pdf_out) {
    V3 wi = v3(0); float pdf = 0;
    V3 f = Material_S_F(*m, v3a(wo), wi, V2{uv[0], uv[1]}, V2{uu[0], uu[1]}, pdf);
    wi_out[0] = wi.x; wi_out[1] = wi.y; wi_out[2] = wi.z;
    f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z;
    *pdf_out = pdf;
}
void orc_material_F(const trc_Material* m, const float wo[3], const float wi[3], const float uv[2], const float uu[2],
                    float f_out[3], float* pdf_out) {
    float pdf = 0;
    V3 f = Material_F(*m, v3a(wo), v3a(wi), V2{uv[0], uv[1]}, pdf, V2{uu[0], uu[1]});
    f_out[0] = f.x; f_out[1] = f.y; f_out[2] = f.z;
    *pdf_out = pdf;
}
float orc_material_PDF(const trc_Material* m, const float wo[3], const float wi[3], const float uu[2]) {
    return Material_PDF(*m, v3a(wo), v3a(wi), V2{uu[0], uu[1]});
}

void orc_offset_ray(const float p[3], const float n[3], float out[3]) {
    V3 r = offset_ray(v3a(p), v3a(n)); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_fr_dielectric(float cosi, float eta) { return FrDielectric(cosi, eta); }
void orc_fr_conductor(float cosi, const float eta[3], const float k[3], float out[3]) {
    V3 r = FrConductor(cosi, v3a(eta), v3a(k)); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
uint64_t orc_sobol_interval_to_index(uint32_t m, uint64_t sample_index, uint32_t px, uint32_t py) {
    return SobolIntervalToIndex(m, sample_index, px, py);
}
float orc_sobol_sample_float(uint64_t index, uint32_t dimension) { return SobolSampleFloat(index, dimension); }
float orc_sobol_sample_dimension(uint32_t frame, uint32_t x, uint32_t y, uint32_t w, uint32_t h, uint32_t dimension) {
    pcg32_t none{0, 1};
    SobolSampler ss(none, frame, x, y, w, h);
    return ss.SampleDimension(ss.mSobolIndex, dimension);
}
float orc_power_heuristic(int nf, float fPdf, int ng, float gPdf) { return PowerHeuristic(nf, fPdf, ng, gPdf); }
void orc_cosine_sample_hemisphere(const float u[2], float out[3]) {
    V3 r = CosineSampleHemisphere(V2{u[0], u[1]}); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
float orc_erf(float x) { return Erf(x); }
float orc_erfinv(float x) { return ErfInv(x); }
int orc_aabb_hit_t(const trc_AABB* box, const trc_ray* ray, float tmin, float tmax, float* t_out) {
    Ray r(v3a(ray->origin), v3a(ray->direction));
    float t = tmax;
    bool h = aabb_hit_t(*box, r, V2{tmin, tmax}, t);
    *t_out = t;
    return h ? 1 : 0;
}
void orc_cast_ray(const trc_Camera* cam, float s, float t, uint64_t* state, uint64_t inc, float origin_out[3], float dir_out[3]) {
    pcg32_t rng{*state, inc};
    RandomSampler rs{&rng};
    Ray r = castRay(cam, s, t, &rs);
    *state = rng.state;
    origin_out[0] = r.origin.x; origin_out[1] = r.origin.y; origin_out[2] = r.origin.z;
    dir_out[0] = r.direction.x; dir_out[1] = r.direction.y; dir_out[2] = r.direction.z;
}
float orc_math(int fn, float a, float b) {
    switch (fn) {
        case 0: return m_sin(a); case 1: return m_cos(a); case 2: return m_exp(a); case 3: return m_log(a);
        case 4: return m_pow(a, b); case 5: return m_asin(a); case 6: return m_acos(a); case 7: return m_atan2(a, b);
        default: return 0.0f;
    }
}

// ---------------------------------------------------------------- SPPM C API
struct orc_sppm { SppmState st; };

orc_sppm* orc_sppm_create(uint32_t W, uint32_t H, uint64_t photon_seed) {
    orc_sppm* s = new orc_sppm();
    s->st.W = W; s->st.H = H;
    s->st.cam.resize((size_t)W * H);
    s->st.pho.resize((size_t)kHashN * kHashN);
    s->st.photon_rng.resize((size_t)kHashN * kHashN * 4);
    s->st.mark.assign((size_t)kHashN * kHashN, -1);
    s->st.count.assign((size_t)kHashN * kHashN, 0);
    for (uint64_t p = 0; p < (uint64_t)kHashN * kHashN; ++p) {      // same as trc_host_fill_rng
        pcg32_t r; pcg32_srandom_r(&r, photon_seed, p);
        for (int c = 0; c < 4; ++c) s->st.photon_rng[4 * p + c] = pcg32_random_r(&r);
    }
    return s;
}
void orc_sppm_destroy(orc_sppm* s) { delete s; }

// `photon:` (AAPLRenderer.mm:1077-1086) for n_frames frames
void orc_sppm_frames(orc_sppm* s, const trc_scene* scene, const trc_Camera* camera, const float env_rgb[3],
                     uint32_t* canvas_rng, float* accum, uint32_t n_frames) {
    SppmState& st = s->st;
    Env env{scene->materials, v3(env_rgb[0], env_rgb[1], env_rgb[2])};
    env.envmap = g_envmap; env.env_w = g_env_w; env.env_h = g_env_h;
    for (uint32_t f = 0; f < n_frames; ++f) {
        if (st.frame_count == 0) {              // photonPrepare (view != nil)
            sppm_camera_pass(st, *scene, camera, env, canvas_rng);
            sppm_prepare_params(st);
        }
        if (st.frame_count % 2) sppm_camera_pass(st, *scene, camera, env, canvas_rng);   // photonPrepare:nil
        sppm_photon_pass(st, *scene, env);
        sppm_hash_pass(st);
        sppm_refine_pass(st, accum);
        st.total_photon_sum += (float)st.frame_photon_sum;      // completion handler, :1031-1036
        st.frame_photon_sum = 0;
        st.frame_count += 1;
    }
}

void orc_sppm_download(const orc_sppm* s, trc_CameraRecord* cam, trc_PhotonRecord* pho, float* mark, float* count,
                       trc_Complex* cx) {
    const SppmState& st = s->st;
    auto put = [](trc_float3& d, const V3& v) { d.x = v.x; d.y = v.y; d.z = v.z; d._pad = 0; };
    if (cam) for (size_t i = 0; i < st.cam.size(); ++i) {
        const CamRec& c = st.cam[i]; trc_CameraRecord& o = cam[i];
        memset(&o, 0, sizeof o);
        put(o.ratio, c.ratio); put(o.position, c.position); put(o.direction, c.direction);
        o.valid = c.valid; put(o.alternative, c.alternative); put(o.flux, c.flux);
        o.radius = c.radius; o.photonCount = c.photonCount;
    }
    if (pho) for (size_t i = 0; i < st.pho.size(); ++i) {
        const PhoRec& p = st.pho[i]; trc_PhotonRecord& o = pho[i];
        memset(&o, 0, sizeof o);
        put(o.flux, p.flux); put(o.normal, p.normal); put(o.position, p.position); put(o.direction, p.direction);
        o.step = p.step; o.active = p.active;
    }
    if (mark) for (size_t c = 0; c < st.mark.size(); ++c) {
        if (st.mark[c] < 0) { mark[4 * c] = mark[4 * c + 1] = mark[4 * c + 2] = mark[4 * c + 3] = -1.0f; }
        else {
            mark[4 * c] = (float)(st.mark[c] % kHashN); mark[4 * c + 1] = (float)(st.mark[c] / kHashN);
            mark[4 * c + 2] = (float)(c % kHashN); mark[4 * c + 3] = (float)(c / kHashN);
        }
    }
    if (count) for (size_t c = 0; c < st.count.size(); ++c) count[c] = (float)st.count[c];
    if (cx) {
        memset(cx, 0, sizeof *cx);
        cx->frame_count = st.frame_count;
        put(cx->photonBox.mini, st.box_min); put(cx->photonBox.maxi, st.box_max); put(cx->photonBoxSize, st.box_size);
        cx->photonInitialRadius = st.initial_radius; cx->photonHashScale = st.hash_scale;
        cx->totalPhotonSum = st.total_photon_sum; cx->framePhotonSum = st.frame_photon_sum;
        cx->tex_size.x = cx->view_size.x = (float)st.W; cx->tex_size.y = cx->view_size.y = (float)st.H;
    }
}
float orc_photon_hash(const float idx[3], float hash_scale) { return ph_hash(v3a(idx), hash_scale, (float)kHashN); }

}  // extern "C"
