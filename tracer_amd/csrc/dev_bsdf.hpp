// dev_bsdf.hpp -- BSDF evaluation / sampling on the device, in the local shading frame (z = normal).
//
// Reference (RT_Metal/Metal/): Sampling.hh:18-203, BXDF.hh:24-81, BXDF.metal:3-34,
// MatteBXDF.hh:6-22, MicrofacetBXDF.h:7-586, Texture.hh:17-43, Material.hh:77-146.
// The reference hard-codes every lobe parameter in its create*() functions
// (MicrofacetBXDF.h:438-455,514-530,576-586); they are compile-time constants here, so the
// material switch carries no per-material parameter loads beyond type + albedo.
#pragma once

#include "dev_prof.hpp"
#include "dev_vec.hpp"

namespace trcdev {

// ---------------------------------------------------------------- Sampling.hh
TRC_DEV void coordinate_system(const F3 a, F3& b, F3& c) {          // Sampling.hh:18-34
    if (fabsf(a.x) > fabsf(a.y)) b = f3(-a.z, 0, a.x);
    else b = f3(0, a.z, -a.y);
    b = normalize(b);
    c = cross(a, b);
}
TRC_DEV F2 concentric_sample_disk(const F2 u) {                      // Sampling.hh:79-99
    F2 uo; uo.x = 2.f * u.x - 1; uo.y = 2.f * u.y - 1;
    F2 r0; r0.x = 0; r0.y = 0;
    if (uo.x == 0 && uo.y == 0) return r0;
    const float PiOver2 = kPi / 2.0f, PiOver4 = kPi / 4.0f;
    // one quotient for both cases (half the lanes of a wavefront take each: as two branches they ran one after the other)
    const bool wide = fabsf(uo.x) > fabsf(uo.y);
    const float r = wide ? uo.x : uo.y;
    const float q = (wide ? uo.y : uo.x) / r;
    const float theta = wide ? PiOver4 * q : PiOver2 - PiOver4 * q;
    float s, c;
    dm_sincosf(theta, &s, &c);
    F2 out; out.x = r * c; out.y = r * s;
    return out;
}
TRC_DEV F3 cosine_sample_hemisphere(const F2 u) {                    // Sampling.hh:125-129
    F2 d = concentric_sample_disk(u);
    float z = sqrt_cr(fmaxf(0.0f, 1.0f - d.x * d.x - d.y * d.y));
    return f3(d.x, d.y, z);
}
TRC_DEV float power_heuristic(int nf, float fPdf, int ng, float gPdf) {   // Sampling.hh:137-140
    float f = nf * fPdf, g = ng * gPdf;
    return (f * f) / (f * f + g * g);
}
// Sampling.hh:148-203
TRC_DEV float cos_theta(F3 w) { return w.z; }
TRC_DEV float cos2_theta(F3 w) { return w.z * w.z; }
TRC_DEV float abs_cos_theta(F3 w) { return fabsf(w.z); }
TRC_DEV float sin2_theta(F3 w) { return fmaxf(0.0f, 1.0f - cos2_theta(w)); }
TRC_DEV float sin_theta(F3 w) { return sqrt_cr(sin2_theta(w)); }
TRC_DEV float tan_theta(F3 v) {
    float temp = 1 - v.z * v.z;
    if (temp <= 0.0f || v.z == 0.0f) return 0.0f;
    return sqrt_cr(temp) / v.z;
}
TRC_DEV float tan2_theta(F3 v) {
    float zz = v.z * v.z;
    float temp = 1 - zz;
    if (temp <= 0.0f || zz == 0.0f) return 0.0f;
    return temp / zz;
}
TRC_DEV float cos_phi(F3 w) { float s = sin_theta(w); return (s == 0) ? 1 : clampf(w.x / s, -1.0f, 1.0f); }
TRC_DEV float sin_phi(F3 w) { float s = sin_theta(w); return (s == 0) ? 0 : clampf(w.y / s, -1.0f, 1.0f); }
TRC_DEV float cos2_phi(F3 w) { float r = cos_phi(w); return r * r; }
TRC_DEV float sin2_phi(F3 w) { float r = sin_phi(w); return r * r; }
TRC_DEV float sqr(float v) { return v * v; }
// cos_phi and sin_phi of one direction: two quotients by the same sin_theta.  TRC_PHI_PAIR: reciprocal + Newton step once,
// the quotients' own corrections, operand guards as one wave-uniform range test (dev_vec.hpp: GuardedDivBy / div_core) --
// the same bits as the two divisions (tests/test_gpu_divby.py), which is what everybody computes when some lane's operand is
// outside [2^-60, 2^60] (a direction along the normal: sin_theta == 0)
struct Phi { float c, s; };
// st = sin_theta(w), which the callers share with tan_theta (the same square root: 1 - z * z where that is positive)
TRC_DEV float tan_theta_st(F3 v, float st) {
    float temp = 1 - v.z * v.z;
    if (temp <= 0.0f || v.z == 0.0f) return 0.0f;
    return st / v.z;
}
TRC_DEV Phi phi_of(F3 w, float st) {
    Phi ph;
#if TRC_WAVE_GUARDS
    const GuardedDivBy by = guarded_div_by(st);
    float qc = div_core(w.x, by), qs = div_core(w.y, by);
    const float small = fmin3(fabsf(w.x), fabsf(w.y), st), large = fmax3(fabsf(w.x), fabsf(w.y), st);
    if (__builtin_expect(!wave_all(small >= 0x1p-60f && large <= 0x1p60f), 0)) { qc = w.x / st; qs = w.y / st; }
    ph.c = (st == 0) ? 1 : clampf(qc, -1.0f, 1.0f);
    ph.s = (st == 0) ? 0 : clampf(qs, -1.0f, 1.0f);
#else
    ph.c = (st == 0) ? 1 : clampf(w.x / st, -1.0f, 1.0f);
    ph.s = (st == 0) ? 0 : clampf(w.y / st, -1.0f, 1.0f);
#endif
    return ph;
}
TRC_DEV Phi phi_of(F3 w) { return phi_of(w, sin_theta(w)); }

// ---------------------------------------------------------------- Math.hh:118-167
TRC_DEV float erf_inv(float x) {
    float w, p;
    x = clampf(x, -.99999f, .99999f);
    w = -dm_logf_pos((1 - x) * (1 + x));             // the clamp above leaves (1 - x)(1 + x) in [2e-5, 1], whatever x was
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
        w = sqrt_cr(w) - 3;
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
TRC_DEV float erf_approx(float x) {
    const float a1 = 0.254829592f, a2 = -0.284496736f, a3 = 1.421413741f, a4 = -1.453152027f, a5 = 1.061405429f;
    const float p = 0.3275911f;
    int sign = 1;
    if (x < 0) sign = -1;
    x = fabsf(x);
    float t = rcp1(1 + p * x);
    float y = 1 - (((((a5 * t + a4) * t) + a3) * t + a2) * t + a1) * t * dm_expf(-x * x);
    return sign * y;
}

// ---------------------------------------------------------------- BXDF.hh / BXDF.metal
TRC_DEV F3 reflect(F3 wo, F3 n) { return -wo + 2 * dot(wo, n) * n; }             // BXDF.hh:24-26
TRC_DEV bool refract(F3 wo, F3 n, float eta, F3& wi) {                            // BXDF.hh:28-41 (cos from wo.z, B-6)
    float cosThetaI = wo.z;
    float sin2ThetaI = fmaxf(0.0f, 1.0f - cosThetaI * cosThetaI);
    float sin2ThetaT = eta * eta * sin2ThetaI;
    if (sin2ThetaT >= 1) return false;
    float cosThetaT = sqrt_cr(1 - sin2ThetaT);
    wi = eta * -wo + (eta * cosThetaI - cosThetaT) * n;
    return true;
}
TRC_DEV float fr_dielectric(float cosi, float eta) {                              // BXDF.metal:3-22
    cosi = clampf(cosi, -1.0f, 1.0f);
    bool entering = cosi > 0.f;
    if (!entering) { eta = rcp1(eta); cosi = -cosi; }
    float sin2Theta_i = 1 - cosi * cosi;
    float sin2Theta_t = sin2Theta_i / sqr(eta);
    if (sin2Theta_t >= 1) return 1.f;
    float cosTheta_t = sqrt_cr(fmaxf(0.0f, 1 - sin2Theta_t));
    float r_parl = (eta * cosi - cosTheta_t) / (eta * cosi + cosTheta_t);
    float r_perp = (cosi - eta * cosTheta_t) / (cosi + eta * cosTheta_t);
    return (r_parl * r_parl + r_perp * r_perp) / 2;
}
template <bool METAL_CONSTANTS = false>
TRC_DEV F3 fr_conductor(float cosi, F3 eta, F3 k) {                               // BXDF.metal:24-34
    F3 tmp = (eta * eta + k * k) * cosi * cosi;
    const F3 n1 = tmp - (2.f * eta * cosi) + f3(1), d1 = tmp + (2.f * eta * cosi) + f3(1);
    F3 tmp_f = eta * eta + k * k;
    const F3 n2 = tmp_f - (2.f * eta * cosi) + f3(cosi * cosi), d2 = tmp_f + (2.f * eta * cosi) + f3(cosi * cosi);
#if TRC_WAVE_GUARDS
    if (METAL_CONSTANTS) {
    // MetalMaterial's constants (eta <= 0.81, k = 1) and 0 <= cosi <= 2 put all twelve operands into [0.5, 16]:
    // n1 = (eta^2 + 1) c^2 - 2 eta c + 1 >= 1 / (eta^2 + 1), n2 = (c - eta)^2 + 1 >= 1, the denominators are sums of
    // non-negative terms and 1 -- the six quotients need no operand guard beyond that ONE compare (which a NaN fails)
    F3 Rparl2 = div_core(n1, d1), Rperp2 = div_core(n2, d2);
    if (__builtin_expect(!wave_all(cosi >= 0.0f && cosi <= 2.0f), 0)) { Rparl2 = n1 / d1; Rperp2 = n2 / d2; }
    return 0.5f * (Rparl2 + Rperp2);
    }
#endif
    F3 Rparl2 = n1 / d1, Rperp2 = n2 / d2;
    return 0.5f * (Rparl2 + Rperp2);
}

// Fresnel policies
struct FrCond {   // BXDF.hh:59-70; MetalMaterial: eta = (0.18, 0.15, 0.81), k = 1 (MicrofacetBXDF.h:445-449)
    TRC_DEV static F3 eval(float cosThetaI) { return fr_conductor<true>(fabsf(cosThetaI), f3(0.18f, 0.15f, 0.81f), f3(1.0f)); }
};
struct FrDiel15 { // BXDF.hh:72-81 with eta = 1.5 (Plastic, Glass reflection)
    TRC_DEV static F3 eval(float cosThetaI) { return f3(fr_dielectric(cosThetaI, 1.5f)); }
};

// ---------------------------------------------------------------- microfacet distributions
// alpha is already clamped to >= 0.001 by the constructors (MicrofacetBXDF.h:164,306-309)
// roughness pairs as types (float literals must match the reference's exactly)
struct Alpha_01_02 {
    TRC_DEV static float x() { return 0.01f; }
    TRC_DEV static float y() { return 0.02f; }
    TRC_DEV static DivConst by_x2() { return div_by_sqr001(); }      // max(0.001, x)^2 and its reciprocal
    TRC_DEV static DivConst by_y2() { return div_by_sqr002(); }
};

// Beckmann distribution, MicrofacetBXDF.h:137-290.  The roughness pair is a run-time value (not a template
// parameter) on purpose: Plastic's specular lobe (0.01, 0.1) and Glass's reflection and transmission lobes
// (0.01, 0.01) then run the SAME instructions, so a wavefront holding lanes of several of these lobes executes
// the expensive visible-normal sampling (a Newton iteration over erf^-1 / exp) once instead of once per lobe.
struct Beckmann {
    float alpha_x, alpha_y;
    DivConst by_ax2, by_ay2;         // alpha^2 and its reciprocal (D divides by them)
    // the two roughness pairs the materials use: Plastic's specular lobe (0.01, 0.1), Glass's lobes (0.01, 0.01)
    TRC_DEV static Beckmann make_lobe(bool plastic) {
        Beckmann d; d.alpha_x = fmaxf(0.001f, 0.01f); d.alpha_y = fmaxf(0.001f, plastic ? 0.1f : 0.01f);
        d.by_ax2 = div_by_sqr001();
        const DivConst rough = div_by_sqr01(), smooth = div_by_sqr001();
        d.by_ay2.c = plastic ? rough.c : smooth.c; d.by_ay2.y = plastic ? rough.y : smooth.y;
        return d;
    }
    TRC_DEV float ax() const { return alpha_x; }
    TRC_DEV float ay() const { return alpha_y; }

    TRC_DEV float lambda(F3 w) const {
        const float st = sin_theta(w);
        float absTanTheta = fabsf(tan_theta_st(w, st));
        if (is_inf(absTanTheta)) return 0.;
        const Phi ph = phi_of(w, st);
        float alpha = sqrt_cr((ph.c * ph.c) * ax() * ax() + (ph.s * ph.s) * ay() * ay());
        float a = rcp1(alpha * absTanTheta);
        if (a >= 1.6f) return 0;
        return (1 - 1.259f * a + 0.396f * a * a) / (3.535f * a + 2.181f * a * a);
    }
    TRC_DEV float D(F3 wh) const {
        float tan2Theta = tan2_theta(wh);
        if (is_inf(tan2Theta)) return 0.;
        float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
        const Phi ph = phi_of(wh);
        float qc, qs;                                     // cos2_phi / (ax * ax), sin2_phi / (ay * ay)
        div_const2(ph.c * ph.c, by_ax2, ph.s * ph.s, by_ay2, qc, qs);
        return dm_expf(-tan2Theta * (qc + qs)) /
               (kPi * ax() * ay() * cos4Theta);
    }
    TRC_DEV float G1(F3 w) const { return rcp1(1 + lambda(w)); }
    TRC_DEV float G(F3 wo, F3 wi) const { return rcp1(1 + lambda(wo) + lambda(wi)); }
    TRC_DEV float pdf(F3 wo, F3 wh) const { return D(wh) * G1(wo) * fabsf(dot(wo, wh)) / abs_cos_theta(wo); }

    TRC_DEV static void sample11(float cosThetaI, float sinThetaI, float U1, float U2, float& slope_x, float& slope_y) {
        if (cosThetaI > .9999f) {
            float r = sqrt_cr(-dm_logf(1.0f - U1));
            float sinPhi, cosPhi;
            dm_sincosf(2 * kPi * U2, &sinPhi, &cosPhi);
            slope_x = r * cosPhi;
            slope_y = r * sinPhi;
            return;
        }
        float tanThetaI = sinThetaI / cosThetaI;          // sinThetaI = sqrt(max(0, 1 - cosThetaI^2)) = sin_theta of the caller's vector
        float cotThetaI = rcp1(tanThetaI);
        float a = -1, c = erf_approx(cotThetaI);
        float sample_x = fmaxf(U1, 1e-6f);
        float thetaI = dm_acosf(cosThetaI);
        float fit = 1 + thetaI * (-0.876f + thetaI * (0.4265f - 0.0594f * thetaI));
        // dm_powf(1 - sample_x, fit) with its operands' ranges used: the base lies in {0} U [2^-24, 1), the exponent in [0.44, 1]
        // (fit's minimum over [0, pi]), so log and exp need none of their special cases; a NaN here is replaced by the
        // bisection step below before anybody sees it
        const float base = 1 - sample_x;
        const float pw = dm_expf_fin(fit * dm_logf_pos(base == 0.0f ? 1.0f : base));
        float b = c - (1 + c) * (base == 0.0f ? 0.0f : pw);
        const float SQRT_PI_INV = 1.f / sqrtf(kPi);
        float normalization = rcp1(1 + c + SQRT_PI_INV * tanThetaI * dm_expf(-cotThetaI * cotThetaI));
        int it = 0;
        while (++it < 10) {
            if (!(b >= a && b <= c)) b = 0.5f * (a + c);
            float invErf = erf_inv(b);
            float value = normalization * (1 + b + SQRT_PI_INV * tanThetaI * dm_expf_fin(-invErf * invErf)) - sample_x;      // |erf_inv| < 3.2
            float derivative = normalization * (1 - invErf * tanThetaI);
            if (fabsf(value) < 1e-5f) break;
            if (value > 0) c = b; else a = b;
            b -= value / derivative;
        }
        slope_x = erf_inv(b);
        slope_y = erf_inv(2.0f * fmaxf(U2, 1e-6f) - 1.0f);
    }
    TRC_DEV F3 sample_wh(F3 wo, F2 u) const {
        const bool flip = wo.z < 0;
        const F3 wi = flip ? -wo : wo;
        F3 wiStretched = normalize(f3(ax() * wi.x, ay() * wi.y, wi.z));
        float slope_x, slope_y;
        const float st = sin_theta(wiStretched);
        sample11(cos_theta(wiStretched), st, u.x, u.y, slope_x, slope_y);
        const Phi ph = phi_of(wiStretched, st);
        float tmp = ph.c * slope_x - ph.s * slope_y;
        slope_y = ph.s * slope_x + ph.c * slope_y;
        slope_x = tmp;
        slope_x = ax() * slope_x;
        slope_y = ay() * slope_y;
        F3 wh = normalize(f3(-slope_x, -slope_y, 1.f));
        return flip ? -wh : wh;
    }
};

template <class Alpha>
struct TrowbridgeReitzD {                                            // MicrofacetBXDF.h:292-434
    TRC_DEV static float ax() { return fmaxf(0.001f, Alpha::x()); }
    TRC_DEV static float ay() { return fmaxf(0.001f, Alpha::y()); }

    TRC_DEV static float D(F3 wh) {
        float tan2Theta = tan2_theta(wh);
        if (is_inf(tan2Theta)) return 0.;
        const float cos4Theta = cos2_theta(wh) * cos2_theta(wh);
        if (cos4Theta < 1e-16f) return 0;
        const Phi ph = phi_of(wh);
        float qc, qs;                                     // cos2_phi / ax^2, sin2_phi / ay^2
        div_const2(ph.c * ph.c, Alpha::by_x2(), ph.s * ph.s, Alpha::by_y2(), qc, qs);
        float e = (qc + qs) * tan2Theta;
        return rcp1(kPi * ax() * ay() * sqr(1 + e) * cos4Theta);
    }
    TRC_DEV static float lambda(F3 w) {
        float tan2Theta = tan2_theta(w);
        if (is_inf(tan2Theta)) return 0.;
        const Phi ph = phi_of(w);
        float alpha2 = sqr(ph.c * ax()) + sqr(ph.s * ay());
        return 0.5f * (sqrt_cr(1 + alpha2 * tan2Theta) - 1);
    }
    TRC_DEV static float G1(F3 w) { return rcp1(1 + lambda(w)); }
    TRC_DEV static float G(F3 wo, F3 wi) { return rcp1(1 + lambda(wo) + lambda(wi)); }
    TRC_DEV static float pdf(F3 wo, F3 wh) { return D(wh) * G1(wo) * fabsf(dot(wo, wh) / cos_theta(wo)); }

    TRC_DEV static void sample11(float cosTheta, float sinTheta, float U1, float U2, float& slope_x, float& slope_y) {
        if (cosTheta > .9999f) {
            float r = sqrt_cr(U1 / (1 - U1));
            float phi = 6.28318530718f * U2;
            float s, c;
            dm_sincosf(phi, &s, &c);
            slope_x = r * c;
            slope_y = r * s;
            return;
        }
        float tanTheta = sinTheta / cosTheta;             // sinTheta = sqrt(max(0, 1 - cosTheta^2)) = sin_theta of the caller's vector
        float a = rcp1(tanTheta);
        float G1 = 2 * rcp1(1 + sqrt_cr(1.f + rcp1(a * a)));     // 2 / x == 2 * (1 / x) bit for bit: x is 1 + a root, so 1 / x is never denormal and the doubling is exact
        float A = 2 * U1 / G1 - 1;
        float tmp = rcp1(A * A - 1.f);
        if (tmp > 1e10f) tmp = 1e10f;
        float B = tanTheta;
        float D = sqrt_cr(fmaxf(B * B * tmp * tmp - (A * A - B * B) * tmp, 0.0f));
        float slope_x_1 = B * tmp - D;
        float slope_x_2 = B * tmp + D;
        slope_x = (A < 0 || slope_x_2 > a) ? slope_x_1 : slope_x_2;              // a = 1 / tanTheta
        float S;
        if (U2 > 0.5f) { S = 1.f; U2 = 2.f * (U2 - .5f); }
        else { S = -1.f; U2 = 2.f * (.5f - U2); }
        float z = (U2 * (U2 * (U2 * 0.27385f - 0.73369f) + 0.46341f)) /
                  (U2 * (U2 * (U2 * 0.093073f + 0.309420f) - 1.000000f) + 0.597999f);
        slope_y = S * z * sqrt_cr(1.f + slope_x * slope_x);
    }
    TRC_DEV static F3 sample_wh(F3 wo, F2 u) {
        const bool flip = wo.z < 0;
        const F3 wi = flip ? -wo : wo;
        F3 wiStretched = normalize(f3(ax() * wi.x, ay() * wi.y, wi.z));
        float slope_x, slope_y;
        const float st = sin_theta(wiStretched);
        sample11(cos_theta(wiStretched), st, u.x, u.y, slope_x, slope_y);
        const Phi ph = phi_of(wiStretched, st);
        float tmp = ph.c * slope_x - ph.s * slope_y;
        slope_y = ph.s * slope_x + ph.c * slope_y;
        slope_x = tmp;
        slope_x = ax() * slope_x;
        slope_y = ay() * slope_y;
        F3 wh = normalize(f3(-slope_x, -slope_y, 1.f));
        return flip ? -wh : wh;
    }
};

// ---------------------------------------------------------------- lobes
struct Lambert {                                                     // MatteBXDF.hh:6-22 (F folds in the cosine)
    TRC_DEV static float F(F3, F3 wi) { return wi.z / kPi; }
    TRC_DEV static float pdf(F3 wo, F3 wi) { return wo.z * wi.z > 0 ? fabsf(wi.z) / kPi : 0; }
    // value and pdf are the same quotient: wi.z comes out of a square root (never negative), so |wi.z| / pi == wi.z / pi
    TRC_DEV static float S_F(F3 wo, F3& wi, F2 uu, float& pdf_out) {
        wi = cosine_sample_hemisphere(uu);
        const float v = div_pi(wi.z);
        pdf_out = wo.z * wi.z > 0 ? fabsf(v) : 0;
        return v;
    }
};

template <class Dist, class Fr>
struct MicroRefl {                                                   // MicrofacetBXDF.h:7-62
    TRC_DEV static F3 F(float R, F3 wo, F3 wi) {
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return f3(0);
        F3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return f3(0);
        wh = normalize(wh);
        // Faceforward(wh, (0,0,1)): dot(wh,(0,0,1)) is evaluated like the oracle does
        const float d001 = wh.x * 0.0f + wh.y * 0.0f + wh.z * 1.0f;
        const F3 whf = (d001 < 0.f) ? -wh : wh;
        F3 Fres = Fr::eval(dot(wi, whf));
        return f3(R) * Dist::D(wh) * Dist::G(wo, wi) * Fres / (4 * cosThetaI * cosThetaO);
    }
    TRC_DEV static float pdf(F3 wo, F3 wi) {
        if (wo.z * wi.z <= 0) return 0;
        F3 wh = normalize(wo + wi);
        return Dist::pdf(wo, wh) / (4 * dot(wo, wh));
    }
    // pdf(wo, wh) and F(wo, wi) both need Lambda(wo): written out so that it is computed ONCE (the guarded square roots and
    // reciprocals inside carry wave-level branches, which the compiler will not merge across the two calls).  Every
    // expression is the one pdf() and F() above evaluate, on the same operands in the same order.
    TRC_DEV static F3 S_F(float R, F3 wo, F3& wi, F2 uu, float& pdf_out) {
        if (wo.z == 0) return f3(0);
        F3 wh = Dist::sample_wh(wo, uu);
        if (dot(wo, wh) <= 0) return f3(0);
        wi = reflect(wo, wh);
        if (wo.z * wi.z <= 0) return f3(0);
        const float lam_o = Dist::lambda(wo);
        const float dist_pdf = Dist::D(wh) * rcp1(1 + lam_o) * fabsf(dot(wo, wh) / cos_theta(wo));      // Dist::pdf(wo, wh)
        pdf_out = dist_pdf / (4 * dot(wo, wh));
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);                                // F(R, wo, wi)
        if (cosThetaI == 0 || cosThetaO == 0) return f3(0);
        F3 wh2 = wi + wo;
        if (wh2.x == 0 && wh2.y == 0 && wh2.z == 0) return f3(0);
        wh2 = normalize(wh2);
        const float d001 = wh2.x * 0.0f + wh2.y * 0.0f + wh2.z * 1.0f;
        const F3 whf = (d001 < 0.f) ? -wh2 : wh2;
        F3 Fres = Fr::eval(dot(wi, whf));
        const float G = rcp1(1 + lam_o + Dist::lambda(wi));                                               // Dist::G(wo, wi)
        return f3(R) * Dist::D(wh2) * G * Fres / (4 * cosThetaI * cosThetaO);
    }
};

// MicrofacetReflection over a run-time Beckmann distribution with FresnelDielectric(1.5): Plastic's specular
// lobe (R = 1) and Glass's reflection lobe (R = kr = 0.98).  Same expressions as MicroRefl above.
struct BeckRefl {
    TRC_DEV static F3 F(const Beckmann& d, float R, F3 wo, F3 wi) {
        float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return f3(0);
        F3 wh = wi + wo;
        if (wh.x == 0 && wh.y == 0 && wh.z == 0) return f3(0);
        wh = normalize(wh);
        const float d001 = wh.x * 0.0f + wh.y * 0.0f + wh.z * 1.0f;
        const F3 whf = (d001 < 0.f) ? -wh : wh;
        F3 Fres = FrDiel15::eval(dot(wi, whf));
        return f3(R) * d.D(wh) * d.G(wo, wi) * Fres / (4 * cosThetaI * cosThetaO);
    }
    TRC_DEV static float pdf(const Beckmann& d, F3 wo, F3 wi) {
        if (wo.z * wi.z <= 0) return 0;
        F3 wh = normalize(wo + wi);
        return d.pdf(wo, wh) / (4 * dot(wo, wh));
    }
    // S_F after wh = d.sample_wh(wo, u) (MicrofacetBXDF.h:36-48); pdf_out untouched on the zero returns
    TRC_DEV static F3 finish(const Beckmann& d, float R, F3 wo, F3 wh, F3& wi, float& pdf_out) {
        if (dot(wo, wh) <= 0) return f3(0);
        wi = reflect(wo, wh);
        if (wo.z * wi.z <= 0) return f3(0);
        pdf_out = d.pdf(wo, wh) / (4 * dot(wo, wh));
        return F(d, R, wo, wi);
    }
};

// MicrofacetBXDF.h:64-135 as instantiated by GlassMaterial (:541): T = 0.98, etaA = 1, etaB = 1.5,
// mode = Importance (factor 1), and its own Fresnel is FresnelDielectric(etaA = 1.0) (:81).
struct BeckTransGlass {
    TRC_DEV static F3 F(const Beckmann& d, F3 wo, F3 wi) {
        const float etaA = 1.0f, etaB = 1.5f;
        if (wo.z * wi.z > 0) return f3(0);
        float cosThetaO = cos_theta(wo), cosThetaI = cos_theta(wi);
        if (cosThetaI == 0 || cosThetaO == 0) return f3(0);
        float eta = cos_theta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        F3 wh = normalize(wo + wi * eta);
        if (wh.z < 0) wh = -wh;
        if (dot(wo, wh) * dot(wi, wh) > 0) return f3(0);
        F3 Fres = f3(fr_dielectric(dot(wo, wh), etaA));
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float factor = 1;
        return (f3(1.0f) - Fres) * f3(0.98f) *
               fabsf(d.D(wh) * d.G(wo, wi) * eta * eta * fabsf(dot(wi, wh)) * fabsf(dot(wo, wh)) * factor * factor /
                     (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
    }
    TRC_DEV static float pdf(const Beckmann& d, F3 wo, F3 wi) {
        const float etaA = 1.0f, etaB = 1.5f;
        if (wo.z * wi.z > 0) return 0;
        float eta = cos_theta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
        F3 wh = normalize(wo + wi * eta);
        if (dot(wo, wh) * dot(wi, wh) > 0) return 0;
        float sqrtDenom = dot(wo, wh) + eta * dot(wi, wh);
        float dwh_dwi = fabsf((eta * eta * dot(wi, wh)) / (sqrtDenom * sqrtDenom));
        return d.pdf(wo, wh) * dwh_dwi;
    }
    // S_F after wh = d.sample_wh(wo, u) (MicrofacetBXDF.h:117-133)
    TRC_DEV static F3 finish(const Beckmann& d, F3 wo, F3 wh, F3& wi, float& pdf_out) {
        const float etaA = 1.0f, etaB = 1.5f;
        if (dot(wo, wh) < 0) return f3(0);
        float eta = cos_theta(wo) > 0 ? (etaA / etaB) : (etaB / etaA);
        if (!refract(wo, wh, eta, wi)) return f3(0);
        pdf_out = pdf(d, wo, wi);
        return F(d, wo, wi);
    }
};

// Tail of S_F for the three Beckmann lobes after wh = d.sample_wh(wo, u): MicrofacetReflection::S_F
// (MicrofacetBXDF.h:36-48 -> PDF :50-57, F :15-34) and MicrofacetTransmission::S_F (:117-133 -> PDF :100-115,
// F :83-98).  Written in three phases so that reflection and transmission lanes of a wavefront share the
// expensive part: (A) per-lobe geometry and the lobe's early-outs, (B) D, Lambda and the dielectric Fresnel
// term -- one instruction stream for all lanes, (C) per-lobe combination.  Every expression is the
// reference's, evaluated on the same operands in the same order; D(-w) == D(w) bit for bit (w enters only
// through squares), which is why transmission needs one D.  pdf_out is left untouched on the early-outs.
TRC_DEV F3 beckmann_finish(const Beckmann& d, bool refl, float R, F3 wo, F3 wh, F3& wi_out, float& pdf_out) {
    const float etaA = 1.0f, etaB = 1.5f;
    bool live;                    // got past the early-outs of S_F: pdf_out will be written
    bool pdf_live = false;        // ... with a non-trivial value
    bool f_live = false;          // F reaches its D * G * Fresnel expression
    F3 wi = wi_out, whA = wh, whB = wh;
    float eta = 0, fres_cos = 0, fres_eta = 1.5f;
    // ---- (A)
    if (refl) {
        live = !(dot(wo, wh) <= 0);
        if (live) { wi = reflect(wo, wh); live = !(wo.z * wi.z <= 0); }
        if (live) {
            pdf_live = true;
            const float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
            F3 sum = wi + wo;
            f_live = !(cosThetaI == 0 || cosThetaO == 0) && !(sum.x == 0 && sum.y == 0 && sum.z == 0);
            if (f_live) {
                whB = normalize(sum);
                const float d001 = whB.x * 0.0f + whB.y * 0.0f + whB.z * 1.0f;      // Faceforward(wh, (0,0,1))
                const F3 whf = (d001 < 0.f) ? -whB : whB;
                fres_cos = dot(wi, whf);
                fres_eta = 1.5f;
            }
        }
    } else {
        live = !(dot(wo, wh) < 0);
        if (live) live = refract(wo, wh, cos_theta(wo) > 0 ? (etaA / etaB) : (etaB / etaA), wi);
        if (live) {
            eta = cos_theta(wo) > 0 ? (etaB / etaA) : (etaA / etaB);
            const bool same_side = wo.z * wi.z > 0;
            whA = normalize(wo + wi * eta);
            const bool backfacing = dot(wo, whA) * dot(wi, whA) > 0;               // same truth value for -whA
            pdf_live = !same_side && !backfacing;
            whB = whA;
            if (whB.z < 0) whB = -whB;
            f_live = pdf_live && !(cos_theta(wi) == 0 || cos_theta(wo) == 0);
            fres_cos = dot(wo, whB);
            fres_eta = etaA;
        }
    }
    // ---- (B)
    float D_A = 0, D_B = 0, lam_o = 0, lam_i = 0, fres = 0;
    if (pdf_live) {
        lam_o = d.lambda(wo);
        D_A = d.D(whA);
        D_B = D_A;
        if (refl && f_live) D_B = d.D(whB);
        if (f_live) {
            lam_i = d.lambda(wi);
            fres = fr_dielectric(fres_cos, fres_eta);
        }
    }
    // ---- (C)
    F3 out = f3(0);
    if (live) {
        wi_out = wi;
        const float G1 = rcp1(1 + lam_o);
        const float G = rcp1(1 + lam_o + lam_i);
        if (refl) {
            const float dist_pdf = D_A * G1 * fabsf(dot(wo, wh)) / abs_cos_theta(wo);
            pdf_out = dist_pdf / (4 * dot(wo, wh));
            if (f_live) {
                const float cosThetaO = abs_cos_theta(wo), cosThetaI = abs_cos_theta(wi);
                out = f3(R) * D_B * G * f3(fres) / (4 * cosThetaI * cosThetaO);
            }
        } else {
            float p = 0;
            if (pdf_live) {
                const float sqrtDenom = dot(wo, whA) + eta * dot(wi, whA);
                const float dwh_dwi = fabsf((eta * eta * dot(wi, whA)) / (sqrtDenom * sqrtDenom));
                const float dist_pdf = D_A * G1 * fabsf(dot(wo, whA)) / abs_cos_theta(wo);
                p = dist_pdf * dwh_dwi;
            }
            pdf_out = p;
            if (f_live) {
                const float cosThetaO = cos_theta(wo), cosThetaI = cos_theta(wi);
                const float sqrtDenom = dot(wo, whB) + eta * dot(wi, whB);
                const float factor = 1;
                out = (f3(1.0f) - f3(fres)) * f3(0.98f) *
                      fabsf(D_B * G * eta * eta * fabsf(dot(wi, whB)) * fabsf(dot(wo, whB)) * factor * factor /
                            (cosThetaI * cosThetaO * sqrtDenom * sqrtDenom));
            }
        }
    }
    return out;
}

// composites (MicrofacetBXDF.h:436-586)
typedef MicroRefl<TrowbridgeReitzD<Alpha_01_02>, FrCond> MetalLobe;       // alpha (0.01, 0.02), R = 1
// Plastic specular: Beckmann (0.01, 0.1), R = 1; Glass: Beckmann (0.01, 0.01), reflection R = kr = 0.98

// material types (Material.hh:18-20) and texture types (Texture.hh:6) by ordinal
constexpr int kMatDiffuse = 0, kMatLambert = 1, kMatPlastic = 3, kMatMetal = 4, kMatGlass = 5, kMatNil = 10;
constexpr int kTexConstant = 0, kTexChecker = 1;

TRC_DEV F3 texture_value(int tex_type, F3 albedo, F2 uv) {           // Texture.hh:17-43
    if (tex_type == kTexChecker) {
        float s, c, s2, c2;
        dm_sincosf(8 * kPi * uv.x, &s, &c);
        dm_sincosf(kPi / 2 + 4 * kPi * uv.y, &s2, &c2);
        float sines = s * c2;
        return albedo * (0.5f * (sines < 0 ? 0.0f : 1.0f) + 0.5f);
    }
    // Constant; Image with a null texture and Noise resolve to albedo (see oracle/oracle.cpp texture_value)
    return (tex_type >= 0 && tex_type <= 3) ? albedo : f3(1.0f);
}

// Material::S_F, Material.hh:124-146.  bxPDF must be pre-set to 0 by the caller (B-3).
// Organised by LOBE rather than by material so that lanes of different materials share instructions: the
// cosine lobe serves Lambert and Plastic's diffuse half, the Beckmann sampling serves Plastic's specular half
// and both Glass lobes.  Every lane evaluates exactly the reference's expressions for its own material.
template <bool STATS>
TRC_DEV F3 material_S_F(int type, F3 color, F3 wo, F3& wi, F2 uu, float& pdf, TravCounters& cnt) {
    const bool plastic = type == kMatPlastic, glass = type == kMatGlass;
    F3 out = f3(0);
    if (type == kMatMetal) {
        ProfScope<STATS> scope(cnt, kProfMetal);
        out = color * MetalLobe::S_F(1.0f, wo, wi, uu, pdf);
    } else if (type == kMatLambert || (plastic && uu.x < 0.5f)) {
        ProfScope<STATS> scope(cnt, kProfLambert);
        F2 u2 = uu;
        if (plastic) u2.x *= 2;                                      // PlasticMaterial::S_F, :497-511
        const float v = Lambert::S_F(wo, wi, u2, pdf);
        out = plastic ? color * (f3(0.35f, 0.12f, 0.48f) * v) : color * f3(v);
    } else if (plastic || glass) {
        const float ratio = 0.25f;                                   // GlassMaterial::S_F, :564-573
        const bool refl = plastic || uu.x < ratio;
        F2 u2 = uu;
        if (plastic) { u2.x -= 0.5f; u2.x *= 2.0f; }
        else if (refl) u2.x = uu.x / ratio;
        else u2.x = (uu.x - ratio) / (1.0f - ratio);
        const Beckmann d = Beckmann::make_lobe(plastic);
        F3 lobe = f3(0);
        if (wo.z != 0) {
            F3 wh;
            { ProfScope<STATS> scope(cnt, kProfBeckSample); wh = d.sample_wh(wo, u2); }
            ProfScope<STATS> scope(cnt, kProfBeckEval);
            lobe = beckmann_finish(d, refl, plastic ? 1.0f : 0.98f, wo, wh, wi, pdf);
        }
        out = plastic ? color * (f3(0.2f) * lobe) : color * lobe;
    }
    return out;
}
TRC_DEV F3 material_S_F(int type, F3 color, F3 wo, F3& wi, F2 uu, float& pdf) {
    TravCounters none;                                                    // STATS = false: nobody touches it, the compiler drops it
    return material_S_F<false>(type, color, wo, wi, uu, pdf, none);
}

// Material::F, Material.hh:77-99 (pdf = bx.PDF, value = color * bx.F)
TRC_DEV F3 material_F(int type, F3 color, F3 wo, F3 wi, F2 uu, float& pdf) {
    const bool plastic = type == kMatPlastic, glass = type == kMatGlass;
    if (type == kMatMetal) {
        pdf = MetalLobe::pdf(wo, wi);
        return color * MetalLobe::F(1.0f, wo, wi);
    }
    if (type == kMatLambert || (plastic && uu.x < 0.5f)) {           // PlasticMaterial::F/PDF, :469-495
        pdf = Lambert::pdf(wo, wi);
        const float v = Lambert::F(wo, wi);
        return plastic ? color * (f3(0.35f, 0.12f, 0.48f) * v) : color * f3(v);
    }
    if (plastic || glass) {                                          // GlassMaterial::F/PDF, :543-562
        const float ratio = 0.25f;
        const Beckmann d = Beckmann::make_lobe(plastic);
        if (plastic || uu.x < ratio) {
            const float p = BeckRefl::pdf(d, wo, wi);
            pdf = plastic ? p : ratio * p;
            const F3 f = BeckRefl::F(d, plastic ? 1.0f : 0.98f, wo, wi);
            return plastic ? color * (f3(0.2f) * f) : color * f;
        }
        pdf = (1 - ratio) * BeckTransGlass::pdf(d, wo, wi);
        return color * BeckTransGlass::F(d, wo, wi);
    }
    pdf = 0;
    return f3(0);
}

}  // namespace trcdev
