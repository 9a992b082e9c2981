/*
 * ppg_detmath.h — the numerical contract shared by the HIP kernels and the CPU oracle.
 *
 * The reference calls libm (sincos, atan2, exp, pow; guided_path.cpp:65, 100, 404, 592, 603, 681 and
 * warp.cpp:81-102).  libm results differ in the last ulp between glibc and the AMD device library, which
 * would make "same seed ⇒ same path" impossible across CPU and GPU.  Both sides therefore evaluate
 * these functions with the fixed sequences of IEEE-754 single-precision +,-,*,/ below (Cephes-style
 * range reduction + polynomial, max error ≈ 1–2 ulp on the ranges used), and both are compiled with
 * -ffp-contract=off, so every result is bit-identical on x86-64 and gfx950.
 *
 * tests/test_detmath.py checks these against numpy's libm within a stated ulp bound.
 */
#ifndef PPG_DETMATH_H
#define PPG_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define PPG_HD __host__ __device__ inline
#else
#define PPG_HD inline
#endif

#define PPG_PI_F 3.14159265358979323846f      /* M_PI of the reference, constants.h:63,80 (float) */
#define PPG_INV_PI_F 0.31830988618379067154f  /* INV_PI, constants.h */
#define PPG_EPSILON 1e-4f                     /* Epsilon, constants.h:28 */
#define PPG_SHADOW_EPSILON 1e-3f              /* ShadowEpsilon, constants.h:29 */

PPG_HD uint32_t ppg_f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
PPG_HD float ppg_u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

PPG_HD bool ppg_isfinite(float x) { return (ppg_f2u(x) & 0x7f800000u) != 0x7f800000u; }
PPG_HD float ppg_max(float a, float b) { return (a < b) ? b : a; } /* std::max semantics */
PPG_HD float ppg_min(float a, float b) { return (b < a) ? b : a; } /* std::min semantics */
PPG_HD float ppg_abs(float x) { return ppg_u2f(ppg_f2u(x) & 0x7fffffffu); }

/* 2^n for -126 <= n <= 127, exact. */
PPG_HD float ppg_exp2i(int n) { return ppg_u2f((uint32_t)(n + 127) << 23); }

/* sin and cos of x (|x| < ~8192), Cephes sinf/cosf scheme. */
PPG_HD void ppg_sincos(float xx, float *s_out, float *c_out) {
    float x = ppg_abs(xx);
    int sin_neg = (xx < 0.0f);
    int cos_neg = 0;
    int j = (int)(x * 1.27323954473516f); /* 4/pi */
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    if (j > 3) { sin_neg = !sin_neg; cos_neg = !cos_neg; j -= 4; }
    if (j > 1) cos_neg = !cos_neg;
    x = ((x - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = x * x;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * x + x;
    float s, c;
    if (j == 1 || j == 2) { s = pc; c = ps; } else { s = ps; c = pc; }
    *s_out = sin_neg ? -s : s;
    *c_out = cos_neg ? -c : c;
}

/* atan(x), Cephes atanf scheme. */
PPG_HD float ppg_atan(float xx) {
    float x = ppg_abs(xx);
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966192f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483096f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return (xx < 0.0f) ? -y : y;
}

/* atan2(y, x) with the usual quadrant rules (finite inputs). */
PPG_HD float ppg_atan2(float y, float x) {
    if (x == 0.0f) {
        if (y == 0.0f) return 0.0f;
        return (y > 0.0f) ? 1.5707963267948966192f : -1.5707963267948966192f;
    }
    float a = ppg_atan(y / x);
    if (x < 0.0f) a = (y < 0.0f) ? (a - PPG_PI_F) : (a + PPG_PI_F);
    return a;
}

/* acos / tan for the visible-normal sampling of microfacet.h:425-470, 645-690 — via the routines above so that CPU and
   GPU agree to the bit: acos(x) = atan2(sqrt((1 - x)(1 + x)), x), tan(x) = sin / cos */
PPG_HD float ppg_acos(float x) { return ppg_atan2(__builtin_sqrtf(ppg_max(0.0f, (1.0f - x) * (1.0f + x))), x); }
PPG_HD float ppg_tan(float x) { float s, c; ppg_sincos(x, &s, &c); return s / c; }

/* exp(x) for |x| <= 80, Cephes expf scheme. */
PPG_HD float ppg_exp(float x) {
    if (x > 80.0f) x = 80.0f;
    if (x < -80.0f) x = -80.0f;
    float t = 1.44269504088896341f * x + 0.5f;
    float fz = (float)(int)t;
    if (fz > t) fz -= 1.0f; /* floor */
    x -= fz * 0.693359375f;
    x -= fz * -2.12194440e-4f;
    int n = (int)fz;
    float z = x * x;
    z = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x
          + 1.6666665459e-1f) * x + 5.0000001201e-1f) * z + x + 1.0f;
    return z * ppg_exp2i(n);
}

/* b^n for integer n >= 0 by binary exponentiation (stands in for std::pow(beta, iter), GP:100). */
/* log(x) for normal x > 0 (log(0) = -inf, x < 0 → NaN), Cephes logf scheme; pow(x, y) = exp(y log x) for x > 0.
   Used by the Beckmann microfacet code (microfacet.h: fastlog / fastexp / std::pow, math.cpp:25-72 erf / erfinv). */
PPG_HD float ppg_log(float x) {
    if (!(x > 0.0f)) return x == 0.0f ? -__builtin_inff() : __builtin_nanf("");
    uint32_t u = ppg_f2u(x);
    int e = (int)(u >> 23) - 126;                            /* frexp: x = m 2^e, m in [0.5, 1) */
    float m = ppg_u2f((u & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else m = m - 1.0f;
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m
                  - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    const float fe = (float)e;
    if (e) y += -2.12194440e-4f * fe;
    y += -0.5f * z;
    z = m + y;
    if (e) z += 0.693359375f * fe;
    return z;
}
PPG_HD float ppg_pow(float x, float y) { return ppg_exp(y * ppg_log(x)); }

PPG_HD float ppg_powi(float b, int n) {
    float r = 1.0f;
    while (n > 0) {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}

/* The bias-corrected learning rate of AdamOptimizer::step (GP:100):
       Float actualLearningRate = learningRate * std::sqrt(1 - std::pow(beta2, iter)) / (1 - std::pow(beta1, iter));
   std::pow(float, int) is the promoting overload of C++11 — both arguments become double — so the reference evaluates the whole right-hand
   side in DOUBLE and rounds once, on the assignment to Float.  The same here: the powers by binary exponentiation in double (a few double
   ulps from libm's pow), the square root by two Newton steps from the correctly rounded float root (IEEE +, *, / only: the same bits on
   x86-64 and gfx950, whatever either does for a double sqrt), one rounding to float at the end — which, the double result being ~1e-15
   accurate, is the reference's float except at a rounding tie. */
PPG_HD double ppg_powi_d(double b, int n) {
    double r = 1.0;
    while (n > 0) {
        if (n & 1) r = r * b;
        b = b * b;
        n >>= 1;
    }
    return r;
}
PPG_HD float ppg_adam_learning_rate(float learningRate, float beta1, float beta2, int iter) {
    const double x = 1.0 - ppg_powi_d((double)beta2, iter);
    double y = (double)__builtin_sqrtf((float)x);
    if (y > 0.0) { y = 0.5 * (y + x / y); y = 0.5 * (y + x / y); }
    return (float)((double)learningRate * y / (1.0 - ppg_powi_d((double)beta1, iter)));
}

/* Fixed-point accumulation of SD-tree statistics: round-to-nearest-even of x * 2^24 as uint64.
   Supported range of ONE contribution: [2^-25, 2^26) — smaller values round to 0, larger ones (and +inf) are clamped to 2^26 = 2^50
   fixed-point units, NaN contributes nothing.  A bin can therefore absorb 2^13 maximal contributions (or 2^39 of unit size) before the
   64-bit sum could wrap; radiance x weight / pdf records of a render stay orders of magnitude below the clamp. */
PPG_HD uint64_t ppg_to_fixed(float x) {
    float v = x * 16777216.0f;
    if (v != v) return 0ull;
    if (!(v < 1125899906842624.0f)) return 1ull << 50;
    return (uint64_t)__builtin_rintf(v); /* exactly specified: nearest-even (v_rndne_f32 / rintf) */
}
PPG_HD float ppg_from_fixed(uint64_t a) { return (float)a * 5.9604644775390625e-8f; /* 2^-24 */ }

/* Signed variant for Adam gradient sums, scale 2^20, |x| clamped to 2^40. */
PPG_HD int64_t ppg_to_sfixed(float x) {
    float a = ppg_abs(x);
    if (!(a < 1.099511627776e12f)) a = 1.099511627776e12f;
    int64_t r = (int64_t)__builtin_rintf(a * 1048576.0f);
    return (x < 0.0f) ? -r : r;
}
PPG_HD float ppg_from_sfixed(int64_t a) { return (float)a * 9.5367431640625e-7f; /* 2^-20 */ }

/* Rounds by image region (include/ppg.h "Rounds by image region"): rank[by * bx] receives, for every 32x32 block (row-major), its position
   in the spiral in which the reference's scheduler hands blocks out — from the central block outwards: right, down, left, up with growing run
   lengths (BlockedImageProcess, imageproc.cpp:29-80).  Host only; shared by product and oracle like the sampler. */
inline void ppg_spiral_block_ranks(int bx, int by, int *rank) {
    const int total = bx * by;
    for (int k = 0; k < total; ++k) rank[k] = -1;
    int x = bx / 2, y = by / 2, next = 0, run = 1, dir = 0;
    const int dx[4] = {1, 0, -1, 0}, dy[4] = {0, 1, 0, -1};
    while (next < total && run <= 4 * (bx + by)) {
        for (int rep = 0; rep < 2 && next < total; ++rep) {
            for (int k = 0; k < run && next < total; ++k) {
                if (x >= 0 && x < bx && y >= 0 && y < by && rank[y * bx + x] < 0) rank[y * bx + x] = next++;
                x += dx[dir]; y += dy[dir];
            }
            dir = (dir + 1) & 3;
        }
        ++run;
    }
    for (int k = 0; k < total; ++k) if (rank[k] < 0) rank[k] = next++;
}

/* Scene-setup constant of the plastic BSDF: fresnelDiffuseReflectance(eta, fast = false) (util.cpp:797-861), the
   hemispherical average ∫0^1 F(sqrt(xi), eta) dxi of the unpolarised dielectric Fresnel reflectance
   (fresnelDielectricExt, util.cpp:651-681).  The reference integrates adaptively (Gauss-Lobatto, 1e-5); here composite
   Simpson with 4096 intervals in double — plain IEEE operations, so the oracle and the HIP library's host code obtain the
   same float.  Host only. */
inline double ppg_fresnel_dielectric_d(double cosThetaI, double eta) {
    if (eta == 1.0) return 0.0;
    double scale = (cosThetaI > 0) ? 1.0 / eta : eta;
    double cosThetaTSqr = 1.0 - (1.0 - cosThetaI * cosThetaI) * (scale * scale);
    if (cosThetaTSqr <= 0.0) return 1.0;
    double ci = cosThetaI < 0 ? -cosThetaI : cosThetaI, ct = __builtin_sqrt(cosThetaTSqr);
    double Rs = (ci - eta * ct) / (ci + eta * ct), Rp = (eta * ci - ct) / (eta * ci + ct);
    return 0.5 * (Rs * Rs + Rp * Rp);
}
inline float ppg_fresnel_diffuse_reflectance(float eta) {
    const int n = 4096;
    const double h = 1.0 / n;
    double acc = ppg_fresnel_dielectric_d(0.0, (double)eta) + ppg_fresnel_dielectric_d(1.0, (double)eta);
    for (int i = 1; i < n; ++i) acc += ((i & 1) ? 4.0 : 2.0) * ppg_fresnel_dielectric_d(__builtin_sqrt(i * h), (double)eta);
    return (float)(acc * h / 3.0);
}

#endif /* PPG_DETMATH_H */
