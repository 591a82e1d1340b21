// uhdr_device_math.h -- gfx950 device functions for the gain-map pixel math.
//
// Two flavours of every transfer function:
//   *_exact : replays the reference's float/double promotion pattern:
//             float sub-expressions stay float, the libm call runs in double (ocml f64), the result
//             is rounded to float once.  Used by generate (always) and by apply in EXACT mode.
//   *_fast  : float only, v_log_f32 / v_exp_f32 / v_rcp_f32 / v_sqrt_f32 special-function ops.
//             Used by apply in FAST mode (tolerance: 1 LSB of 10 bit / 1 half-ULP).
//
// This translation unit is compiled with -ffp-contract=off: the reference's x86-64 build has no
// FMA, and a*b+c must round twice to stay bit-exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uhdr {

struct F3 {
  float x, y, z;
};

// gainmapmath.cpp:115-118
__device__ __forceinline__ float clamp01(float v) { return (v < 0.0f) ? 0.0f : (v > 1.0f) ? 1.0f : v; }

// ---- sRGB EOTF (gainmapmath.cpp:149-155) -------------------------------------------------------
__device__ __forceinline__ float srgb_inv_oetf_exact(float e) {
  if (e <= 0.04045f) return e / 12.92f;
  return (float)pow((double)((e + 0.055f) / 1.055f), 2.4);
}
// x^2.4 = x^2 * 2^(0.4*log2 x): keeps the exponent fed to v_exp_f32 in [-1.4, 0]
__device__ __forceinline__ float srgb_inv_oetf_fast(float e) {
  float lin = e * (1.0f / 12.92f);
  float x = (e + 0.055f) * (1.0f / 1.055f);
  float p = (x * x) * __builtin_amdgcn_exp2f(0.4f * __builtin_amdgcn_logf(x));
  return (e <= 0.04045f) ? lin : p;
}

// ---- HLG (gainmapmath.cpp:257-286) -------------------------------------------------------------
#define UHDR_HLG_A 0.17883277f
#define UHDR_HLG_B 0.28466892f
#define UHDR_HLG_C ((float)0.55991073)

__device__ __forceinline__ float hlg_oetf_exact(float e) {
  if (e <= 1.0f / 12.0f) return (float)sqrt((double)(3.0f * e));
  return (float)((double)UHDR_HLG_A * log((double)(12.0f * e - UHDR_HLG_B)) + (double)UHDR_HLG_C);
}
__device__ __forceinline__ float hlg_oetf_fast(float e) {
  float lo = __builtin_amdgcn_sqrtf(3.0f * e);
  // a*ln(x) = (a*ln2)*log2(x)
  float hi = (UHDR_HLG_A * 0.693147180559945f) * __builtin_amdgcn_logf(12.0f * e - UHDR_HLG_B) + UHDR_HLG_C;
  return (e <= 1.0f / 12.0f) ? lo : hi;
}
__device__ __forceinline__ float hlg_inv_oetf_exact(float e) {
  // pow(e, 2.0) of a float-valued double is exactly e*e in double
  if (e <= 0.5f) return (float)(((double)e * (double)e) / (double)3.0f);
  return (float)((exp((double)((e - UHDR_HLG_C) / UHDR_HLG_A)) + (double)UHDR_HLG_B) / (double)12.0f);
}

// e <= 0.5: e^2/3;  else (exp((e-c)/a) + b)/12 with exp(x) = 2^(x log2 e)
__device__ __forceinline__ float hlg_inv_oetf_fast(float e) {
  const float lo = (e * e) * (1.0f / 3.0f);
  const float hi = (__builtin_amdgcn_exp2f((e - UHDR_HLG_C) * (1.4426950408889634f / UHDR_HLG_A)) + UHDR_HLG_B) * (1.0f / 12.0f);
  return (e <= 0.5f) ? lo : hi;
}

// ---- PQ (gainmapmath.cpp:305-338) --------------------------------------------------------------
#define UHDR_PQ_M1 (2610.0f / 16384.0f)
#define UHDR_PQ_M2 (2523.0f / 4096.0f * 128.0f)
#define UHDR_PQ_C1 (3424.0f / 4096.0f)
#define UHDR_PQ_C2 (2413.0f / 4096.0f * 32.0f)
#define UHDR_PQ_C3 (2392.0f / 4096.0f * 32.0f)

__device__ __forceinline__ float pq_oetf_exact(float e) {
  if (e <= 0.0f) return 0.0f;
  double p = pow((double)e, (double)UHDR_PQ_M1);
  return (float)pow(((double)UHDR_PQ_C1 + (double)UHDR_PQ_C2 * p) / (1 + (double)UHDR_PQ_C3 * p),
                    (double)UHDR_PQ_M2);
}
__device__ __forceinline__ float pq_oetf_fast(float e) {
  float p = __builtin_amdgcn_exp2f(UHDR_PQ_M1 * __builtin_amdgcn_logf(e));
  float q = (UHDR_PQ_C1 + UHDR_PQ_C2 * p) * __builtin_amdgcn_rcpf(1.0f + UHDR_PQ_C3 * p);
  float r = __builtin_amdgcn_exp2f(UHDR_PQ_M2 * __builtin_amdgcn_logf(q));
  return (e <= 0.0f) ? 0.0f : r;
}
// pqOetf on the f32 units to ~3e-4 of a 10-bit code, for EXACT apply's pre-filter (tests/test_gpu_exact_filter.py measures it for
// every float in [0, 64]).  The naive form raises q = (c1 + c2 p) / (1 + c3 p), a number within 0.164 of 1, to the power 78.8: every
// rounding of q costs 5e-6 of the result.  Since c1 = 1 + c3 - c2 (the PQ constants are built that way, and their float values keep
// the identity: 3424/4096 = 1 + (2392 - 2413)/128), 1 - q = w = (1 - c1) (1 - p) / (1 + c3 p) exactly, and
//   ln q = ln(1 - w) = -2 atanh(w / (2 - w)) = -2 (s + s^3/3 + s^5/5 + s^7/7),  s <= 0.09,
// is formed from w directly: relative error ~2e-7 of a logarithm that is at most 7 where the code is at least 1.
__device__ __forceinline__ float pq_oetf_est(float e) {
  const float p = __builtin_amdgcn_exp2f(UHDR_PQ_M1 * __builtin_amdgcn_logf(e));
  const float w = ((1.0f - UHDR_PQ_C1) * (1.0f - p)) * __builtin_amdgcn_rcpf(__builtin_fmaf(UHDR_PQ_C3, p, 1.0f));
  const float s = w * __builtin_amdgcn_rcpf(2.0f - w);
  const float s2 = s * s;
  float h = __builtin_fmaf(s2, 1.0f / 7.0f, 1.0f / 5.0f);
  h = __builtin_fmaf(s2, h, 1.0f / 3.0f);
  h = __builtin_fmaf(s2 * s, h, s);                                   // atanh(s)
  const float r = __builtin_amdgcn_exp2f(h * (-2.0f * 1.4426950408889634f * UHDR_PQ_M2));
  return (e <= 0.0f) ? 0.0f : r;
}
__device__ __forceinline__ float pq_inv_oetf_exact(float e) {
  if (e <= 0.0001f) return 0.0f;
  double p = pow((double)e, (double)0.0126833f);
  return (float)pow(((double)128.0f * p - (double)107.0f) / ((double)2413.0f - (double)2392.0f * p),
                    (double)6.2773946361f);
}

// The same on the f32 units, for generate's pre-filter: relative error <= 3e-6 for every float in (1e-4, 1]
// (tests/test_gpu_filter.py measures it).  The naive form loses 4-5 digits: p = e^0.0127 is within 0.11 of 1, so
// 2413 - 2392 p cancels, and the outer power multiplies every relative error by 6.28.  Here
//   u = p - 1 = expm1(a ln e) directly (polynomial, |a ln e| <= 0.117), with ln e = ln2 (k + log2 m) kept as two terms so that
//     neither v_log_f32's error nor a rounding happens at the magnitude of k (up to 13);
//   128 p - 107 = 21 + 128 u and 2413 - 2392 p = 21 - 2392 u: no cancellation above the e <= 1e-4 cut-off;
//   q^c with q = mq 2^kq: exp2(c log2 mq + frac(c) kq) scaled by 2^(6 kq), again so that nothing is rounded at magnitude 30.
__device__ __forceinline__ float pq_inv_oetf_fast(float e) {
  constexpr float kA = (float)(0.012683300301432610 * 0.6931471805599453);   // (double)0.0126833f * ln 2
  const float m = __builtin_amdgcn_frexp_mantf(e);
  const float k = (float)__builtin_amdgcn_frexp_expf(e);
  const float x = __builtin_fmaf(__builtin_amdgcn_logf(m), kA, k * kA);
  float h = __builtin_fmaf(x, 1.0f / 5040.0f, 1.0f / 720.0f);
  h = __builtin_fmaf(x, h, 1.0f / 120.0f);
  h = __builtin_fmaf(x, h, 1.0f / 24.0f);
  h = __builtin_fmaf(x, h, 1.0f / 6.0f);
  h = __builtin_fmaf(x, h, 0.5f);
  h = __builtin_fmaf(x, h, 1.0f);
  const float u = x * h;
  const float q = __builtin_fmaf(128.0f, u, 21.0f) * __builtin_amdgcn_rcpf(__builtin_fmaf(-2392.0f, u, 21.0f));
  const float mq = __builtin_amdgcn_frexp_mantf(q);
  const int kq = __builtin_amdgcn_frexp_expf(q);
  const float z = __builtin_fmaf(6.2773946361f - 6.0f, (float)kq, 6.2773946361f * __builtin_amdgcn_logf(mq));   // (c - 6 is exact in float)
  const float r = __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(z), 6 * kq);
  return (e <= 0.0001f) ? 0.0f : r;
}

// =================================================================================================
// "guarded" double-precision transfer functions for generate.
//
// The reference rounds glibc's double pow/exp to float.  ocml's f64 pow/exp cost ~150-200 VALU slots
// each and made k_generate compute-bound (round-1 v1 profile: 2790 VALU instructions per wave).
// Here each call first evaluates a lean f64 polynomial (fast_log2 / fast_exp2, relative error well
// below 2^-44), then applies Ziv's rounding test: if every double within 2^-38 (relative) of the
// result rounds to the same float, that float is returned; otherwise (probability ~2^-13 per call)
// the exact ocml path is taken in an out-of-line call.  Either way the returned float is the one
// the exact path returns; uhdr_hip_selftest() verifies that exhaustively over the input domains.
// =================================================================================================

// Horner steps with the coefficient in an SGPR pair.  v_fma_f64 cannot take a 64-bit literal; left to itself
// hipcc materialises every constant addend with two v_mov_b32 into the destination of a v_fmac_f64
// (3 VALU slots per step -- 18 % of k_generate's VALU instructions in the first version); a __constant__
// table costs a scalar-memory round trip per call site; and constants handed to the compiler in SGPRs get
// hoisted out of the span loop, where 44 live SGPRs push the kernel over the SGPR file and into
// v_writelane/v_readlane spills (90 VALU instructions).  So each step is ONE asm statement: two s_mov_b32
// into a reserved scratch pair (scalar ALU, off the VALU pipe) followed by the N independent v_fma_f64 of
// the lock-step group.
template <uint32_t HI, uint32_t LO, int N>
__device__ __forceinline__ void horner_step(double (&p)[N], const double (&x)[N]) {
  static_assert(N == 1 || N == 2 || N == 3 || N == 6, "lock-step width");
#define UHDR_SETC "s_mov_b32 s96, %[lo]\n\ts_mov_b32 s97, %[hi]\n\t"
  if constexpr (N == 1) {
    asm(UHDR_SETC "v_fma_f64 %0, %0, %1, s[96:97]" : "+v"(p[0]) : "v"(x[0]), [lo] "i"(LO), [hi] "i"(HI) : "s96", "s97");
  } else if constexpr (N == 2) {
    asm(UHDR_SETC "v_fma_f64 %0, %0, %2, s[96:97]\n\tv_fma_f64 %1, %1, %3, s[96:97]"
        : "+v"(p[0]), "+v"(p[1]) : "v"(x[0]), "v"(x[1]), [lo] "i"(LO), [hi] "i"(HI) : "s96", "s97");
  } else if constexpr (N == 3) {
    asm(UHDR_SETC "v_fma_f64 %0, %0, %3, s[96:97]\n\tv_fma_f64 %1, %1, %4, s[96:97]\n\tv_fma_f64 %2, %2, %5, s[96:97]"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]) : "v"(x[0]), "v"(x[1]), "v"(x[2]), [lo] "i"(LO), [hi] "i"(HI) : "s96", "s97");
  } else {
    asm(UHDR_SETC
        "v_fma_f64 %0, %0, %6, s[96:97]\n\tv_fma_f64 %1, %1, %7, s[96:97]\n\tv_fma_f64 %2, %2, %8, s[96:97]\n\t"
        "v_fma_f64 %3, %3, %9, s[96:97]\n\tv_fma_f64 %4, %4, %10, s[96:97]\n\tv_fma_f64 %5, %5, %11, s[96:97]"
        : "+v"(p[0]), "+v"(p[1]), "+v"(p[2]), "+v"(p[3]), "+v"(p[4]), "+v"(p[5])
        : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), [lo] "i"(LO), [hi] "i"(HI) : "s96", "s97");
  }
#undef UHDR_SETC
}

// log2(x[j]) for normal positive x; |relative error| < 2^-50 (atanh series to s^19, division by a
// Newton-refined reciprocal).  N independent evaluations advance in lock step so that consecutive
// instructions never depend on each other (the f64 FMA latency would otherwise be exposed: measured
// +60 % kernel time for serial Horner chains at 7 waves per SIMD).
template <int N>
__device__ __forceinline__ void fast_log2_n(const double (&x)[N], double (&out)[N]) {
  double s[N], s2[N], p[N];
  int e[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double m = __builtin_amdgcn_frexp_mant(x[j]);  // [0.5, 1)
    e[j] = __builtin_amdgcn_frexp_exp(x[j]);       // x = m * 2^e
    const bool lo = m < 0x1.6a09e667f3bcdp-1;      // sqrt(1/2)
    m = lo ? (m + m) : m;                          // [sqrt(1/2), sqrt(2))
    e[j] = lo ? e[j] - 1 : e[j];
    const double d = m + 1.0, n = m - 1.0;
    double r = (double)__builtin_amdgcn_rcpf((float)d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    s[j] = n * r;
    s2[j] = s[j] * s[j];
  }
  // 2/((2k+1) ln 2), k = 9..0
#pragma unroll
  for (int j = 0; j < N; ++j) p[j] = 0x1.3703c1f4d0ffep-3;
  horner_step<0x3fc5b9acu, 0x9b743f0du>(p, s2);
  horner_step<0x3fc89f3bu, 0x1694cffeu>(p, s2);
  horner_step<0x3fcc68f5u, 0x68d31760u>(p, s2);
  horner_step<0x3fd0c9a8u, 0x4994022du>(p, s2);
  horner_step<0x3fd484b1u, 0x3d7c02a9u>(p, s2);
  horner_step<0x3fda6176u, 0x2a7aded9u>(p, s2);
  horner_step<0x3fe2776cu, 0x50ef9bfeu>(p, s2);
  horner_step<0x3feec709u, 0xdc3a03fdu>(p, s2);
  horner_step<0x40071547u, 0x652b82feu>(p, s2);
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = __builtin_fma(s[j], p[j], (double)e[j]);
}
__device__ __forceinline__ double fast_log2(double x) {
  const double in[1] = {x};
  double out[1];
  fast_log2_n<1>(in, out);
  return out[0];
}

// 2^P[j] for |P| < 1000; |relative error| < 2^-50 (Taylor to f^12 on |f| <= 1/2); N in lock step
template <int N>
__device__ __forceinline__ void fast_exp2_n(const double (&P)[N], double (&out)[N]) {
  double k[N], f[N], p[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    k[j] = __builtin_rint(P[j]);
    f[j] = P[j] - k[j];
  }
  // (ln 2)^i / i!, i = 12..1
#pragma unroll
  for (int j = 0; j < N; ++j) p[j] = 0x1.c3bd650fc2986p-36;
  horner_step<0x3dfe8cacu, 0x7351bb25u>(p, f);
  horner_step<0x3e3e4cf5u, 0x158b8ecau>(p, f);
  horner_step<0x3e7b5253u, 0xd395e7c4u>(p, f);
  horner_step<0x3eb62c02u, 0x23a5c824u>(p, f);
  horner_step<0x3eeffcbfu, 0xc588b0c7u>(p, f);
  horner_step<0x3f243091u, 0x2f86c787u>(p, f);
  horner_step<0x3f55d87fu, 0xe78a6731u>(p, f);
  horner_step<0x3f83b2abu, 0x6fba4e77u>(p, f);
  horner_step<0x3fac6b08u, 0xd704a0c0u>(p, f);
  horner_step<0x3fcebfbdu, 0xff82c58fu>(p, f);
  horner_step<0x3fe62e42u, 0xfefa39efu>(p, f);
#pragma unroll
  for (int j = 0; j < N; ++j) out[j] = __builtin_ldexp(__builtin_fma(p[j], f[j], 1.0), (int)k[j]);
}
__device__ __forceinline__ double fast_exp2(double P) {
  const double in[1] = {P};
  double out[1];
  fast_exp2_n<1>(in, out);
  return out[0];
}

// Ziv test: may y (relative error < 2^-38) be rounded to float without knowing its last bits?
// NORMAL_RANGE: the caller guarantees y rounds to a normal float (otherwise the midpoint pattern differs and
// the test must fail, which the y >= 2^-126 comparison ensures).
template <bool NORMAL_RANGE = false>
__device__ __forceinline__ bool ziv_safe(double y) {
  const uint32_t lo = (uint32_t)__double2loint(y);
  const int32_t dist = (int32_t)(lo & 0x1FFFFFFFu) - 0x10000000;  // low 29 bits vs the float midpoint
  const uint32_t ad = (uint32_t)(dist < 0 ? -dist : dist);
  return ad > (1u << 16) && (NORMAL_RANGE || y >= 0x1p-126);  // 2^-38 relative = 2^15 double ulps; 2x margin
}

// x / a for a compile-time constant a, as q + fma(-q, a, x) * (1/a).  Not correctly rounded for
// every conceivable x; uhdr_hip_selftest() proves it equals IEEE division on the whole input
// domain each call site can see.
__device__ __forceinline__ float div_const(float x, float a, float ra) {
  const float q = x * ra;
  return __builtin_fmaf(__builtin_fmaf(-q, a, x), ra, q);
}

__device__ __attribute__((noinline)) float srgb_inv_oetf_slow(float e) { return srgb_inv_oetf_exact(e); }
__device__ __attribute__((noinline)) float hlg_inv_oetf_slow(float e) { return hlg_inv_oetf_exact(e); }
__device__ __attribute__((noinline)) float pq_inv_oetf_slow(float e) { return pq_inv_oetf_exact(e); }

// sRGB EOTF.  x^2.4 = x^2 * z with z = x^(2/5), i.e. z^5 = x^2: the special-function unit gives z0 = 2^(0.4 log2 x)
// to ~2^-22, and ONE Newton step in f64, z1 = z0 - (z0^5 - x^2) * R with R ~ 1/(5 z0^4) from v_rcp_f32, leaves a
// relative error ~2^-42 (2 e0^2 from the iteration + 2^-44 from R) -- 24 VALU slots instead of the 45 of
// fast_log2 + fast_exp2.  The Ziv test below (2^-38) decides whether the float rounding is safe.
template <int N>
__device__ __forceinline__ void srgb_inv_oetf_guarded_n(float (&e)[N]) {
  float lin[N], xf[N];
  double y[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    lin[j] = div_const(e[j], 12.92f, 1.0f / 12.92f);
    // the fma remainder of div_const underflows for |e| < ~7e-32 (selftest): such values (never produced by
    // 8-bit content, but representable) take the hardware IEEE division
    if (e[j] != 0.0f && e[j] < 0x1p-100f) lin[j] = e[j] / 12.92f;
    xf[j] = div_const(e[j] + 0.055f, 1.055f, 1.0f / 1.055f);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double x = (double)xf[j];
    const double z0 = (double)__builtin_amdgcn_exp2f(0.4f * __builtin_amdgcn_logf(xf[j]));
    const double X = x * x;                       // exact (48 significant bits)
    const double z2 = z0 * z0, z4 = z2 * z2;
    const double d = __builtin_fma(z4, z0, -X);   // z0^5 - x^2
    const double R = (double)(0.2f * __builtin_amdgcn_rcpf((float)z4));
    y[j] = X * __builtin_fma(-d, R, z0);
  }
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float p = (float)y[j];
    if (e[j] > 0.04045f && !ziv_safe<true>(y[j])) p = srgb_inv_oetf_slow(e[j]);   // p >= 0.0031 here
    e[j] = (e[j] <= 0.04045f) ? lin[j] : p;
  }
}
__device__ __forceinline__ float srgb_inv_oetf_guarded(float e) {
  float v[1] = {e};
  srgb_inv_oetf_guarded_n<1>(v);
  return v[0];
}

template <int N>
__device__ __forceinline__ void hlg_inv_oetf_guarded_n(float (&e)[N]) {
  float lo[N];
  double y[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    // e <= 0.5: (e*e)/3 in double, division by 3 as q + fma(-q,3,t)/3 (exact; selftest)
    const double t = (double)e[j] * (double)e[j];
    const double q0 = t * 0x1.5555555555555p-2;
    lo[j] = (float)__builtin_fma(__builtin_fma(-q0, 3.0, t), 0x1.5555555555555p-2, q0);
    // e > 0.5: (exp((e-c)/a) + b) / 12
    y[j] = (double)div_const(e[j] - UHDR_HLG_C, UHDR_HLG_A, 1.0f / UHDR_HLG_A) * 0x1.71547652b82fep+0;
  }
  fast_exp2_n<N>(y, y);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double yy = (y[j] + (double)UHDR_HLG_B) * (1.0 / 12.0);
    float hi = (float)yy;
    if (e[j] > 0.5f && !ziv_safe<true>(yy)) hi = hlg_inv_oetf_slow(e[j]);         // hi >= 1/12 here
    e[j] = (e[j] <= 0.5f) ? lo[j] : hi;
  }
}
__device__ __forceinline__ float hlg_inv_oetf_guarded(float e) {
  float v[1] = {e};
  hlg_inv_oetf_guarded_n<1>(v);
  return v[0];
}

template <int N>
__device__ __forceinline__ void pq_inv_oetf_guarded_n(float (&e)[N]) {
  double t[N];
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = (double)((e[j] <= 0.0001f) ? 1.0f : e[j]);
  fast_log2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] *= (double)0.0126833f;
  fast_exp2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double num = __builtin_fma(128.0, t[j], -107.0), den = __builtin_fma(-2392.0, t[j], 2413.0);
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    t[j] = num * r;
  }
  fast_log2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] *= (double)6.2773946361f;
  fast_exp2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float o = (float)t[j];
    if (e[j] > 0.0001f && !ziv_safe(t[j])) o = pq_inv_oetf_slow(e[j]);
    e[j] = (e[j] <= 0.0001f) ? 0.0f : o;
  }
}
__device__ __forceinline__ float pq_inv_oetf_guarded(float e) {
  float v[1] = {e};
  pq_inv_oetf_guarded_n<1>(v);
  return v[0];
}

// ---- the forward OETFs and applyGain's factor for apply's EXACT mode: the same lean f64 + rounding test -------------------
__device__ __attribute__((noinline)) float hlg_oetf_slow(float e) { return hlg_oetf_exact(e); }
__device__ __attribute__((noinline)) float pq_oetf_slow(float e) { return pq_oetf_exact(e); }
__device__ __attribute__((noinline)) float exp2_to_float_slow(float x) { return (float)exp2((double)x); }

// hlgOetf (gainmapmath.cpp:259-265).  e <= 1/12: (float)sqrt((double)(3.0f * e)) from the f32 special-function square root
// (2^-23) and one Newton step in double -- t - y0^2 is exact there -- leaving ~2^-45, then the rounding test; otherwise
// a * ln(12 e - b) + c in double from the lean log2.
template <int N>
__device__ __forceinline__ void hlg_oetf_guarded_n(float (&e)[N]) {
  double t[N];
  float lo[N];
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const float t3 = 3.0f * e[j];
    const float y0 = __builtin_amdgcn_sqrtf(t3);
    const double d = __builtin_fma(-(double)y0, (double)y0, (double)t3);
    const double y1 = __builtin_fma(d, (double)(0.5f * __builtin_amdgcn_rcpf(y0)), (double)y0);
    lo[j] = (float)y1;
    if (t3 == 0.0f) lo[j] = 0.0f;
    else if (e[j] <= 1.0f / 12.0f && (t3 < 0x1p-100f || !ziv_safe<true>(y1))) lo[j] = hlg_oetf_slow(e[j]);
    const float x = 12.0f * e[j] - UHDR_HLG_B;
    t[j] = (double)((e[j] <= 1.0f / 12.0f) ? 1.0f : x);   // (x >= 0.715 on the branch that uses it)
  }
  fast_log2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double yy = __builtin_fma((double)UHDR_HLG_A * 0x1.62e42fefa39efp-1, t[j], (double)UHDR_HLG_C);
    float hi = (float)yy;
    // a * ln2 is one more rounding than the reference's a * log(x) (2^-53): inside the test's margin
    if (e[j] > 1.0f / 12.0f && !ziv_safe<true>(yy)) hi = hlg_oetf_slow(e[j]);
    e[j] = (e[j] <= 1.0f / 12.0f) ? lo[j] : hi;
  }
}
// pqOetf (gainmapmath.cpp:309-312): two pow() as exp2(m log2 x)
template <int N>
__device__ __forceinline__ void pq_oetf_guarded_n(float (&e)[N]) {
  double t[N];
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] = (double)((e[j] <= 0.0f) ? 1.0f : e[j]);
  fast_log2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] *= (double)UHDR_PQ_M1;
  fast_exp2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double num = __builtin_fma((double)UHDR_PQ_C2, t[j], (double)UHDR_PQ_C1), den = __builtin_fma((double)UHDR_PQ_C3, t[j], 1.0);
    double r = (double)__builtin_amdgcn_rcpf((float)den);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-den, r, 1.0), r, r);
    t[j] = num * r;
  }
  fast_log2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) t[j] *= (double)UHDR_PQ_M2;
  fast_exp2_n<N>(t, t);
#pragma unroll
  for (int j = 0; j < N; ++j) {
    float o = (float)t[j];
    if (e[j] > 0.0f && !ziv_safe(t[j])) o = pq_oetf_slow(e[j]);
    e[j] = (e[j] <= 0.0f) ? 0.0f : o;
  }
}
// (float)exp2((double)x), applyGain's factor (gainmapmath.cpp:553)
__device__ __forceinline__ float exp2_to_float_guarded(float x) {
  if (!(__builtin_fabsf(x) < 120.0f)) return exp2_to_float_slow(x);   // NaN, overflow and the subnormal results
  const double y = fast_exp2((double)x);
  return ziv_safe<true>(y) ? (float)y : exp2_to_float_slow(x);
}

// ---- encodeGain (gainmapmath.cpp:529-541) ------------------------------------------------------
__device__ __forceinline__ float raw_gain(float y_sdr, float y_hdr) {
  float gain = 1.0f;
  if (y_sdr > 0.0f) gain = y_hdr / y_sdr;
  return gain;
}
__device__ __forceinline__ uint8_t encode_gain(float gain, float min_boost, float max_boost,
                                               float log2_min, float log2_max) {
  if (gain < min_boost) gain = min_boost;
  if (gain > max_boost) gain = max_boost;
  return (uint8_t)((log2((double)gain) - (double)log2_min) / (double)(log2_max - log2_min) *
                   (double)255.0f);
}

__device__ __attribute__((noinline)) uint8_t encode_gain_slow(float gain, float min_boost, float max_boost,
                                                              float log2_min, float log2_max) {
  return encode_gain(gain, min_boost, max_boost, log2_min, log2_max);
}
// same byte as encode_gain(): clamped gains use host-computed bytes, the rest a fast log2 plus a
// distance-to-integer test (the value is truncated, so only the integer part matters)
__device__ __forceinline__ uint8_t encode_gain_guarded(float gain, float min_boost, float max_boost, float log2_min,
                                                       float log2_max, double k_scale, uint32_t byte_min,
                                                       uint32_t byte_max) {
  if (!(gain > min_boost)) return (uint8_t)byte_min;
  if (!(gain < max_boost)) return (uint8_t)byte_max;
  const double v = (fast_log2((double)gain) - (double)log2_min) * k_scale;
  const double fl = __builtin_floor(v), fr = v - fl;
  if (fr < 0x1p-20 || fr > 1.0 - 0x1p-20 || !(v > 0.0) || !(v < 256.0))
    return encode_gain_slow(gain, min_boost, max_boost, log2_min, log2_max);
  return (uint8_t)(uint32_t)fl;
}

// ---- output packing (gainmapmath.cpp:722-732, gainmapmath.h:136-147) ---------------------------
__device__ __forceinline__ uint32_t pack_1010102(float r, float g, float b) {
  return (0x3ffu & (uint32_t)(r * 1023.0f)) | ((0x3ffu & (uint32_t)(g * 1023.0f)) << 10) |
         ((0x3ffu & (uint32_t)(b * 1023.0f)) << 20) | (0x3u << 30);
}
__device__ __forceinline__ uint32_t float_to_half(float f) {
  const uint32_t b = __float_as_uint(f) + 0x00001000u;
  const int32_t e = (int32_t)((b & 0x7F800000u) >> 23);
  const uint32_t m = b & 0x007FFFFFu;
  uint32_t r = (b & 0x80000000u) >> 16;
  r |= (e > 112) ? ((((uint32_t)(e - 112) << 10) & 0x7C00u) | (m >> 13)) : 0u;
  r |= ((e < 113) && (e > 101)) ? ((((0x007FF000u + m) >> (125 - e)) + 1u) >> 1) : 0u;
  r |= (e > 143) ? 0x7FFFu : 0u;
  return r & 0xFFFFu;
}
// returns {lo, hi} 32-bit halves of the reference's uint64 (R | G<<16 | B<<32 | 1.0h<<48)
__device__ __forceinline__ uint2 pack_f16(float r, float g, float b) {
  return make_uint2(float_to_half(r) | (float_to_half(g) << 16), float_to_half(b) | (0x3C00u << 16));
}

// FAST apply only: the hardware conversion (v_cvt_f16_f32, round to nearest even, half subnormals kept) instead of the ~20 integer
// slots per value of the routine above.  The reference rounds half UP (the + 0x1000), so the two differ on exact ties only (one
// value in 8192) and then by one half-ULP, which is FAST mode's stated tolerance; values here are <= 1, far from the 0x7FFF
// saturation branch.
__device__ __forceinline__ uint2 pack_f16_hw(float r, float g, float b) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  const h2 rg = {(_Float16)r, (_Float16)g}, ba = {(_Float16)b, (_Float16)1.0f};
  return make_uint2(__builtin_bit_cast(uint32_t, rg), __builtin_bit_cast(uint32_t, ba));
}

// ---- order-preserving float <-> uint key for atomicMin/atomicMax -------------------------------
__device__ __forceinline__ uint32_t float_to_key(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

}  // namespace uhdr
