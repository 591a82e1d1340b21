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
__device__ __forceinline__ float pq_inv_oetf_exact(float e) {
  if (e <= 0.0001f) return 0.0f;
  double p = pow((double)e, (double)0.0126833f);
  return (float)pow(((double)128.0f * p - (double)107.0f) / ((double)2413.0f - (double)2392.0f * p),
                    (double)6.2773946361f);
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

// ---- order-preserving float <-> uint key for atomicMin/atomicMax -------------------------------
__device__ __forceinline__ uint32_t float_to_key(float f) {
  uint32_t b = __float_as_uint(f);
  return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ float key_to_float(uint32_t k) {
  return __uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
}

}  // namespace uhdr
