// uhdr_wave_scan.h -- workgroup sums and prefix sums for gfx950, written for 64-wide waves: a butterfly / a Hillis-Steele scan by
// lane shuffles inside the wave (6 steps), one LDS word per wave across the workgroup (at most 4 waves here, added up serially).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace uhdr {

template <class T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
template <class T>
__device__ __forceinline__ T wave_inclusive_sum(T v) {
  const int lane = (int)(threadIdx.x & 63u);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const T u = __shfl_up(v, off, 64);
    if (lane >= off) v += u;
  }
  return v;
}
// sum over the workgroup's BLOCK threads; every thread gets it.  s_part: BLOCK / 64 words of LDS, free again on return.
template <int BLOCK, class T>
__device__ __forceinline__ T block_sum(T v, T* s_part) {
  static_assert(BLOCK % 64 == 0 && BLOCK <= 1024, "whole waves");
  v = wave_sum(v);
  if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = v;
  __syncthreads();
  T total = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w) total += s_part[w];
  __syncthreads();
  return total;
}
// exclusive prefix sum over the workgroup's threads (thread t gets the sum of the values of threads 0 .. t-1)
template <int BLOCK, class T>
__device__ __forceinline__ T block_exclusive_sum(T v, T* s_part) {
  static_assert(BLOCK % 64 == 0 && BLOCK <= 1024, "whole waves");
  const T incl = wave_inclusive_sum(v);
  if ((threadIdx.x & 63u) == 63u) s_part[threadIdx.x >> 6] = incl;
  __syncthreads();
  T before = 0;
#pragma unroll
  for (int w = 0; w < BLOCK / 64; ++w)
    if (w < (int)(threadIdx.x >> 6)) before += s_part[w];
  __syncthreads();
  return before + incl - v;
}

}  // namespace uhdr
