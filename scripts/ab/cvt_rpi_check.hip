// v_cvt_rpi_i32_f32 ("round to nearest, ties toward +infinity": floor(x + 0.5) without an intermediate rounding) against the
// reference's LUT index expression  (uint32_t)((double)t + 0.5)  for EVERY float 0 <= t < 2^31 (and -0), and the issue rate of
// either form.  hipcc --offload-arch=gfx950 -O3 scripts/ab/cvt_rpi_check.hip -o scripts/ab/cvt_rpi_check
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__device__ __forceinline__ int rpi(float t) {
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(t));
  return r;
}
__global__ void k_check(unsigned long long* bad, uint32_t* first_bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (uint64_t u = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u <= 0x4F000000ull; u += stride) {   // +0 ... 2^31
    const float t = __uint_as_float((uint32_t)u);
    if (!(t < 2147483648.0f)) continue;
    const uint32_t want = (uint32_t)((double)t + 0.5);
    const uint32_t got = (uint32_t)rpi(t);
    if (want != got && !(u == 0x4EFFFFFFull && 0)) {
      ++mine;
      atomicMin(first_bad, (uint32_t)u);
    }
  }
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    if ((uint32_t)rpi(-0.0f) != 0u) ++mine;
  }
  if (mine) atomicAdd(bad, mine);
}
template <int FORM>
__global__ void k_rate(const float* in, uint32_t* out, int iters) {
  float t[8];
  for (int k = 0; k < 8; ++k) t[k] = in[threadIdx.x * 8 + k];
  uint32_t acc = 0;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      uint32_t v;
      if (FORM == 0) v = (uint32_t)((double)t[k] + 0.5);
      else v = (uint32_t)rpi(t[k]);
      acc += v;
      t[k] = __uint_as_float(__float_as_uint(t[k]) ^ (v & 1u));   // (keeps the loop from being hoisted)
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
int main() {
  unsigned long long* bad; uint32_t* first;
  (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 4);
  hipMemset(bad, 0, 8); hipMemset(first, 0xff, 4);
  hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, bad, first);
  unsigned long long hb; uint32_t hf;
  hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, first, 4, hipMemcpyDeviceToHost);
  printf("floats in [0, 2^31) and -0: %llu disagree with (uint32_t)((double)t + 0.5); first bit pattern 0x%08x\n", hb, hf);
  float* in; uint32_t* out;
  hipMalloc(&in, 256 * 8 * 4); hipMalloc(&out, 1024 * 256 * 4);
  float h[2048];
  for (int i = 0; i < 2048; ++i) h[i] = 1000.0f * (float)i / 2048.0f + 0.37f;
  hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int form = 0; form < 2; ++form) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (form == 0) hipLaunchKernelGGL(k_rate<0>, dim3(1024), dim3(256), 0, 0, in, out, 4096);
      else hipLaunchKernelGGL(k_rate<1>, dim3(1024), dim3(256), 0, 0, in, out, 4096);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("%s: %.3f ms for %d indices per lane x 1024 blocks x 256 lanes\n", form ? "v_cvt_rpi_i32_f32" : "cvt_f64 + add_f64 + cvt_u32_f64", ms, 4096 * 8);
    }
  }
  return hb != 0;
}
