// Does it matter WHICH XCD writes (reads) a 4 KiB piece of memory?  (scripts/ab, never shipped; round 4)
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ab/xcd_affinity scripts/ab/xcd_affinity.hip && scripts/ab/xcd_affinity
// Round 2 found that one-shot blocks of ONE 16-byte store per thread fill memory at 6.9 TB/s while every long-lived form and every
// form with more stores per thread stops at 5.0-6.1 (profiles/r02_store_and_read_patterns.txt), and did not find out why.  The
// hypothesis tested here: workgroups are dealt to the 8 XCDs round-robin (block b -> XCD b mod 8) and memory is interleaved over
// the HBM stacks in pieces of a few KiB, so a launch whose block b writes piece b keeps every XCD on "its own" stack(s), and any
// other assignment of pieces to blocks does not.  If that is it, a long-lived block that walks pieces b, b + G, b + 2G, ... with
// G a multiple of 8 (times the interleave) keeps the property -- and can afford a table prologue.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xfu;
}
__global__ void k_where(uint32_t* out) { if (threadIdx.x == 0) out[blockIdx.x] = xcc_id(); }

// piece (PIECE bytes, PIECE / 16 threads... here always 256 threads x 16 B x U) that block b takes, as a bijection of [0, 2^LOG)
template <int MODE> __device__ __forceinline__ uint32_t perm(uint32_t b, uint32_t mask) {
  if (MODE == 0) return b;
  if (MODE == 1) return b ^ 1u;                                   // neighbours swapped
  if (MODE == 2) return (b & ~7u) | ((b + 1u) & 7u);               // rotated by one inside every group of 8
  if (MODE == 3) return (b & ~7u) | ((b + 4u) & 7u);               // rotated by four
  if (MODE == 4) return (b * 0x9E3779B1u) & mask;                  // scattered (odd multiplier: a bijection mod 2^k)
  if (MODE == 5) return (b & ~63u) | ((b & 7u) << 3) | ((b >> 3) & 7u);   // XCD x takes pieces 8x .. 8x+7 of every 64
  if (MODE == 6) return (b & ~7u) | (7u - (b & 7u));               // reversed inside every group of 8
  if (MODE == 7) return (b & ~15u) | ((b & 7u) << 1) | ((b >> 3) & 1u);   // XCD x takes pieces 2x, 2x+1 of every 16
  return b;
}
// one-shot blocks: 256 threads, U stores of 16 B per thread, block b writes piece perm(b) of U x 4 KiB
template <int MODE, int U, bool NT> __global__ void __launch_bounds__(256) k_fill(uint4* out, uint32_t mask) {
  const uint32_t piece = perm<MODE>(blockIdx.x, mask);
  uint4* p = out + (size_t)piece * 256u * U + threadIdx.x;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const u32x4 v = {piece, (uint32_t)j, 2u, 3u};
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p + j * 256));
    else *reinterpret_cast<u32x4*>(p + j * 256) = v;
  }
}
// long-lived blocks: block b writes pieces b, b + G, b + 2G ... (4 KiB each); G = gridDim.x
template <bool NT> __global__ void __launch_bounds__(256) k_fill_stride(uint4* out, uint32_t npieces) {
  for (uint32_t piece = blockIdx.x; piece < npieces; piece += gridDim.x) {
    const u32x4 v = {piece, 1u, 2u, 3u};
    uint4* p = out + (size_t)piece * 256u + threadIdx.x;
    if (NT) __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = v;
  }
}
// the same with blocks of BLOCK threads walking pieces of BLOCK x 16 B
template <int BLOCK> __global__ void __launch_bounds__(BLOCK) k_fill_stride_b(uint4* out, uint32_t npieces) {
  for (uint32_t piece = blockIdx.x; piece < npieces; piece += gridDim.x) {
    const u32x4 v = {piece, 1u, 2u, 3u};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x4*>(out + (size_t)piece * BLOCK + threadIdx.x));
  }
}
// one-shot reads
template <int MODE> __global__ void __launch_bounds__(256) k_read(const uint4* in, uint32_t mask, uint32_t* sink) {
  const uint32_t piece = perm<MODE>(blockIdx.x, mask);
  const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + (size_t)piece * 256u + threadIdx.x));
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) sink[threadIdx.x] = v.x;
}
__global__ void __launch_bounds__(256) k_read_stride(const uint4* in, uint32_t npieces, uint32_t* sink) {
  uint32_t acc = 0;
  for (uint32_t piece = blockIdx.x; piece < npieces; piece += gridDim.x) {
    const u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(in + (size_t)piece * 256u + threadIdx.x));
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <class F> float timed(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
template <int MODE, int U, bool NT> void fill_line(uint4* out, size_t bytes, const char* what) {
  const uint32_t n = (uint32_t)(bytes / (4096u * U));
  const float ms = timed([&] { hipLaunchKernelGGL((k_fill<MODE, U, NT>), dim3(n), dim3(256), 0, 0, out, n - 1u); });
  printf("one-shot fill, %d x 16 B per thread, %-5s piece(b) = %-44s %.3f ms %5.0f GB/s\n", U, NT ? "nt," : "plain,", what, ms, bytes / 1e9 / ms * 1e3);
}
template <int MODE> void read_line(const uint4* in, size_t bytes, uint32_t* sink, const char* what) {
  const uint32_t n = (uint32_t)(bytes / 4096u);
  const float ms = timed([&] { hipLaunchKernelGGL((k_read<MODE>), dim3(n), dim3(256), 0, 0, in, n - 1u, sink); });
  printf("one-shot read, 16 B per thread, piece(b) = %-44s %.3f ms %5.0f GB/s\n", what, ms, bytes / 1e9 / ms * 1e3);
}
int main() {
  const size_t bytes = (size_t)2 << 30;   // 2 GiB = 2^19 pieces of 4 KiB
  uint4* out; CK(hipMalloc(&out, bytes)); CK(hipMemset(out, 1, bytes));
  uint32_t* sink; CK(hipMalloc(&sink, 4096));
  {
    uint32_t* w; CK(hipMalloc(&w, 64 * 4));
    hipLaunchKernelGGL(k_where, dim3(64), dim3(64), 0, 0, w);
    uint32_t h[64]; CK(hipMemcpy(h, w, sizeof(h), hipMemcpyDeviceToHost));
    printf("XCC_ID of blocks 0..63:"); for (int i = 0; i < 64; ++i) printf(" %u", h[i]); printf("\n");
    printf("buffer at %p\n", (void*)out);
  }
  fill_line<0, 1, false>(out, bytes, "b (linear)");
  fill_line<0, 1, true>(out, bytes, "b (linear)");
  fill_line<1, 1, true>(out, bytes, "b ^ 1");
  fill_line<2, 1, true>(out, bytes, "rotated by 1 within groups of 8");
  fill_line<3, 1, true>(out, bytes, "rotated by 4 within groups of 8");
  fill_line<6, 1, true>(out, bytes, "reversed within groups of 8");
  fill_line<4, 1, true>(out, bytes, "scattered (b * odd mod 2^19)");
  fill_line<5, 1, true>(out, bytes, "XCD x <- pieces 8x..8x+7 of every 64");
  fill_line<7, 1, true>(out, bytes, "XCD x <- pieces 2x, 2x+1 of every 16");
  fill_line<0, 2, true>(out, bytes, "b (8 KiB per block)");
  fill_line<0, 4, true>(out, bytes, "b (16 KiB per block)");
  fill_line<0, 8, true>(out, bytes, "b (32 KiB per block)");
  fill_line<0, 16, true>(out, bytes, "b (64 KiB per block)");
  fill_line<0, 4, false>(out, bytes, "b (16 KiB per block)");
  const uint32_t np = (uint32_t)(bytes / 4096u);
  for (unsigned g : {256u, 512u, 1024u, 2048u, 4096u, 8192u, 2047u, 2049u, 2052u, 4100u, 16384u, 65536u}) {
    const float ms = timed([&] { hipLaunchKernelGGL((k_fill_stride<true>), dim3(g), dim3(256), 0, 0, out, np); });
    const float ms2 = timed([&] { hipLaunchKernelGGL((k_fill_stride<false>), dim3(g), dim3(256), 0, 0, out, np); });
    printf("long-lived fill, %5u blocks of 256, pieces b, b+G, ...: nt %.3f ms %5.0f GB/s | plain %.3f ms %5.0f GB/s\n", g, ms, bytes / 1e9 / ms * 1e3, ms2, bytes / 1e9 / ms2 * 1e3);
  }
  for (unsigned g : {1024u, 2048u, 4096u}) {
    const float ms = timed([&] { hipLaunchKernelGGL((k_fill_stride_b<512>), dim3(g), dim3(512), 0, 0, out, np / 2); });
    const float ms2 = timed([&] { hipLaunchKernelGGL((k_fill_stride_b<1024>), dim3(g), dim3(1024), 0, 0, out, np / 4); });
    const float ms3 = timed([&] { hipLaunchKernelGGL((k_fill_stride_b<128>), dim3(g), dim3(128), 0, 0, out, np * 2); });
    const float ms4 = timed([&] { hipLaunchKernelGGL((k_fill_stride_b<64>), dim3(g), dim3(64), 0, 0, out, np * 4); });
    printf("long-lived fill nt, %5u blocks: of 512 (8 KiB pieces) %5.0f | of 1024 (16 KiB) %5.0f | of 128 (2 KiB) %5.0f | of 64 (1 KiB) %5.0f GB/s\n", g,
           bytes / 1e9 / ms * 1e3, bytes / 1e9 / ms2 * 1e3, bytes / 1e9 / ms3 * 1e3, bytes / 1e9 / ms4 * 1e3);
  }
  read_line<0>(out, bytes, sink, "b (linear)");
  read_line<1>(out, bytes, sink, "b ^ 1");
  read_line<3>(out, bytes, sink, "rotated by 4 within groups of 8");
  read_line<4>(out, bytes, sink, "scattered (b * odd mod 2^19)");
  for (unsigned g : {1024u, 2048u, 4096u, 2047u, 4100u}) {
    const float ms = timed([&] { hipLaunchKernelGGL(k_read_stride, dim3(g), dim3(256), 0, 0, out, np, sink); });
    printf("long-lived read nt, %5u blocks of 256, pieces b, b+G, ...: %.3f ms %5.0f GB/s\n", g, ms, bytes / 1e9 / ms * 1e3);
  }
  return 0;
}
