// Write-only / read-only floors of the access patterns of k_apply_s4 on this box (scripts/ab, never shipped).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/store_pattern scripts/ab/store_pattern.hip && /tmp/store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t W = 3840, H = 2160, MW = 960, MH = 540, N = 64;

template <bool NT> __device__ __forceinline__ void st(uint4* p, uint4 v) {
  if (NT) __builtin_nontemporal_store((u32x4){v.x, v.y, v.z, v.w}, reinterpret_cast<u32x4*>(p)); else *p = v;
}
// 0: linear fill, 16 B per lane, grid-stride
template <bool NT> __global__ void __launch_bounds__(512) k_linear(uint4* out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) st<NT>(out + i, make_uint4(i, 1, 2, 3));
}
// 1: apply's pattern: thread = map cell, 4 rows x 16 B; block walks cpt consecutive chunks of 512 cells (mode 0) or grid-strides (mode 1)
template <bool NT> __global__ void __launch_bounds__(512) k_cells(uint32_t* out, uint32_t cpt, int mode) {
  uint32_t* img = out + (size_t)blockIdx.y * W * H;
  const uint32_t total = MW * MH;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = mode == 0 ? (blockIdx.x * cpt + it) * 512u + threadIdx.x : (it * gridDim.x + blockIdx.x) * 512u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t cy = idx / MW, cx = idx - cy * MW;
    for (int oy = 0; oy < 4; ++oy) st<NT>(reinterpret_cast<uint4*>(img + (4u * cy + oy) * W + 4u * cx), make_uint4(idx, oy, 2, 3));
  }
}
// apply's pattern with a chip-wide stride: G long-lived blocks, block b takes the 512-cell chunks b, b + G, b + 2G, ... of the whole
// batch (image-major), so that the chunks in flight at any time are neighbours -- what short-lived blocks get for free
template <bool NT> __global__ void __launch_bounds__(512) k_cells_chip(uint32_t* out) {
  const uint32_t per_img = (MW * MH + 511u) / 512u, total = per_img * N;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    const uint32_t im = id / per_img, idx = (id - im * per_img) * 512u + threadIdx.x;
    if (idx >= MW * MH) continue;
    uint32_t* img = out + (size_t)im * W * H;
    const uint32_t cy = idx / MW, cx = idx - cy * MW;
    for (int oy = 0; oy < 4; ++oy) st<NT>(reinterpret_cast<uint4*>(img + (4u * cy + oy) * W + 4u * cx), make_uint4(idx, oy, 2, 3));
  }
}
// the same with a thread per cell ROW (a wave = 1 KiB of one image row, 16 waves = a row and a bit): blocks of 1024, chip-wide stride
template <bool NT> __global__ void __launch_bounds__(1024) k_rows_chip(uint32_t* out) {
  const uint32_t per_row = W / 4u, per_img = (per_row * H + 1023u) / 1024u, total = per_img * N;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    const uint32_t im = id / per_img, idx = (id - im * per_img) * 1024u + threadIdx.x;
    if (idx >= per_row * H) continue;
    st<NT>(reinterpret_cast<uint4*>(out + (size_t)im * W * H) + idx, make_uint4(idx, 1, 2, 3));
  }
}
// long-lived blocks that take the next chunk of U x 4 KiB from one counter when they are ready for it (in-flight chunks stay
// neighbours whatever the blocks' speeds); SCOPE 0: device-scope atomic, 1: workgroup-scope (executes in the XCD's L2 -- not a
// correct queue across XCDs, timing only)
template <int U, int SCOPE> __global__ void __launch_bounds__(256) k_fill_dyn(uint4* out, size_t n16, uint32_t* counter) {
  __shared__ uint32_t s_id;
  const uint32_t chunks = (uint32_t)((n16 + 256u * U - 1) / (256u * U));
  for (;;) {
    if (threadIdx.x == 0)
      s_id = SCOPE == 0 ? atomicAdd(counter, 1u) : __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __syncthreads();
    const uint32_t id = s_id;
    __syncthreads();
    if (id >= chunks) return;
    const size_t base = (size_t)id * U * 256u + threadIdx.x;
#pragma unroll
    for (int j = 0; j < U; ++j) { const size_t i = base + (size_t)j * 256u; if (i < n16) st<false>(out + i, make_uint4((uint32_t)i, 1, 2, 3)); }
  }
}
// linear order, long-lived blocks, at most K stores of a wave in flight (s_waitcnt vmcnt): does the depth of a wave's store queue
// -- how far a fast wave runs ahead of its neighbours -- decide the rate?
template <int K> __global__ void __launch_bounds__(256) k_rows_wait(uint4* out, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) {
    st<false>(out + i, make_uint4((uint32_t)i, 1, 2, 3));
    if (K == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (K == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
    else if (K == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  }
}
// block-contiguous linear fill: each block writes one chunk of `per` x 8 KiB, U stores in flight per thread
template <int MODE> __device__ __forceinline__ void st_mode(uint4* p, uint4 vv) {
  const u32x4 v = {vv.x, vv.y, vv.z, vv.w};
  if (MODE == 0) *p = vv;
  else if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" :: "v"(p), "v"(v) : "memory");
  else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
  else if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
  else if (MODE == 4) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" :: "v"(p), "v"(v) : "memory");
  else if (MODE == 5) asm volatile("global_store_dwordx4 %0, %1, off sc0 nt" :: "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" :: "v"(p), "v"(v) : "memory");
}
template <int MODE> __global__ void __launch_bounds__(512) k_chunk(uint4* out, uint32_t per) {
  uint4* p = out + (size_t)blockIdx.x * per * 512 + threadIdx.x;
  for (uint32_t i = 0; i < per; i += 4) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st_mode<MODE>(p + (size_t)(i + k) * 512, make_uint4(i, k, 2, 3));
  }
}
template <int MODE> __global__ void __launch_bounds__(512) k_cells_mode(uint32_t* out, uint32_t cpt) {
  uint32_t* img = out + (size_t)blockIdx.y * W * H;
  const uint32_t total = MW * MH;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = (blockIdx.x * cpt + it) * 512u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t cy = idx / MW, cx = idx - cy * MW;
    for (int oy = 0; oy < 4; ++oy) st_mode<MODE>(reinterpret_cast<uint4*>(img + (4u * cy + oy) * W + 4u * cx), make_uint4(idx, oy, 2, 3));
  }
}
// torch's fill reaches 6.9 TB/s on this chip where every pattern above stops at 5.6: what is different?  BLOCK threads, each
// thread writes U consecutive-by-block 16 B words (word j of thread t at (blk * U + j) * BLOCK + t); CONST: every word the same
template <int BLOCK, int U, bool CONST, bool NT> __global__ void __launch_bounds__(BLOCK) k_fill(uint4* out, size_t n16) {
  const size_t base = (size_t)blockIdx.x * U * BLOCK + threadIdx.x;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const size_t i = base + (size_t)j * BLOCK;
    if (i < n16) st<NT>(out + i, CONST ? make_uint4(0x40000000u, 0x40000000u, 0x40000000u, 0x40000000u) : make_uint4((uint32_t)i, 1, 2, 3));
  }
}
// F16 output of apply: 8 B per pixel, a cell row = 32 B per lane.  PAIR 0: the kernel's pattern, two 16 B stores per lane at a 32 B
// stride (each instruction fills half of every line it touches); PAIR 1: each instruction writes 1 KiB contiguous per wave (what a
// lane exchange in front of the stores would give)
template <int PAIR> __global__ void __launch_bounds__(512) k_cells_f16(uint2* out, uint32_t cpt) {
  uint2* img = out + (size_t)blockIdx.y * W * H;
  const uint32_t total = MW * MH, lane = threadIdx.x & 63u;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = (blockIdx.x * cpt + it) * 512u + threadIdx.x;
    if (idx >= total) return;
    const uint32_t cy = idx / MW, cx = idx - cy * MW;
    for (int oy = 0; oy < 4; ++oy) {
      uint4* p = reinterpret_cast<uint4*>(img + (4u * cy + oy) * W + 4u * cx);
      if (PAIR == 0) { st<true>(p, make_uint4(idx, oy, 2, 3)); st<true>(p + 1, make_uint4(idx, oy, 4, 5)); }
      else { uint4* q = p - 2 * lane + lane; st<true>(q, make_uint4(idx, oy, 2, 3)); st<true>(q + 64, make_uint4(idx, oy, 4, 5)); }
    }
  }
}
// 2: thread = 4 pixels of one row (a wave = 1 KiB of one row), rows in order: the image written strictly linearly
// 3: read pattern of apply (Y 4 x 4 B, U/V 2 x 2 B each, 4 map bytes), result folded into one rare store
__global__ void __launch_bounds__(512) k_cells_read(const uint8_t* y, const uint8_t* u, const uint8_t* map, uint32_t* sink, uint32_t cpt, int mode) {
  const uint8_t* iy = y + (size_t)blockIdx.y * W * H; const uint8_t* iu = u + (size_t)blockIdx.y * (W / 2) * H; const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint8_t* im = map + (size_t)blockIdx.y * MW * MH;
  const uint32_t total = MW * MH; uint32_t acc = 0;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = mode == 0 ? (blockIdx.x * cpt + it) * 512u + threadIdx.x : (it * gridDim.x + blockIdx.x) * 512u + threadIdx.x;
    if (idx >= total) break;
    const uint32_t cy = idx / MW, cx = idx - cy * MW;
    for (int r = 0; r < 4; ++r) acc ^= __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(iy + (4u * cy + r) * W + 4u * cx));
    for (int r = 0; r < 2; ++r) { acc ^= *reinterpret_cast<const uint16_t*>(iu + (2u * cy + r) * (W / 2) + 2u * cx); acc ^= *reinterpret_cast<const uint16_t*>(iv + (2u * cy + r) * (W / 2) + 2u * cx); }
    const uint32_t xu = min(cx + 1u, MW - 1u), yu = min(cy + 1u, MH - 1u);
    acc ^= im[cy * MW + cx] ^ im[yu * MW + cx] ^ im[cy * MW + xu] ^ im[yu * MW + xu];
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

// generate's read pattern: thread = 2 map pixels = 8 x 4 px of both images: P010 Y 4 x 16 B, P010 UV 2 x 16 B, Y8 4 x 8 B, U / V 2 x 4 B each
template <int BLOCK, bool NT> __global__ void __launch_bounds__(BLOCK) k_gen_read(const uint16_t* hy, const uint16_t* huv, const uint8_t* y, const uint8_t* u, uint32_t* sink, uint32_t tiles, int image_major) {
  const uint32_t img = image_major ? blockIdx.y : blockIdx.x, blk = image_major ? blockIdx.x : blockIdx.y;
  const uint16_t* ihy = hy + (size_t)img * W * H; const uint16_t* ihuv = huv + (size_t)img * W * (H / 2);
  const uint8_t* iy = y + (size_t)img * W * H; const uint8_t* iu = u + (size_t)img * (W / 2) * H; const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint32_t ppr = MW / 2, total = ppr * MH; uint32_t acc = 0;
  for (uint32_t t = 0; t < tiles; ++t) {
    const uint32_t idx = (blk * tiles + t) * BLOCK + threadIdx.x;
    if (idx >= total) break;
    const uint32_t my = idx / ppr, pr = idx - my * ppr;
    for (int r = 0; r < 4; ++r) {
      const u32x4* p = reinterpret_cast<const u32x4*>(ihy + (4u * my + r) * W + 8u * pr);
      const u32x4 q = NT ? __builtin_nontemporal_load(p) : *p; acc ^= q.x ^ q.y ^ q.z ^ q.w;
      const uint2* p2 = reinterpret_cast<const uint2*>(iy + (4u * my + r) * W + 8u * pr);
      const uint2 q2 = *p2; acc ^= q2.x ^ q2.y;
    }
    for (int r = 0; r < 2; ++r) {
      const u32x4* p = reinterpret_cast<const u32x4*>(ihuv + (2u * my + r) * W + 8u * pr);
      const u32x4 q = NT ? __builtin_nontemporal_load(p) : *p; acc ^= q.x ^ q.y ^ q.z ^ q.w;
      acc ^= *reinterpret_cast<const uint32_t*>(iu + (2u * my + r) * (W / 2) + 4u * pr) ^ *reinterpret_cast<const uint32_t*>(iv + (2u * my + r) * (W / 2) + 4u * pr);
    }
  }
  if (acc == 0x12345678u) sink[threadIdx.x] = acc;
}

template <class F> float timed(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / 20;
}
template <int BLOCK, int U, bool CONST, bool NT> float run_fill(uint4* out, size_t n16) {
  const unsigned grid = (unsigned)((n16 + (size_t)BLOCK * U - 1) / ((size_t)BLOCK * U));
  return timed([&] { hipLaunchKernelGGL((k_fill<BLOCK, U, CONST, NT>), dim3(grid), dim3(BLOCK), 0, 0, out, n16); });
}
int main() {
  const size_t obytes = (size_t)N * W * H * 4;
  uint32_t* out; CK(hipMalloc(&out, obytes));
  uint8_t *y, *u, *map; CK(hipMalloc(&y, (size_t)N * W * H)); CK(hipMalloc(&u, (size_t)N * W * H / 2)); CK(hipMalloc(&map, (size_t)N * MW * MH));
  CK(hipMemset(y, 1, (size_t)N * W * H)); CK(hipMemset(u, 2, (size_t)N * W * H / 2)); CK(hipMemset(map, 3, (size_t)N * MW * MH));
  const double gb = obytes / 1e9, rgb = ((double)N * W * H * 1.5 + (double)N * MW * MH) / 1e9;
  float t;
  {
    const size_t n16 = obytes / 16;
    printf("fill  64 thr x 4, varying, plain  %.0f GB/s\n", gb / run_fill<64, 4, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 128 thr x 4, varying, plain  %.0f GB/s\n", gb / run_fill<128, 4, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 1, varying, plain  %.0f GB/s\n", gb / run_fill<256, 1, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 4, varying, plain  %.0f GB/s\n", gb / run_fill<256, 4, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 4, constant, plain %.0f GB/s\n", gb / run_fill<256, 4, true, false>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 4, varying, nt     %.0f GB/s\n", gb / run_fill<256, 4, false, true>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 4, constant, nt    %.0f GB/s\n", gb / run_fill<256, 4, true, true>((uint4*)out, n16) * 1e3);
    printf("fill 256 thr x 16, varying, plain %.0f GB/s\n", gb / run_fill<256, 16, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 512 thr x 8, varying, plain  %.0f GB/s\n", gb / run_fill<512, 8, false, false>((uint4*)out, n16) * 1e3);
    printf("fill 1024 thr x 4, varying, plain %.0f GB/s\n", gb / run_fill<1024, 4, false, false>((uint4*)out, n16) * 1e3);
  }
  t = timed([&] { hipLaunchKernelGGL(k_linear<true>, dim3(2048), dim3(512), 0, 0, (uint4*)out, obytes / 16); }); printf("linear nt, 2048 blocks          %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { hipLaunchKernelGGL(k_linear<false>, dim3(2048), dim3(512), 0, 0, (uint4*)out, obytes / 16); }); printf("linear plain, 2048 blocks       %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
  t = timed([&] { hipLaunchKernelGGL(k_linear<true>, dim3(512), dim3(512), 0, 0, (uint4*)out, obytes / 16); }); printf("linear nt, 512 blocks           %.3f ms %.0f GB/s\n", t, gb / t * 1e3);
  {
    const char* names[7] = {"plain", "nt", "sc1", "sc0 sc1", "sc0 sc1 nt", "sc0 nt", "sc1 nt"};
    float r[7][3];
    for (int per_i = 0; per_i < 3; ++per_i) {
      const uint32_t per = per_i == 0 ? 8 : per_i == 1 ? 64 : 512;   // 64 KiB / 512 KiB / 4 MiB per block
      const dim3 grid((unsigned)(obytes / 16 / 512 / per));
      r[0][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<0>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[1][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<1>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[2][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<2>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[3][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<3>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[4][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<4>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[5][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<5>, grid, dim3(512), 0, 0, (uint4*)out, per); });
      r[6][per_i] = timed([&] { hipLaunchKernelGGL(k_chunk<6>, grid, dim3(512), 0, 0, (uint4*)out, per); });
    }
    for (int m = 0; m < 7; ++m) printf("chunk fill %-11s 64K/512K/4M per block: %.0f %.0f %.0f GB/s\n", names[m], gb / r[m][0] * 1e3, gb / r[m][1] * 1e3, gb / r[m][2] * 1e3);
    const dim3 grid((MW * MH + 512 * 32 - 1) / (512 * 32), N);
    float c[7];
    c[0] = timed([&] { hipLaunchKernelGGL(k_cells_mode<0>, grid, dim3(512), 0, 0, out, 32u); });
    c[1] = timed([&] { hipLaunchKernelGGL(k_cells_mode<1>, grid, dim3(512), 0, 0, out, 32u); });
    c[2] = timed([&] { hipLaunchKernelGGL(k_cells_mode<2>, grid, dim3(512), 0, 0, out, 32u); });
    c[3] = timed([&] { hipLaunchKernelGGL(k_cells_mode<3>, grid, dim3(512), 0, 0, out, 32u); });
    c[4] = timed([&] { hipLaunchKernelGGL(k_cells_mode<4>, grid, dim3(512), 0, 0, out, 32u); });
    c[5] = timed([&] { hipLaunchKernelGGL(k_cells_mode<5>, grid, dim3(512), 0, 0, out, 32u); });
    c[6] = timed([&] { hipLaunchKernelGGL(k_cells_mode<6>, grid, dim3(512), 0, 0, out, 32u); });
    for (int m = 0; m < 7; ++m) printf("cells cpt 32 %-11s %.3f ms %.0f GB/s\n", names[m], c[m], gb / c[m] * 1e3);
  }
  {
    uint2* o8; CK(hipMalloc(&o8, (size_t)N * W * H * 8));
    const dim3 grid((MW * MH + 512 * 32 - 1) / (512 * 32), N);
    const double gb8 = (double)N * W * H * 8 / 1e9;
    t = timed([&] { hipLaunchKernelGGL(k_cells_f16<0>, grid, dim3(512), 0, 0, o8, 32u); }); printf("F16 cells, 2 x 16 B per lane at 32 B stride   %.3f ms %.0f GB/s\n", t, gb8 / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_cells_f16<1>, grid, dim3(512), 0, 0, o8, 32u); }); printf("F16 cells, each store 1 KiB contiguous per wave %.3f ms %.0f GB/s\n", t, gb8 / t * 1e3);
    CK(hipFree(o8));
  }
  {
    uint16_t *hy, *huv; CK(hipMalloc(&hy, (size_t)N * W * H * 2)); CK(hipMalloc(&huv, (size_t)N * W * H)); CK(hipMemset(hy, 1, (size_t)N * W * H * 2)); CK(hipMemset(huv, 1, (size_t)N * W * H));
    const double ggb = (double)N * W * H * 4.5 / 1e9;
    for (int im = 0; im < 2; ++im)
      for (uint32_t tiles : {1u, 4u, 16u}) {
        const uint32_t tot = (MW / 2) * MH;
        const uint32_t b256 = (tot + 256 * tiles - 1) / (256 * tiles), b512 = (tot + 512 * tiles - 1) / (512 * tiles);
        t = timed([&] { hipLaunchKernelGGL((k_gen_read<256, true>), im ? dim3(b256, N) : dim3(N, b256), dim3(256), 0, 0, hy, huv, y, u, out, tiles, im); });
        printf("generate reads nt, block 256, tiles %2u, %s  %.3f ms %.0f GB/s\n", tiles, im ? "image-major" : "interleaved", t, ggb / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL((k_gen_read<512, true>), im ? dim3(b512, N) : dim3(N, b512), dim3(512), 0, 0, hy, huv, y, u, out, tiles, im); });
        printf("generate reads nt, block 512, tiles %2u, %s  %.3f ms %.0f GB/s\n", tiles, im ? "image-major" : "interleaved", t, ggb / t * 1e3);
        t = timed([&] { hipLaunchKernelGGL((k_gen_read<256, false>), im ? dim3(b256, N) : dim3(N, b256), dim3(256), 0, 0, hy, huv, y, u, out, tiles, im); });
        printf("generate reads plain, block 256, tiles %2u, %s  %.3f ms %.0f GB/s\n", tiles, im ? "image-major" : "interleaved", t, ggb / t * 1e3);
      }
  }
  {
    uint32_t* ctr; CK(hipMalloc(&ctr, 4));
    const size_t n16 = obytes / 16;
    for (unsigned g : {1024u, 2048u}) {
      t = timed([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL((k_fill_dyn<4, 0>), dim3(g), dim3(256), 0, 0, (uint4*)out, n16, ctr); });
      printf("fill, chunks of 16 KiB from a counter (device scope), %u blocks  %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
      t = timed([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL((k_fill_dyn<16, 0>), dim3(g), dim3(256), 0, 0, (uint4*)out, n16, ctr); });
      printf("fill, chunks of 64 KiB from a counter (device scope), %u blocks  %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
      t = timed([&] { hipMemsetAsync(ctr, 0, 4, 0); hipLaunchKernelGGL((k_fill_dyn<4, 1>), dim3(g), dim3(256), 0, 0, (uint4*)out, n16, ctr); });
      printf("fill, chunks of 16 KiB from a counter (L2 scope), %u blocks      %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    }
  }
  for (unsigned g : {2048u, 4096u, 8192u}) {
    const size_t n16 = obytes / 16;
    t = timed([&] { hipLaunchKernelGGL(k_rows_wait<0>, dim3(g), dim3(256), 0, 0, (uint4*)out, n16); }); printf("linear loop, %u blocks, vmcnt(0) after each store  %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_rows_wait<1>, dim3(g), dim3(256), 0, 0, (uint4*)out, n16); }); printf("linear loop, %u blocks, vmcnt(1)                   %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_rows_wait<3>, dim3(g), dim3(256), 0, 0, (uint4*)out, n16); }); printf("linear loop, %u blocks, vmcnt(3)                   %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_rows_wait<99>, dim3(g), dim3(256), 0, 0, (uint4*)out, n16); }); printf("linear loop, %u blocks, no wait                    %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
  }
  for (unsigned g : {256u, 512u, 1024u, 2048u, 4096u}) {
    t = timed([&] { hipLaunchKernelGGL(k_cells_chip<true>, dim3(g), dim3(512), 0, 0, out); });
    printf("cells, chip-wide stride, %4u blocks, nt    %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_cells_chip<false>, dim3(g), dim3(512), 0, 0, out); });
    printf("cells, chip-wide stride, %4u blocks, plain %.3f ms %.0f GB/s\n", g, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_rows_chip<true>, dim3(g / 2), dim3(1024), 0, 0, out); });
    printf("rows,  chip-wide stride, %4u blocks of 1024, nt    %.3f ms %.0f GB/s\n", g / 2, t, gb / t * 1e3);
    t = timed([&] { hipLaunchKernelGGL(k_rows_chip<false>, dim3(g / 2), dim3(1024), 0, 0, out); });
    printf("rows,  chip-wide stride, %4u blocks of 1024, plain %.3f ms %.0f GB/s\n", g / 2, t, gb / t * 1e3);
  }
  for (uint32_t cpt : {1u, 2u, 4u, 8u, 32u})
    for (int mode = 0; mode < 2; ++mode) {
      const dim3 grid((MW * MH + 512 * cpt - 1) / (512 * cpt), N);
      t = timed([&] { hipLaunchKernelGGL(k_cells<true>, grid, dim3(512), 0, 0, out, cpt, mode); });
      printf("cells nt, cpt %3u, %s   %.3f ms %.0f GB/s\n", cpt, mode ? "grid-stride " : "block-chunks", t, gb / t * 1e3);
      t = timed([&] { hipLaunchKernelGGL(k_cells<false>, grid, dim3(512), 0, 0, out, cpt, mode); });
      printf("cells plain, cpt %3u, %s   %.3f ms %.0f GB/s\n", cpt, mode ? "grid-stride " : "block-chunks", t, gb / t * 1e3);
      t = timed([&] { hipLaunchKernelGGL(k_cells_read, grid, dim3(512), 0, 0, y, u, map, out, cpt, mode); });
      printf("cells read, cpt %3u, %s %.3f ms %.0f GB/s\n", cpt, mode ? "grid-stride " : "block-chunks", t, rgb / t * 1e3);
    }
  return 0;
}
