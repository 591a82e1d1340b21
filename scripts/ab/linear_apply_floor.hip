// Memory floors of apply's bytes (64 x 4K: 12.4 MB YUV420 + 0.5 MB map in, 33.2 MB RGBA1010102 out per frame) under different
// assignments of pixels to threads and of blocks to time -- loads, one XOR, stores; no arithmetic, no tables (scripts/ab, never
// shipped; round 4).     hipcc --offload-arch=gfx950 -O3 -o scripts/ab/linear_apply_floor scripts/ab/linear_apply_floor.hip
//
// Why: scripts/ab/xcd_affinity.hip shows a fill running at 6.7 TB/s when one-shot blocks write memory in address order and at
// 4.4-5.8 in every other form; k_apply_s4's own floor (its loads and stores without arithmetic, X1) is 5.5 TB/s.  Is that the
// price of "thread = map cell" (a lane stores 4 rows x 16 B, a block 4 x 8 KiB) and of long-lived blocks -- would a kernel whose
// thread is ONE row of a cell (4 pixels, one 16-byte store), in one-shot blocks that sweep each image linearly, move apply's bytes
// faster?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr uint32_t W = 3840, H = 2160, MW = 960, MH = 540, N = 64;
struct Planes { const uint8_t* y; const uint8_t* u; const uint8_t* map; uint32_t* out; };
__device__ __forceinline__ void st_nt(uint32_t* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  __builtin_nontemporal_store((u32x4){a, b, c, d}, reinterpret_cast<u32x4*>(p));
}
struct __attribute__((packed)) U16Any { uint16_t v; };

// thread = one row of a cell: quad q of image im (row = q / MW, cell column = q % MW): Y dword, U / V ushort, the map's two
// byte pairs -> one 16-byte store.  A wave = 256 pixels of one row = 1 KiB out.
__device__ __forceinline__ void quad(const Planes& p, uint32_t im, uint32_t q) {
  const uint32_t row = q / MW, cx = q - row * MW, cy = row >> 2;
  const uint8_t* iy = p.y + (size_t)im * W * H;
  const uint8_t* iu = p.u + (size_t)im * (W / 2) * H;
  const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint8_t* imap = p.map + (size_t)im * MW * MH;
  uint32_t x = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(iy + row * W + 4u * cx));
  x ^= *reinterpret_cast<const uint16_t*>(iu + (row >> 1) * (W / 2) + 2u * cx);
  x ^= *reinterpret_cast<const uint16_t*>(iv + (row >> 1) * (W / 2) + 2u * cx);
  const uint32_t mx = cx - (cx + 1u == MW ? 1u : 0u), cy1 = min(cy + 1u, MH - 1u);
  x ^= reinterpret_cast<const U16Any*>(imap + cy * MW + mx)->v;
  x ^= reinterpret_cast<const U16Any*>(imap + cy1 * MW + mx)->v;
  st_nt(p.out + (size_t)im * W * H + (size_t)row * W + 4u * cx, x, x + 1u, x ^ 1u, x ^ 2u);
}
// one-shot blocks of BLOCK quads; ORDER 0: image-major (blockIdx.y = image), 1: images interleaved (blockIdx.x = image)
template <int BLOCK, int ORDER> __global__ void __launch_bounds__(BLOCK) k_quads(const Planes p) {
  const uint32_t im = ORDER == 0 ? blockIdx.y : blockIdx.x, blk = ORDER == 0 ? blockIdx.x : blockIdx.y;
  const uint32_t q = blk * BLOCK + threadIdx.x;
  if (q < MW * H) quad(p, im, q);
}
// one-shot blocks, U quads per thread (consecutive pieces of BLOCK quads)
template <int BLOCK, int U> __global__ void __launch_bounds__(BLOCK) k_quads_u(const Planes p) {
  const uint32_t im = blockIdx.y;
#pragma unroll
  for (int j = 0; j < U; ++j) {
    const uint32_t q = (blockIdx.x * U + j) * BLOCK + threadIdx.x;
    if (q < MW * H) quad(p, im, q);
  }
}
// long-lived blocks with a chip-wide stride over the whole batch's pieces of BLOCK quads (image-major)
template <int BLOCK> __global__ void __launch_bounds__(BLOCK) k_quads_stride(const Planes p) {
  const uint32_t per_img = (MW * H + BLOCK - 1u) / BLOCK, total = per_img * N;
  for (uint32_t id = blockIdx.x; id < total; id += gridDim.x) {
    const uint32_t im = id / per_img, q = (id - im * per_img) * BLOCK + threadIdx.x;
    if (q < MW * H) quad(p, im, q);
  }
}
// thread = map cell (the shipped assignment): 4 Y dwords, 2 + 2 chroma ushorts, 2 map byte pairs -> 4 stores of 16 B
__device__ __forceinline__ void cell(const Planes& p, uint32_t im, uint32_t idx) {
  const uint32_t cy = idx / MW, cx = idx - cy * MW;
  const uint8_t* iy = p.y + (size_t)im * W * H;
  const uint8_t* iu = p.u + (size_t)im * (W / 2) * H;
  const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint8_t* imap = p.map + (size_t)im * MW * MH;
  uint32_t x = 0;
  for (int r = 0; r < 4; ++r) x ^= __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(iy + (4u * cy + r) * W + 4u * cx));
  for (int r = 0; r < 2; ++r) {
    x ^= *reinterpret_cast<const uint16_t*>(iu + (2u * cy + r) * (W / 2) + 2u * cx);
    x ^= *reinterpret_cast<const uint16_t*>(iv + (2u * cy + r) * (W / 2) + 2u * cx);
  }
  const uint32_t mx = cx - (cx + 1u == MW ? 1u : 0u), cy1 = min(cy + 1u, MH - 1u);
  x ^= reinterpret_cast<const U16Any*>(imap + cy * MW + mx)->v;
  x ^= reinterpret_cast<const U16Any*>(imap + cy1 * MW + mx)->v;
  uint32_t* o = p.out + (size_t)im * W * H;
  for (int oy = 0; oy < 4; ++oy) st_nt(o + (size_t)(4u * cy + oy) * W + 4u * cx, x, x + oy, x ^ 1u, x ^ 2u);
}
template <int ORDER> __global__ void __launch_bounds__(512) k_cells(const Planes p, uint32_t cpt) {
  const uint32_t im = ORDER == 0 ? blockIdx.y : blockIdx.x, blk = ORDER == 0 ? blockIdx.x : blockIdx.y;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = (blk * cpt + it) * 512u + threadIdx.x;
    if (idx >= MW * MH) return;
    cell(p, im, idx);
  }
}
// the walk with the co-resident blocks of an image INTERLEAVED chunk by chunk: blocks come in groups of GRP, block j of a group takes
// the chunks j, j + GRP, j + 2 GRP, ... of the group's GRP x cpt chunks -- so that what the blocks of an image write at any one time
// lies side by side (as with one-shot blocks) instead of cpt x 32 KiB apart
template <int GRP> __global__ void __launch_bounds__(512) k_cells_grp(const Planes p, uint32_t cpt) {
  const uint32_t im = blockIdx.x, blk = blockIdx.y, grp = blk / GRP, j = blk % GRP;
  for (uint32_t it = 0; it < cpt; ++it) {
    const uint32_t idx = ((grp * cpt + it) * GRP + j) * 512u + threadIdx.x;
    if (idx >= MW * MH) return;
    cell(p, im, idx);
  }
}
// the walk with the inputs of the cell DEPTH steps ahead requested before the current cell is stored (the shipped kernel: DEPTH 1),
// at a resident-wave count set by `lds` bytes of dynamic shared memory per block (80 KiB: two blocks of 512 = 16 waves per CU, what
// the shipped kernel's 122 VGPRs allow; 0: as many as fit).  Is the walk's floor a matter of bytes in flight per CU?
struct CellIn { uint32_t y[4], c[4], m[2]; };
__device__ __forceinline__ void cell_load(const Planes& p, uint32_t im, uint32_t idx, CellIn& o) {
  const uint32_t cy = idx / MW, cx = idx - cy * MW;
  const uint8_t* iy = p.y + (size_t)im * W * H;
  const uint8_t* iu = p.u + (size_t)im * (W / 2) * H;
  const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint8_t* imap = p.map + (size_t)im * MW * MH;
  for (int r = 0; r < 4; ++r) o.y[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(iy + (4u * cy + r) * W + 4u * cx));
  for (int r = 0; r < 2; ++r) {
    o.c[2 * r] = *reinterpret_cast<const uint16_t*>(iu + (2u * cy + r) * (W / 2) + 2u * cx);
    o.c[2 * r + 1] = *reinterpret_cast<const uint16_t*>(iv + (2u * cy + r) * (W / 2) + 2u * cx);
  }
  const uint32_t mx = cx - (cx + 1u == MW ? 1u : 0u), cy1 = min(cy + 1u, MH - 1u);
  o.m[0] = reinterpret_cast<const U16Any*>(imap + cy * MW + mx)->v;
  o.m[1] = reinterpret_cast<const U16Any*>(imap + cy1 * MW + mx)->v;
}
__device__ __forceinline__ void cell_store(const Planes& p, uint32_t im, uint32_t idx, const CellIn& in) {
  const uint32_t cy = idx / MW, cx = idx - cy * MW;
  const uint32_t x = in.y[0] ^ in.y[1] ^ in.y[2] ^ in.y[3] ^ in.c[0] ^ in.c[1] ^ in.c[2] ^ in.c[3] ^ in.m[0] ^ in.m[1];
  uint32_t* o = p.out + (size_t)im * W * H;
  for (int oy = 0; oy < 4; ++oy) st_nt(o + (size_t)(4u * cy + oy) * W + 4u * cx, x, x + oy, x ^ 1u, x ^ 2u);
}
template <int DEPTH> __global__ void __launch_bounds__(512) k_cells_pf(const Planes p, uint32_t cpt) {
  extern __shared__ uint4 s_dummy[];
  const uint32_t im = blockIdx.x, blk = blockIdx.y, last = MW * MH - 1u;
  const uint32_t first = blk * cpt * 512u + threadIdx.x;
  if (first > last) return;
  if (s_dummy[0].x == 0x12345678u && threadIdx.x == 77u) p.out[0] = 1u;   // (keeps the allocation)
  CellIn buf[DEPTH + 1];
#pragma unroll
  for (int d = 0; d < DEPTH; ++d) cell_load(p, im, min(first + d * 512u, last), buf[d]);
  for (uint32_t it = 0; it < cpt; it += DEPTH + 1) {
#pragma unroll
    for (int j = 0; j <= DEPTH; ++j) {
      const uint32_t idx = first + (it + j) * 512u;
      if (it + j >= cpt || idx > last) return;
      cell_load(p, im, min(idx + DEPTH * 512u, last), buf[(j + DEPTH) % (DEPTH + 1)]);   // (past the end: the last cell once more)
      cell_store(p, im, idx, buf[j]);
    }
  }
}
// one-shot blocks of BLOCK cells that first copy a table of PBYTES from global memory (L2) into LDS, as a short-lived k_apply_s4 would
// have to (the cell's own loads are issued first); the cell then reads one table entry it depends on
template <int BLOCK, int PBYTES> __global__ void __launch_bounds__(BLOCK) k_cells_pro(const Planes p, const uint4* __restrict__ table) {
  __shared__ uint4 s_tab[PBYTES / 16];
  const uint32_t im = blockIdx.x, idx = blockIdx.y * BLOCK + threadIdx.x;
  const bool any = idx < MW * MH;
  const uint32_t cy = any ? idx / MW : 0u, cx = any ? idx - cy * MW : 0u;
  const uint8_t* iy = p.y + (size_t)im * W * H;
  const uint8_t* iu = p.u + (size_t)im * (W / 2) * H;
  const uint8_t* iv = iu + (size_t)(W / 2) * (H / 2);
  const uint8_t* imap = p.map + (size_t)im * MW * MH;
  uint32_t yv[4], cv[4], mv[2];
  for (int r = 0; r < 4; ++r) yv[r] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(iy + (4u * cy + r) * W + 4u * cx));
  for (int r = 0; r < 2; ++r) {
    cv[2 * r] = *reinterpret_cast<const uint16_t*>(iu + (2u * cy + r) * (W / 2) + 2u * cx);
    cv[2 * r + 1] = *reinterpret_cast<const uint16_t*>(iv + (2u * cy + r) * (W / 2) + 2u * cx);
  }
  const uint32_t mx = cx - (cx + 1u == MW ? 1u : 0u), cy1 = min(cy + 1u, MH - 1u);
  mv[0] = reinterpret_cast<const U16Any*>(imap + cy * MW + mx)->v;
  mv[1] = reinterpret_cast<const U16Any*>(imap + cy1 * MW + mx)->v;
  constexpr int kN = PBYTES / 16, kPer = (kN + BLOCK - 1) / BLOCK;
  uint4 t[kPer];
#pragma unroll
  for (int k = 0; k < kPer; ++k) { const int i = k * BLOCK + (int)threadIdx.x; t[k] = table[i < kN ? i : kN - 1]; }
#pragma unroll
  for (int k = 0; k < kPer; ++k) { const int i = k * BLOCK + (int)threadIdx.x; if (i < kN) s_tab[i] = t[k]; }
  __syncthreads();
  if (!any) return;
  uint32_t x = yv[0] ^ yv[1] ^ yv[2] ^ yv[3] ^ cv[0] ^ cv[1] ^ cv[2] ^ cv[3] ^ mv[0] ^ mv[1];
  x ^= reinterpret_cast<const uint32_t*>(s_tab)[(x * 2654435761u >> 8) % (PBYTES / 4)];
  uint32_t* o = p.out + (size_t)im * W * H;
  for (int oy = 0; oy < 4; ++oy) st_nt(o + (size_t)(4u * cy + oy) * W + 4u * cx, x, x + oy, x ^ 1u, x ^ 2u);
}
template <int BLOCK, int PBYTES> float run_pro(const Planes& p, const uint4* table);

// thread = map cell, but a BLOCK = one row of cells (960 threads = 15 waves): the block's stores are 4 full pixel rows = 60 KiB
// contiguous; one-shot, image-major
__global__ void __launch_bounds__(960) k_cellrows(const Planes p) { cell(p, blockIdx.y, blockIdx.x * MW + threadIdx.x); }

template <class F> float timed(F f);
template <int BLOCK, int PBYTES> float run_pro(const Planes& p, const uint4* table) {
  return timed([&] { hipLaunchKernelGGL((k_cells_pro<BLOCK, PBYTES>), dim3(N, (MW * MH + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, 0, p, table); });
}
template <class F> float timed(F f) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) f();
  (void)hipEventRecord(a); for (int i = 0; i < 20; ++i) f(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms / 20;
}
int main() {
  uint8_t *y, *u, *map; uint32_t* out;
  CK(hipMalloc(&y, (size_t)N * W * H)); CK(hipMalloc(&u, (size_t)N * W * H / 2)); CK(hipMalloc(&map, (size_t)N * MW * MH)); CK(hipMalloc(&out, (size_t)N * W * H * 4));
  CK(hipMemset(y, 1, (size_t)N * W * H)); CK(hipMemset(u, 2, (size_t)N * W * H / 2)); CK(hipMemset(map, 3, (size_t)N * MW * MH));
  const Planes p{y, u, map, out};
  uint4* table; CK(hipMalloc(&table, 64 * 1024)); CK(hipMemset(table, 5, 64 * 1024));
  const double gb = (double)N * (W * H * 1.5 + MW * MH + W * H * 4.0) / 1e9;
  const uint32_t quads = MW * H;
  auto line = [&](const char* what, float ms) { printf("%-100s %.4f ms %5.0f GB/s\n", what, ms, gb / ms * 1e3); };
  for (int rep = 0; rep < 2; ++rep) {
    line("thread = cell, 512-thread blocks, 32 cells per thread, images interleaved (the shipped walk's order)",
         timed([&] { hipLaunchKernelGGL((k_cells<1>), dim3(N, (MW * MH + 512 * 32 - 1) / (512 * 32)), dim3(512), 0, 0, p, 32u); }));
    line("thread = cell, 512-thread blocks, 32 cells per thread, image-major",
         timed([&] { hipLaunchKernelGGL((k_cells<0>), dim3((MW * MH + 512 * 32 - 1) / (512 * 32), N), dim3(512), 0, 0, p, 32u); }));
    line("thread = cell, 512-thread blocks, 1 cell per thread (one-shot), image-major",
         timed([&] { hipLaunchKernelGGL((k_cells<0>), dim3((MW * MH + 511) / 512, N), dim3(512), 0, 0, p, 1u); }));
    for (unsigned cpt : {2u, 4u, 8u, 16u}) {
      char s2[160];
      snprintf(s2, sizeof s2, "thread = cell, 512-thread blocks, %u cells per thread, images interleaved", cpt);
      line(s2, timed([&] { hipLaunchKernelGGL((k_cells<1>), dim3(N, (MW * MH + 512 * cpt - 1) / (512 * cpt)), dim3(512), 0, 0, p, cpt); }));
    }
    {
      const unsigned nb = (MW * MH + 512 * 32 - 1) / (512 * 32);
      line("walk (32 cells per thread, images interleaved), blocks of an image interleaved in groups of 2", timed([&] { hipLaunchKernelGGL((k_cells_grp<2>), dim3(N, (nb + 1) / 2 * 2), dim3(512), 0, 0, p, 32u); }));
      line("walk, blocks of an image interleaved in groups of 4", timed([&] { hipLaunchKernelGGL((k_cells_grp<4>), dim3(N, (nb + 3) / 4 * 4), dim3(512), 0, 0, p, 32u); }));
      line("walk, blocks of an image interleaved in groups of 8", timed([&] { hipLaunchKernelGGL((k_cells_grp<8>), dim3(N, (nb + 7) / 8 * 8), dim3(512), 0, 0, p, 32u); }));
      line("walk, blocks of an image interleaved in groups of 16", timed([&] { hipLaunchKernelGGL((k_cells_grp<16>), dim3(N, (nb + 15) / 16 * 16), dim3(512), 0, 0, p, 32u); }));
      line("walk, blocks of an image interleaved in groups of 32 (= all of them)", timed([&] { hipLaunchKernelGGL((k_cells_grp<32>), dim3(N, (nb + 31) / 32 * 32), dim3(512), 0, 0, p, 32u); }));
      const unsigned nb8 = (MW * MH + 512 * 8 - 1) / (512 * 8);
      line("walk with 8 cells per thread, groups of 8", timed([&] { hipLaunchKernelGGL((k_cells_grp<8>), dim3(N, (nb8 + 7) / 8 * 8), dim3(512), 0, 0, p, 8u); }));
      line("walk with 8 cells per thread, groups of 32", timed([&] { hipLaunchKernelGGL((k_cells_grp<32>), dim3(N, (nb8 + 31) / 32 * 32), dim3(512), 0, 0, p, 8u); }));
    }
    {
      const dim3 g(N, (MW * MH + 512 * 32 - 1) / (512 * 32));
      for (unsigned lds : {80u * 1024u, 40u * 1024u, 0u}) {
        char s3[200];
        if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_pf<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_pf<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_pf<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (lds) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_cells_pf<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        snprintf(s3, sizeof s3, "walk, %u blocks of 512 per CU, no prefetch", lds ? 160u * 1024u / lds : 4u);
        line(s3, timed([&] { hipLaunchKernelGGL((k_cells_pf<0>), g, dim3(512), lds, 0, p, 32u); }));
        snprintf(s3, sizeof s3, "walk, %u blocks of 512 per CU, next cell requested before the stores (the shipped depth)", lds ? 160u * 1024u / lds : 4u);
        line(s3, timed([&] { hipLaunchKernelGGL((k_cells_pf<1>), g, dim3(512), lds, 0, p, 32u); }));
        snprintf(s3, sizeof s3, "walk, %u blocks of 512 per CU, two cells ahead", lds ? 160u * 1024u / lds : 4u);
        line(s3, timed([&] { hipLaunchKernelGGL((k_cells_pf<2>), g, dim3(512), lds, 0, p, 32u); }));
        snprintf(s3, sizeof s3, "walk, %u blocks of 512 per CU, three cells ahead", lds ? 160u * 1024u / lds : 4u);
        line(s3, timed([&] { hipLaunchKernelGGL((k_cells_pf<3>), g, dim3(512), lds, 0, p, 33u); }));
      }
    }
    line("thread = cell, 512-thread blocks, 1 cell per thread (one-shot), images interleaved",
         timed([&] { hipLaunchKernelGGL((k_cells<1>), dim3(N, (MW * MH + 511) / 512), dim3(512), 0, 0, p, 1u); }));
    line("one-shot cells, blocks of 512, images interleaved, + 5 KiB of tables L2 -> LDS per block", run_pro<512, 5 * 1024>(p, table));
    line("one-shot cells, blocks of 512, images interleaved, + 13 KiB of tables per block", run_pro<512, 13 * 1024>(p, table));
    line("one-shot cells, blocks of 512, images interleaved, + 37 KiB of tables per block", run_pro<512, 37 * 1024>(p, table));
    line("one-shot cells, blocks of 256, images interleaved, + 5 KiB of tables per block", run_pro<256, 5 * 1024>(p, table));
    line("one-shot cells, blocks of 256, images interleaved, + 13 KiB of tables per block", run_pro<256, 13 * 1024>(p, table));
    line("one-shot cells, blocks of 1024, images interleaved, + 5 KiB of tables per block", run_pro<1024, 5 * 1024>(p, table));
    line("one-shot cells, blocks of 1024, images interleaved, + 13 KiB of tables per block", run_pro<1024, 13 * 1024>(p, table));
    line("one-shot cells, blocks of 1024, images interleaved, + 37 KiB of tables per block", run_pro<1024, 37 * 1024>(p, table));
    line("thread = cell, block = one row of cells (960 threads, 60 KiB contiguous out), one-shot, image-major",
         timed([&] { hipLaunchKernelGGL(k_cellrows, dim3(MH, N), dim3(960), 0, 0, p); }));
    line("thread = cell row (4 px, one store), one-shot blocks of 256, image-major (each image swept linearly)",
         timed([&] { hipLaunchKernelGGL((k_quads<256, 0>), dim3((quads + 255) / 256, N), dim3(256), 0, 0, p); }));
    line("thread = cell row, one-shot blocks of 256, images interleaved",
         timed([&] { hipLaunchKernelGGL((k_quads<256, 1>), dim3(N, (quads + 255) / 256), dim3(256), 0, 0, p); }));
    line("thread = cell row, one-shot blocks of 512, image-major",
         timed([&] { hipLaunchKernelGGL((k_quads<512, 0>), dim3((quads + 511) / 512, N), dim3(512), 0, 0, p); }));
    line("thread = cell row, one-shot blocks of 1024, image-major",
         timed([&] { hipLaunchKernelGGL((k_quads<1024, 0>), dim3((quads + 1023) / 1024, N), dim3(1024), 0, 0, p); }));
    line("thread = cell row, blocks of 256 x 2 quads per thread, image-major",
         timed([&] { hipLaunchKernelGGL((k_quads_u<256, 2>), dim3((quads + 511) / 512, N), dim3(256), 0, 0, p); }));
    line("thread = cell row, blocks of 256 x 4 quads per thread, image-major",
         timed([&] { hipLaunchKernelGGL((k_quads_u<256, 4>), dim3((quads + 1023) / 1024, N), dim3(256), 0, 0, p); }));
    line("thread = cell row, blocks of 256 x 16 quads per thread, image-major",
         timed([&] { hipLaunchKernelGGL((k_quads_u<256, 16>), dim3((quads + 4095) / 4096, N), dim3(256), 0, 0, p); }));
    for (unsigned g : {256u, 512u, 1024u, 2048u}) {
      char s[160];
      snprintf(s, sizeof s, "thread = cell row, %u long-lived blocks of 256, chip-wide stride over the batch", g);
      line(s, timed([&] { hipLaunchKernelGGL((k_quads_stride<256>), dim3(g), dim3(256), 0, 0, p); }));
      snprintf(s, sizeof s, "thread = cell row, %u long-lived blocks of 1024, chip-wide stride over the batch", g / 2);
      line(s, timed([&] { hipLaunchKernelGGL((k_quads_stride<1024>), dim3(g / 2), dim3(1024), 0, 0, p); }));
    }
  }
  return 0;
}
