// A device-memory arena whose PHYSICAL backing is chosen piece by piece (HIP virtual-memory management): `size` bytes of contiguous
// virtual addresses backed by separately created chunks of `chunk` bytes, mapped in a shuffled order when seed != 0.
//   hipcc -shared -fPIC -O2 scripts/ab/vmm_arena.cpp -o scripts/ab/libvmm_arena.so
// scripts/time_placement_slab.py vmm uses it to ask what placement does to the step (profiles/r04_placement.txt).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>
extern "C" int vmm_granularity(size_t* out) {
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = 0;
  return (int)hipMemGetAllocationGranularity(out, &p, hipMemAllocationGranularityMinimum);
}
extern "C" int vmm_alloc(size_t size, size_t chunk, unsigned seed, void** out) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return -1;
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = dev;
  const size_t n = (size + chunk - 1) / chunk;
  void* va = nullptr;
  hipError_t e = hipMemAddressReserve(&va, n * chunk, chunk < ((size_t)2 << 20) ? ((size_t)2 << 20) : 0, nullptr, 0);
  if (e != hipSuccess) { fprintf(stderr, "hipMemAddressReserve: %s\n", hipGetErrorString(e)); return -2; }
  std::vector<hipMemGenericAllocationHandle_t> h(n);
  for (size_t i = 0; i < n; ++i) {
    e = hipMemCreate(&h[i], chunk, &p, 0);
    if (e != hipSuccess) { fprintf(stderr, "hipMemCreate(%zu): %s\n", i, hipGetErrorString(e)); return -3; }
  }
  std::vector<size_t> order(n);
  for (size_t i = 0; i < n; ++i) order[i] = i;
  if (seed != 0) { std::mt19937 g(seed); std::shuffle(order.begin(), order.end(), g); }
  for (size_t i = 0; i < n; ++i) {
    e = hipMemMap(static_cast<char*>(va) + i * chunk, chunk, 0, h[order[i]], 0);
    if (e != hipSuccess) { fprintf(stderr, "hipMemMap(%zu): %s\n", i, hipGetErrorString(e)); return -4; }
  }
  hipMemAccessDesc a = {};
  a.location.type = hipMemLocationTypeDevice;
  a.location.id = dev;
  a.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(va, n * chunk, &a, 1);
  if (e != hipSuccess) { fprintf(stderr, "hipMemSetAccess: %s\n", hipGetErrorString(e)); return -5; }
  *out = va;
  return 0;
}

// The same physical chunks under several mappings: vmm_create makes the chunks and reserves the addresses, vmm_map maps them in the
// order seed gives (0: as created), replacing the mapping before it.
struct VmmCtx { void* va; size_t n, chunk; std::vector<hipMemGenericAllocationHandle_t> h; bool mapped; int dev; };
extern "C" void* vmm_create(size_t size, size_t chunk) {
  VmmCtx* c = new VmmCtx();
  c->chunk = chunk; c->n = (size + chunk - 1) / chunk; c->mapped = false; c->va = nullptr;
  if (hipGetDevice(&c->dev) != hipSuccess) return nullptr;
  hipMemAllocationProp p = {};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = c->dev;
  if (hipMemAddressReserve(&c->va, c->n * chunk, 0, nullptr, 0) != hipSuccess) return nullptr;
  c->h.resize(c->n);
  for (size_t i = 0; i < c->n; ++i)
    if (hipMemCreate(&c->h[i], chunk, &p, 0) != hipSuccess) return nullptr;
  return c;
}
extern "C" void* vmm_map(void* ctx, unsigned seed) {
  VmmCtx* c = static_cast<VmmCtx*>(ctx);
  if (c->mapped && hipMemUnmap(c->va, c->n * c->chunk) != hipSuccess) return nullptr;
  std::vector<size_t> order(c->n);
  for (size_t i = 0; i < c->n; ++i) order[i] = i;
  if (seed != 0) { std::mt19937 g(seed); std::shuffle(order.begin(), order.end(), g); }
  for (size_t i = 0; i < c->n; ++i)
    if (hipMemMap(static_cast<char*>(c->va) + i * c->chunk, c->chunk, 0, c->h[order[i]], 0) != hipSuccess) return nullptr;
  hipMemAccessDesc a = {};
  a.location.type = hipMemLocationTypeDevice;
  a.location.id = c->dev;
  a.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(c->va, c->n * c->chunk, &a, 1) != hipSuccess) return nullptr;
  c->mapped = true;
  return c->va;
}

// A pool of physical chunks taken from the device one after another (so that, on a device whose memory is mostly free, they walk
// through it), and arenas built from chosen members: vmm_pool_map maps chunks first, first + stride, first + 2 stride, ... (count of them).
struct VmmPool { size_t n, chunk; std::vector<hipMemGenericAllocationHandle_t> h; int dev; };
extern "C" void* vmm_pool_create(size_t total, size_t chunk) {
  VmmPool* p = new VmmPool();
  p->chunk = chunk; p->n = total / chunk;
  if (hipGetDevice(&p->dev) != hipSuccess) return nullptr;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = p->dev;
  p->h.resize(p->n);
  for (size_t i = 0; i < p->n; ++i)
    if (hipMemCreate(&p->h[i], chunk, &prop, 0) != hipSuccess) { fprintf(stderr, "hipMemCreate failed at chunk %zu\n", i); p->n = i; p->h.resize(i); break; }
  return p;
}
extern "C" size_t vmm_pool_chunks(void* pool) { return static_cast<VmmPool*>(pool)->n; }
extern "C" void* vmm_pool_map(void* pool, size_t first, size_t stride, size_t count) {
  VmmPool* p = static_cast<VmmPool*>(pool);
  void* va = nullptr;
  if (hipMemAddressReserve(&va, count * p->chunk, 0, nullptr, 0) != hipSuccess) return nullptr;
  for (size_t i = 0; i < count; ++i) {
    const size_t k = (first + i * stride) % p->n;
    if (hipMemMap(static_cast<char*>(va) + i * p->chunk, p->chunk, 0, p->h[k], 0) != hipSuccess) return nullptr;
  }
  hipMemAccessDesc a = {};
  a.location.type = hipMemLocationTypeDevice;
  a.location.id = p->dev;
  a.flags = hipMemAccessFlagsProtReadWrite;
  if (hipMemSetAccess(va, count * p->chunk, &a, 1) != hipSuccess) return nullptr;
  return va;
}
