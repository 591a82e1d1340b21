// uhdr_comm.hip -- include/uhdr_hip_comm.h: the path's one exchange between GPUs, over RCCL.  (libuhdr_hip_comm.so; nothing of the
// pixel path is in here, and nothing in libuhdr_hip.so knows about ranks.)
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>
#include <new>

#include "../../include/uhdr_hip_comm.h"

static_assert(sizeof(ncclUniqueId) == UHDR_HIP_COMM_ID_BYTES, "the id travels as 128 opaque bytes");

namespace {
// status codes of uhdr_hip.h (status_t, ultrahdr.h:91-120)
constexpr int kOk = 0, kUnknown = -1, kInvalidParam = -10001 /* ERROR_BAD_PTR: a null or out-of-range argument */, kInsufficientResource = -20009;

// this rank's (min, max) pairs -> (-min, max) of all of them: one all-reduce(MAX) serves both.  One block; a wave64 butterfly per
// wave, the waves' results through LDS.
__global__ void __launch_bounds__(256) k_fold_minmax(const float* __restrict__ per_image, int images, float* __restrict__ red) {
  float nmin = -__builtin_inff(), vmax = -__builtin_inff();   // (-min, max) of nothing
  for (int i = (int)threadIdx.x; i < images; i += 256) {
    nmin = fmaxf(nmin, -per_image[2 * i]);
    vmax = fmaxf(vmax, per_image[2 * i + 1]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    nmin = fmaxf(nmin, __shfl_xor(nmin, off, 64));
    vmax = fmaxf(vmax, __shfl_xor(vmax, off, 64));
  }
  __shared__ float s[2][4];
  if ((threadIdx.x & 63u) == 0u) { s[0][threadIdx.x >> 6] = nmin; s[1][threadIdx.x >> 6] = vmax; }
  __syncthreads();
  if (threadIdx.x == 0u) {
    red[0] = fmaxf(fmaxf(s[0][0], s[0][1]), fmaxf(s[0][2], s[0][3]));
    red[1] = fmaxf(fmaxf(s[1][0], s[1][1]), fmaxf(s[1][2], s[1][3]));
  }
}
// the communicator's device is current inside a call and the caller's again afterwards
struct DeviceGuard {
  int before = -1;
  bool ok = false;
  explicit DeviceGuard(int device) {
    if (hipGetDevice(&before) != hipSuccess) before = -1;
    ok = hipSetDevice(device) == hipSuccess;
  }
  ~DeviceGuard() { if (before >= 0) (void)hipSetDevice(before); }
};

__global__ void k_unfold_minmax(float* red) { red[0] = -red[0]; }   // (-min, max) -> (min, max), in place
}  // namespace

struct uhdr_hip_comm {
  ncclComm_t comm;
  int world, rank, device;
};

extern "C" {

int uhdr_hip_comm_get_unique_id(void* id) {
  if (id == nullptr) return kInvalidParam;
  ncclUniqueId u;
  if (ncclGetUniqueId(&u) != ncclSuccess) return kUnknown;
  memcpy(id, &u, sizeof u);
  return kOk;
}

int uhdr_hip_comm_init(const void* id, int world, int rank, int device, uhdr_hip_comm** out) {
  if (id == nullptr || out == nullptr || world < 1 || rank < 0 || rank >= world || device < 0) return kInvalidParam;
  *out = nullptr;
  DeviceGuard on(device);
  if (!on.ok) return kInvalidParam;
  uhdr_hip_comm* c = new (std::nothrow) uhdr_hip_comm();
  if (c == nullptr) return kInsufficientResource;
  c->world = world; c->rank = rank; c->device = device; c->comm = nullptr;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  if (ncclCommInitRank(&c->comm, world, u, rank) != ncclSuccess) { delete c; return kUnknown; }
  *out = c;
  return kOk;
}

int uhdr_hip_comm_world(const uhdr_hip_comm* c, int* world, int* rank) {
  if (c == nullptr) return kInvalidParam;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  return kOk;
}

int uhdr_hip_comm_allreduce_minmax(uhdr_hip_comm* c, const float* per_image, int images, float* batch_minmax, void* stream) {
  if (c == nullptr || batch_minmax == nullptr || images < 0 || (images > 0 && per_image == nullptr)) return kInvalidParam;
  if (images > 0 && batch_minmax + 2 > per_image && batch_minmax < per_image + 2 * (size_t)images) return kInvalidParam;   // (the result may not lie inside the input)
  hipStream_t s = static_cast<hipStream_t>(stream);
  DeviceGuard on(c->device);
  if (!on.ok) return kUnknown;
  // the caller's two floats are the reduction buffer all the way -- (-min, max) folded into them, reduced in place, the sign put
  // back in place --, so the communicator holds no state a second exchange could meet: calls on different streams with different
  // batch_minmax are independent (RCCL itself wants a communicator's collectives issued in one order on every rank)
  hipLaunchKernelGGL(k_fold_minmax, dim3(1), dim3(256), 0, s, per_image, images, batch_minmax);
  if (hipGetLastError() != hipSuccess) return kUnknown;
  if (ncclAllReduce(batch_minmax, batch_minmax, 2, ncclFloat, ncclMax, c->comm, s) != ncclSuccess) return kUnknown;
  hipLaunchKernelGGL(k_unfold_minmax, dim3(1), dim3(1), 0, s, batch_minmax);
  return hipGetLastError() == hipSuccess ? kOk : kUnknown;
}

int uhdr_hip_comm_destroy(uhdr_hip_comm* c) {
  if (c == nullptr) return kInvalidParam;
  DeviceGuard on(c->device);
  const bool ok = ncclCommDestroy(c->comm) == ncclSuccess;
  delete c;
  return ok ? kOk : kUnknown;
}

}  // extern "C"
