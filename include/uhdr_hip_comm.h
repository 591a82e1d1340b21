/* uhdr_hip_comm.h -- the multi-GPU side of the path, for a C or C++ host (BASELINE.json north_star: "image batches
 * shard one-image-per-GPU ... with RCCL over xGMI only for the tiny metadata all-reduce").
 *
 * One process per GPU.  Images never leave the GPU they were given to (uhdr_hip.h: every entry point works on the caller's device
 * pointers), so the path has exactly one exchange: the batch-wide content min / max boost -- two floats -- that the metadata of a
 * batch encoded with one common boost range needs.  The reference has no counterpart (it writes constants: lib/src/ultrahdr.cpp:
 * 250-257); uhdr_hip_generate_gainmap[_batch]'s `minmax` output (uhdr_hip.h) is what is reduced here.
 *
 * This library (libuhdr_hip_comm.so) is separate from libuhdr_hip.so so that the pixel path does not depend on RCCL.  It links
 * librccl.so.1; in a process that has loaded another copy under that name (PyTorch ships one) the dynamic loader hands it that one.
 *
 * Bootstrap, as with any NCCL / RCCL program: rank 0 calls uhdr_hip_comm_get_unique_id and the host side carries the 128 bytes to the
 * other ranks by whatever it has (a file, a socket, MPI, torch.distributed's store); every rank then calls uhdr_hip_comm_init, which
 * is collective.  Return values are uhdr_hip.h's status codes: UHDR_HIP_NO_ERROR (0), UHDR_HIP_ERROR_BAD_PTR for a null or
 * out-of-range argument, UHDR_HIP_ERROR_INSUFFICIENT_RESOURCE, UHDR_HIP_UNKNOWN_ERROR when RCCL or HIP reports a failure. */
#ifndef UHDR_HIP_COMM_H
#define UHDR_HIP_COMM_H

#ifdef __cplusplus
extern "C" {
#endif

#define UHDR_HIP_COMM_ID_BYTES 128

typedef struct uhdr_hip_comm uhdr_hip_comm;

/* id: UHDR_HIP_COMM_ID_BYTES bytes, written (ncclGetUniqueId).  Call on ONE rank. */
int uhdr_hip_comm_get_unique_id(void* id);

/* Collective over all `world` ranks: joins the communicator the id names, on HIP device `device` (current inside every call on
 * the communicator, the caller's current device again when the call returns).
 * world == 1 is valid (a single-GPU run of a multi-GPU program). */
int uhdr_hip_comm_init(const void* id, int world, int rank, int device, uhdr_hip_comm** comm);

int uhdr_hip_comm_world(const uhdr_hip_comm* comm, int* world, int* rank);

/* The exchange.  per_image_minmax: device pointer to 2 * images floats (min_0, max_0, min_1, ...), this rank's images as
 * uhdr_hip_generate_gainmap_batch wrote them; images may be 0 (a rank without images contributes nothing).  batch_minmax: device
 * pointer to 2 floats, written: (min over all images of all ranks, max over them); (+inf, -inf) when no rank had an image.
 * Asynchronous on `stream` (a hipStream_t, NULL = the null stream): a fold kernel, one ncclAllReduce of 2 floats, a store; the
 * caller's next kernels on other streams overlap it.  Collective: every rank calls it once per batch, in the same order.
 * batch_minmax itself is the reduction buffer (it holds intermediate values until the call's work on `stream` is done; it must
 * not alias per_image_minmax), the communicator keeps no scratch: exchanges in flight on different streams, each with its own
 * batch_minmax, do not meet -- as long as every rank issues them in one order, which RCCL requires of a communicator. */
int uhdr_hip_comm_allreduce_minmax(uhdr_hip_comm* comm, const float* per_image_minmax, int images, float* batch_minmax, void* stream);

/* Collective (ncclCommDestroy), frees the handle. */
int uhdr_hip_comm_destroy(uhdr_hip_comm* comm);

#ifdef __cplusplus
}
#endif
#endif
