// multi_gpu_batch.cpp -- the sharded form of the path driven from C++ alone (BASELINE.json north_star: "C++ host code calls hand-
// written CDNA4 HIP kernels through a thin C-ABI ... image batches shard one-image-per-GPU ... with RCCL over xGMI only for the tiny
// metadata all-reduce"): one PROCESS per GPU, each with its own slice of a batch of 4K pairs resident in its HBM; per step every
// rank runs generateGainMap + applyGainMap on its slice (include/uhdr_hip.h) and the ranks exchange the batch-wide content
// min / max boost (include/uhdr_hip_comm.h) -- 8 bytes, on a stream of its own so that it overlaps the apply kernels.
//
//   ./multi_gpu_batch [--gpus N] [--frames F per GPU] [--steps K] [--warmup W]        (defaults 1, 16, 10, 3)
//
// The parent forks the ranks BEFORE anything touches a GPU (nothing is exec'ed afterwards), hands rank 0's RCCL id to the others
// through shared memory, and prints one JSON line from what rank 0 reports.  bench.py is the measured benchmark (and the one the
// driver runs); this program is the same step without Python or PyTorch in the process, and tests/test_gpu_comm.py runs it with
// one rank on the one GPU a test box has.  Build: see tests/test_gpu_comm.py (hipcc, -luhdr_hip -luhdr_hip_comm).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <vector>

#include "uhdr_hip.h"
#include "uhdr_hip_comm.h"

namespace {
constexpr size_t W = 3840, H = 2160;

struct Shared {   // parent <-> ranks (anonymous shared mapping)
  volatile int id_ready;
  unsigned char id[UHDR_HIP_COMM_ID_BYTES];
  volatile int arrived[2];   // two barriers' worth of counters (sense by index)
  volatile int failed;       // a rank gave up: the others stop waiting for it
  double seconds[64];
  float minmax[64][2];
  int status[64];
};

// every wait is bounded (60 s) and ends when a rank has failed: a rank that dies must not leave the others spinning
bool wait_until(Shared* sh, volatile int* word, int at_least) {
  for (int spins = 0; *word < at_least; ++spins) {
    if (sh->failed || spins > 1200000) return false;
    usleep(50);
  }
  return true;
}
bool barrier(Shared* sh, int which, int world) {
  __sync_fetch_and_add(&sh->arrived[which], 1);
  return wait_until(sh, &sh->arrived[which], world);
}

#define CHECK(expr) do { const int _rc = (expr); if (_rc != 0) { fprintf(stderr, "rank %d: %s -> %d (%s)\n", rank, #expr, _rc, uhdr_hip_last_error()); sh->failed = 1; return 10; } } while (0)
#define HIPCHECK(expr) do { const hipError_t _e = (expr); if (_e != hipSuccess) { fprintf(stderr, "rank %d: %s -> %s\n", rank, #expr, hipGetErrorString(_e)); sh->failed = 1; return 11; } } while (0)
#define WAITCHECK(expr) do { if (!(expr)) { fprintf(stderr, "rank %d: gave up waiting in %s\n", rank, #expr); sh->failed = 1; return 12; } } while (0)

int run_rank(Shared* sh, int rank, int world, int frames, int steps, int warmup) {
  CHECK(uhdr_hip_init(rank));   // device = rank
  if (rank == 0) {
    CHECK(uhdr_hip_comm_get_unique_id(sh->id));
    __sync_synchronize();
    sh->id_ready = 1;
  }
  WAITCHECK(wait_until(sh, &sh->id_ready, 1));
  uhdr_hip_comm* comm = nullptr;
  CHECK(uhdr_hip_comm_init(sh->id, world, rank, rank, &comm));

  hipStream_t s, side;
  HIPCHECK(hipStreamCreate(&s));
  HIPCHECK(hipStreamCreate(&side));
  hipEvent_t generated, reduced;
  HIPCHECK(hipEventCreateWithFlags(&generated, hipEventDisableTiming));
  HIPCHECK(hipEventCreateWithFlags(&reduced, hipEventDisableTiming));

  // this rank's slice: global image index = rank * frames + i, seed 1234 + index
  const size_t p010_bytes = W * H * 3, yuv_bytes = W * H * 3 / 2, map_bytes = (W / 4) * (H / 4), out_bytes = W * H * 4;
  std::vector<uhdr_hip_image_t> yi(frames), pi(frames), mi(frames), oi(frames);
  // What stays resident -- four arenas, frames back to back -- comes from a placement pool (include/uhdr_hip.h): drawn
  // from a wide stretch of the card, every arena spread evenly over it, the rest handed back.  Plain hipMalloc serves as well (6-10 %
  // slower on a card whose memory is free) and is what a runtime without the virtual-memory calls gets.
  const size_t sizes[4] = {p010_bytes, yuv_bytes, (map_bytes + 255) / 256 * 256, out_bytes};
  char* arena[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t free_b = 0, total_b = 0, need = 0;
  HIPCHECK(hipMemGetInfo(&free_b, &total_b));
  for (size_t sz : sizes) need += ((sz * (size_t)frames + ((size_t)16 << 20) - 1) >> 24) << 24;
  size_t span = free_b / 10 * 7;
  if (span > ((size_t)200 << 30)) span = (size_t)200 << 30;
  uhdr_hip_mem_pool_t* pool = nullptr;
  bool pooled = uhdr_hip_mem_pool_create(rank, span > need ? span : need, 0, &pool) == UHDR_HIP_NO_ERROR;
  for (int k = 0; k < 4 && pooled; ++k) pooled = uhdr_hip_mem_pool_alloc(pool, sizes[k] * (size_t)frames, reinterpret_cast<void**>(&arena[k])) == UHDR_HIP_NO_ERROR;
  if (pooled) {
    CHECK(uhdr_hip_mem_pool_trim(pool));
  } else {
    if (pool) CHECK(uhdr_hip_mem_pool_destroy(pool));
    for (int k = 0; k < 4; ++k) HIPCHECK(hipMalloc(&arena[k], sizes[k] * (size_t)frames));
  }
  for (int i = 0; i < frames; ++i) {
    char *p = arena[0] + sizes[0] * i, *y = arena[1] + sizes[1] * i, *m = arena[2] + sizes[2] * i, *o = arena[3] + sizes[3] * i;
    CHECK(uhdr_hip_synth_lcg_frame(W, H, 1234u + (unsigned)(rank * frames + i), p, y, s));
    yi[i] = uhdr_hip_image_t{y, W, H, UHDR_HIP_CG_BT709, y + W * H, W, W / 2, UHDR_HIP_PIX_FMT_YUV420};
    pi[i] = uhdr_hip_image_t{p, W, H, UHDR_HIP_CG_BT2100, p + W * H * 2, W, W, UHDR_HIP_PIX_FMT_P010};
    mi[i] = uhdr_hip_image_t{m, 0, 0, UHDR_HIP_CG_UNSPECIFIED, nullptr, 0, 0, UHDR_HIP_PIX_FMT_UNSPECIFIED};
    oi[i] = uhdr_hip_image_t{o, 0, 0, UHDR_HIP_CG_UNSPECIFIED, nullptr, 0, 0, UHDR_HIP_PIX_FMT_UNSPECIFIED};
  }
  float *per_image, *batch_mm;
  HIPCHECK(hipMalloc(&per_image, sizeof(float) * 2 * frames));
  HIPCHECK(hipMalloc(&batch_mm, sizeof(float) * 2));
  uhdr_hip_metadata_t md;
  memset(&md, 0, sizeof md);

  auto step = [&]() -> int {
    for (int lo = 0; lo < frames; lo += 64) {   // (a launch takes up to 64 images)
      const int n = frames - lo < 64 ? frames - lo : 64;
      CHECK(uhdr_hip_generate_gainmap_batch(n, &yi[lo], &pi[lo], UHDR_HIP_TF_HLG, &md, &mi[lo], 0, per_image + 2 * lo, s));
    }
    // the exchange waits for generate, runs beside apply, and the step ends when both have
    HIPCHECK(hipEventRecord(generated, s));
    HIPCHECK(hipStreamWaitEvent(side, generated, 0));
    CHECK(uhdr_hip_comm_allreduce_minmax(comm, per_image, frames, batch_mm, side));
    HIPCHECK(hipEventRecord(reduced, side));
    for (int lo = 0; lo < frames; lo += 64) {
      const int n = frames - lo < 64 ? frames - lo : 64;
      CHECK(uhdr_hip_apply_gainmap_batch(n, &yi[lo], &mi[lo], &md, UHDR_HIP_OUTPUT_HDR_HLG, 3.4028235e38f, &oi[lo], UHDR_HIP_APPLY_FAST, s));
    }
    HIPCHECK(hipStreamWaitEvent(s, reduced, 0));
    return 0;
  };

  for (int k = 0; k < warmup; ++k) if (int rc = step()) return rc;
  HIPCHECK(hipDeviceSynchronize());
  WAITCHECK(barrier(sh, 0, world));
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < steps; ++k) if (int rc = step()) return rc;
  HIPCHECK(hipDeviceSynchronize());
  WAITCHECK(barrier(sh, 1, world));
  sh->seconds[rank] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  HIPCHECK(hipMemcpy((void*)sh->minmax[rank], batch_mm, sizeof(float) * 2, hipMemcpyDeviceToHost));
  CHECK(uhdr_hip_comm_destroy(comm));
  return 0;
}
}  // namespace

int main(int argc, char** argv) {
  int world = 1, frames = 16, steps = 10, warmup = 3;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (!strcmp(argv[i], "--gpus")) world = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--frames")) frames = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--steps")) steps = atoi(argv[i + 1]);
    else if (!strcmp(argv[i], "--warmup")) warmup = atoi(argv[i + 1]);
  }
  if (world < 1 || world > 64 || frames < 1 || steps < 1 || warmup < 0) { fprintf(stderr, "bad arguments\n"); return 2; }
  Shared* sh = static_cast<Shared*>(mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0));
  if (sh == MAP_FAILED) return 3;
  memset(sh, 0, sizeof *sh);
  std::vector<pid_t> kids;
  for (int r = 0; r < world; ++r) {   // (fork before any HIP call: the parent never initialises a GPU)
    const pid_t p = fork();
    if (p == 0) _exit(run_rank(sh, r, world, frames, steps, warmup));
    kids.push_back(p);
  }
  int bad = 0;
  for (pid_t p : kids) { int st = 0; waitpid(p, &st, 0); if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) bad = 1; }
  if (bad) { fprintf(stderr, "a rank failed\n"); return 1; }
  double t = 0;
  for (int r = 0; r < world; ++r) t = sh->seconds[r] > t ? sh->seconds[r] : t;   // the slowest rank
  int agree = 1;
  for (int r = 1; r < world; ++r) agree = agree && sh->minmax[r][0] == sh->minmax[0][0] && sh->minmax[r][1] == sh->minmax[0][1];
  printf("{\"program\": \"examples/multi_gpu_batch.cpp\", \"n_gpus\": %d, \"frames_per_gpu\": %d, \"steps\": %d, \"ms_per_step\": %.4f, "
         "\"value\": %.1f, \"unit\": \"MPix/s\", \"content_minmax\": [%.9g, %.9g], \"ranks_agree\": %s}\n",
         world, frames, steps, t / steps * 1e3, (double)world * frames * steps * (W * H / 1e6) / t, sh->minmax[0][0], sh->minmax[0][1],
         agree ? "true" : "false");
  return agree ? 0 : 4;
}
