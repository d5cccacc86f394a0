// code_size_probe.cpp -- stand-alone repro attempt for the "device code size cliff" (profiles/r04_tuning.md): with 0.72 MB of device
// code in the library, every small synchronous null-stream operation of a process that had copied through IPC mappings took
// 14 ms.  Which of {size of the code object, number of kernels in it, the IPC copy} flips it?  No library involved.
//
//   hipcc --offload-arch=gfx950 -O1 -DNK=<filler kernels> -DBODY=<statements per filler kernel> code_size_probe.cpp -o probe_NK_BODY
//   RANK=r WORLD_SIZE=n JOB=tag ./probe_NK_BODY [ipc|noipc]        (n processes sharing the GPU; 1 is allowed)
//
// Every process: launches ONE filler kernel (so the code object is loaded), times 40 x { tiny kernel + 8-byte synchronous
// hipMemcpy D2H } and 40 x hipDeviceSynchronize-after-tiny-kernel; then (ipc) exports a 64 MiB buffer, opens the next rank's,
// copies 64 MiB into it with hipMemcpy and with a kernel; then times the same loops again.  One line per process.
#include <hip/hip_runtime.h>
#include <unistd.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <utility>

#ifndef NK
#define NK 16
#endif
#ifndef BODY
#define BODY 64
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int I>
__global__ void filler(float* p) {
  float a = p[threadIdx.x], b = p[threadIdx.x + 64];
#pragma unroll
  for (int k = 0; k < BODY; ++k) {  // distinct constants per (I, k): nothing folds across kernels
    a = a * (1.0f + 0.001f * (I * 131 + k)) + b;
    b = b * (0.5f + 0.003f * (I * 17 + k * 3)) - a;
  }
  p[threadIdx.x] = a + b;
}
typedef void (*kern_t)(float*);
template <int... Is>
static void table(kern_t* t, std::integer_sequence<int, Is...>) {
  ((t[Is] = filler<Is>), ...);
}
__global__ void tiny(unsigned long long* p) { if (threadIdx.x == 0) p[0] += 1; }
__global__ void copyk(const uint4* s, uint4* d, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) d[i] = s[i];
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void measure(unsigned long long* d, double* memcpy_ms, double* sync_ms) {
  unsigned long long h;
  double t0 = now();
  for (int i = 0; i < 40; ++i) { tiny<<<1, 64>>>(d); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); }
  *memcpy_ms = (now() - t0) * 1e3 / 40;
  t0 = now();
  for (int i = 0; i < 40; ++i) { tiny<<<1, 64>>>(d); CK(hipDeviceSynchronize()); }
  *sync_ms = (now() - t0) * 1e3 / 40;
}

int main(int argc, char** argv) {
  const bool ipc = argc > 1 && !strcmp(argv[1], "ipc");
  const int rank = getenv("RANK") ? atoi(getenv("RANK")) : 0, n = getenv("WORLD_SIZE") ? atoi(getenv("WORLD_SIZE")) : 1;
  const std::string job = std::string("/dev/shm/csp_") + (getenv("JOB") ? getenv("JOB") : "job");
  static kern_t t[NK];
  table(t, std::make_integer_sequence<int, NK>{});
  float* f;
  unsigned long long* d;
  CK(hipMalloc(&f, 4096));
  CK(hipMalloc(&d, 8));
  CK(hipMemset(f, 0, 4096));
  CK(hipMemset(d, 0, 8));
  hipLaunchKernelGGL(t[rank % NK], dim3(1), dim3(64), 0, 0, f);
  CK(hipDeviceSynchronize());
  double m0, s0, m1 = -1, s1 = -1;
  measure(d, &m0, &s0);
  if (ipc) {
    const size_t bytes = 64u << 20;
    char *mine, *local;
    CK(hipMalloc(&mine, bytes));
    CK(hipMalloc(&local, bytes));
    hipIpcMemHandle_t h;
    CK(hipIpcGetMemHandle(&h, mine));
    { FILE* fp = fopen((job + "_" + std::to_string(rank) + ".tmp").c_str(), "wb"); fwrite(&h, sizeof(h), 1, fp); fclose(fp);
      rename((job + "_" + std::to_string(rank) + ".tmp").c_str(), (job + "_" + std::to_string(rank)).c_str()); }
    const int peer = (rank + 1) % n;
    hipIpcMemHandle_t ph;
    for (;;) { FILE* fp = fopen((job + "_" + std::to_string(peer)).c_str(), "rb"); if (fp) { size_t got = fread(&ph, sizeof(ph), 1, fp); fclose(fp); if (got == 1) break; } usleep(1000); }
    void* mapped = mine;
    if (n > 1) CK(hipIpcOpenMemHandle(&mapped, ph, hipIpcMemLazyEnablePeerAccess));
    CK(hipMemcpy(mapped, local, bytes, hipMemcpyDeviceToDevice));
    copyk<<<1024, 256>>>((const uint4*)local, (uint4*)mapped, bytes / 16);
    CK(hipDeviceSynchronize());
    measure(d, &m1, &s1);
  }
  printf("NK=%d BODY=%d rank %d of %d %s: tiny kernel + 8-byte hipMemcpy %.3f ms, + hipDeviceSynchronize %.3f ms | after the IPC copy: %.3f ms, %.3f ms\n",
         NK, BODY, rank, n, ipc ? "ipc" : "noipc", m0, s0, m1, s1);
  fflush(stdout);
  usleep(300000);  // keep the mapping's owner alive until the peers are done
  return 0;
}
