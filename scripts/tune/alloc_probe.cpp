// Allocation-size probe (tuning aid): hipMalloc / pageable hipMemcpy H2D+D2H / hipFree cost per size with the
// ROCm runtime the native programs link against.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  const size_t sizes[] = {2095104, 2097152, 4190208, 4194304, 4198400, 8380416, 8388608, 16760832, 16777216, 33554432};
  (void)hipFree(nullptr);
  for (size_t n : sizes) {
    double tm = 0, tc = 0, tf = 0;
    const int reps = 10;
    for (int r = 0; r < reps; ++r) {
      std::vector<char> host(n, 1);
      void* d = nullptr;
      double t0 = now();
      if (hipMalloc(&d, n) != hipSuccess) return 1;
      double t1 = now();
      (void)hipMemcpy(d, host.data(), n, hipMemcpyHostToDevice);
      (void)hipMemcpy(host.data(), d, n, hipMemcpyDeviceToHost);
      double t2 = now();
      (void)hipFree(d);
      double t3 = now();
      tm += t1 - t0;
      tc += t2 - t1;
      tf += t3 - t2;
    }
    printf("%10zu B: malloc %7.3f ms  H2D+D2H %7.3f ms  free %7.3f ms\n", n, tm / reps * 1e3, tc / reps * 1e3, tf / reps * 1e3);
  }
  return 0;
}
