// s_memtime ticks per microsecond: a kernel that spins for N ticks, timed with HIP events.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_spin(unsigned long long n, unsigned long long* out) {
  const unsigned long long t0 = __builtin_readcyclecounter();
  unsigned long long t = t0;
  while (t - t0 < n) t = __builtin_readcyclecounter();
  if (threadIdx.x == 0 && blockIdx.x == 0) *out = t - t0;
}
int main() {
  unsigned long long* d; (void)hipMalloc(&d, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int grid : {1, 256}) for (unsigned long long n : {1000000ull, 10000000ull}) {
    k_spin<<<grid, 64>>>(n, d);
    (void)hipEventRecord(e0); k_spin<<<grid, 64>>>(n, d); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("grid %d: %llu ticks in %.1f us -> %.1f ticks/us\n", grid, n, ms * 1000, n / (ms * 1000));
  }
  return 0;
}
