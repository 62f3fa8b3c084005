// Which CUs does a stream created with hipExtStreamCreateWithCUMask use on MI355X (8 XCDs x 32 CUs)?  Every workgroup records HW_ID / XCC_ID; the host prints, per mask,
// how many distinct (XCC, SE, CU) it saw and how many per XCC.      hipcc --offload-arch=gfx950 -O2 tools/probe/cu_mask_probe.hip -o /tmp/cu_mask_probe && /tmp/cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>
__global__ void k(unsigned* out, int spin) {
  extern __shared__ char smem[];
  unsigned hw = 0, xcc = 0;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  smem[threadIdx.x] = (char)hw;
  long long t0 = clock64();
  while (clock64() - t0 < spin) { __builtin_amdgcn_s_sleep(10); }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc + (unsigned)smem[1] * 0; }
}
static void run(const char* name, const std::vector<uint32_t>& mask) {
  hipStream_t st;
  if (mask.empty()) { if (hipStreamCreate(&st) != hipSuccess) { printf("%s: stream create failed\n", name); return; } }
  else if (hipExtStreamCreateWithCUMask(&st, (uint32_t)mask.size(), mask.data()) != hipSuccess) { printf("%s: hipExtStreamCreateWithCUMask failed\n", name); return; }
  const int n = 2048;
  unsigned* d; hipMalloc(&d, n * 8); hipMemset(d, 0xff, n * 8);
  hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, st);
  k<<<n, 64, 120 * 1024, st>>>(d, 20000);
  hipEventRecord(e1, st);
  hipError_t e = hipStreamSynchronize(st);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned> h(2 * n); hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  std::set<unsigned> cus; std::map<unsigned, std::set<unsigned>> per;
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned id = (se << 5) | (sh << 4) | cu;
    cus.insert((xcc << 8) | id); per[xcc].insert(id);
  }
  printf("%-44s err %d  %.2f ms  distinct CUs %3zu  per XCC:", name, (int)e, ms, cus.size());
  for (auto& kv : per) printf(" %u:%zu", kv.first, kv.second.size());
  printf("\n");
  if (cus.size() <= 40) { printf("    (xcc.se.sh.cu):"); for (unsigned c : cus) printf(" %u.%u.%u.%u", c >> 8, (c >> 5) & 7, (c >> 4) & 1, c & 15); printf("\n"); }
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  run("no mask", {});
  run("bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
  run("bits 0..31", {0xffffffffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 0..7", {0xffu, 0, 0, 0, 0, 0, 0, 0});
  run("bits 64..255", {0, 0, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu});
  run("every 4th bit of 256", {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u});
  run("low 8 bits of every 32", {0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu});
  run("one word only (32 bits), size 1", {0xffffffffu});
  return 0;
}
