// GPU tool (round 6): which CUs does a hipExtStreamCreateWithCUMask mask select on this 8-XCD part?
// Each block of a launch on the masked stream records (XCC_ID, SE_ID, CU_ID) of where it ran; the host prints, per
// mask, the number of distinct CUs seen on every XCD.  Beside tools/adam_cu_mask.py: "bits 0..127" slows the optimiser
// by 1.58x, "every other bit" not at all -- this says what each of the two masks is.
// build: hipcc --offload-arch=gfx950 -O2 -o /tmp/cu_mask_map tools/ubench/cu_mask_map.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <set>
#include <vector>
__global__ void where(unsigned* out) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // spin a little so that blocks spread over every CU the mask allows
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) {}
  if (threadIdx.x == 0) out[blockIdx.x] = (xcc << 24) | (hwid & 0xffffff);
}
static void run(const char* label, const std::vector<unsigned>& mask) {
  hipStream_t s;
  if (mask.empty()) hipStreamCreate(&s);
  else if (hipExtStreamCreateWithCUMask(&s, (unsigned)mask.size(), mask.data()) != hipSuccess) { printf("%s: mask refused\n", label); return; }
  const int nb = 8192;
  unsigned* d;
  hipMalloc(&d, nb * 4);
  hipLaunchKernelGGL(where, dim3(nb), dim3(64), 0, s, d);
  hipStreamSynchronize(s);
  std::vector<unsigned> h(nb);
  hipMemcpy(h.data(), d, nb * 4, hipMemcpyDeviceToHost);
  std::set<unsigned> cus[8];
  for (unsigned v : h) {
    const unsigned xcc = v >> 24, hw = v & 0xffffff;
    // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
    cus[xcc & 7].insert((hw >> 8) & 0xff);
  }
  int total = 0;
  printf("%-40s CUs per XCD:", label);
  for (int x = 0; x < 8; ++x) { printf(" %2zu", cus[x].size()); total += (int)cus[x].size(); }
  printf("   total %d\n", total);
  hipFree(d);
  hipStreamDestroy(s);
}
int main() {
  run("plain stream", {});
  run("256 bits", std::vector<unsigned>(8, 0xffffffffu));
  run("bits 0..127", {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu, 0, 0, 0, 0});
  run("every other bit", std::vector<unsigned>(8, 0x55555555u));
  run("bits 0..63", {0xffffffffu, 0xffffffffu, 0, 0, 0, 0, 0, 0});
  run("bits 0..7 of every word", std::vector<unsigned>(8, 0x000000ffu));
  return 0;
}
