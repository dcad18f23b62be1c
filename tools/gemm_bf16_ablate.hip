// GPU tool: where does the time of the LDS-DMA bf16 GEMM go?  Compiles tfkaldi_amd/csrc/gemm_bf16.hip with pieces of
// the K loop removed (-DTFKB_ABL: 1 no MFMAs, 2 no LDS-DMA pieces, 4 no fragment reads) and times a shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTFKB_ABL=<n> -I tfkaldi_amd/csrc tools/gemm_bf16_ablate.hip -o tools/bin/abl<n>
//   tools/bin/abl<n> <layout> <M> <N> <K> <cfg>
#include "../tfkaldi_amd/csrc/gemm_bf16.hip"

#include <stdio.h>
#include <string.h>
#include <vector>

int main(int argc, char** argv) {
  if (argc < 6) return 1;
  const int layout = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]), cfg = atoi(argv[5]);
  auto p8 = [](int n) { return (n + 7) & ~7; };
  const int a_rows = layout == 2 ? K : M, a_cols = layout == 2 ? M : K;
  const int b_rows = layout == 1 ? N : K, b_cols = layout == 1 ? K : N;
  const int lda = p8(a_cols), ldb = p8(b_cols), ldc = (N + 3) & ~3;
  std::vector<uint16_t> ha((size_t)a_rows * lda), hb((size_t)b_rows * ldb);
  unsigned s = 12345u;
  auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (uint16_t)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 8) & 0x8000u)); };
  const bool zeros = getenv("TFK_ABL_DATA") && !strcmp(getenv("TFK_ABL_DATA"), "zero");  // (operands that switch no bits)
  for (auto& x : ha) x = zeros ? 0 : rnd();
  for (auto& x : hb) x = zeros ? 0 : rnd();
  uint16_t *dA, *dB;
  float* dC;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, (size_t)M * ldc * 4);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  tfk::GemmArgsB g = {};
  g.A = dA; g.B = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  tfk::gemm_bf16_force_config(cfg);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) tfk::gemm_bf16((tfk::GemmLayout)layout, g, 0);
  const int iters = getenv("TFK_ABL_ITERS") ? atoi(getenv("TFK_ABL_ITERS")) : 30;  // (thousands: long enough to sample power)
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) tfk::gemm_bf16((tfk::GemmLayout)layout, g, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("ABL=%d layout %d %dx%dx%d cfg %d: %7.1f us  %7.1f TF-equivalent  %6.1f B/clk/CU fill-equivalent at 2.1 GHz\n", TFKB_ABL, layout,
         M, N, K, cfg, ms * 1e3, 2.0 * M * N * K / ms / 1e9,
         cfg == 5 ? ((double)((M + 255) / 256) * ((N + 127) / 128) * ((K + 63) / 64) * 49152.0) / (ms * 1e-3 * 2.1e9 * 256) : 0.0);
  return 0;
}
