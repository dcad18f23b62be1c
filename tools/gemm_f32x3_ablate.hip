// GPU tool: where does the time of the fp32-emulating GEMM (gemm_bf16x3) go?  Compiles tfkaldi_amd/csrc/gemm_bf16.hip with pieces
// of the K loop removed (-DTFKB_ABL: 1 no MFMAs, 2 no LDS-DMA pieces, 4 no fragment reads; sums combine) and times a shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DTFKB_ABL=<n> -I tfkaldi_amd/csrc tools/gemm_f32x3_ablate.hip -o tools/bin/x3abl<n>
//   tools/bin/x3abl<n> <layout> <M> <N> <K>         (block geometry: env TFK_BF16X3_CFG)
//   env: TFK_ABL_ITERS (launches in the timed loop, default 30; thousands for a SUSTAINED figure -- a burst starts on a cold clock),
//        TFK_ABL_DATA = random | zero | p0 | real (below), TFK_BF16X3_WAVES / TFK_BF16X3_CFG (block form / geometry)
//   tools/bin/x3abl<n> 3 <frames> <d_in> <d_out>    the backward pair of a layer in one launch (gemm_bf16x3_dual):
//                                                   dA[frames, d_in] = dZ . W^T  and  dW[d_in, d_out] = in^T . dZ
#include "../tfkaldi_amd/csrc/gemm_bf16.hip"

#include <stdio.h>
#include <string.h>
#include <vector>

// TFK_ABL_DATA: random (default: every plane ~ +-1, the worst case for switching power) | zero | p0 (planes 1 and 2 zero) |
// real (plane q scaled by 2^-8q: what a split fp32 value looks like)
static void fill_planes(std::vector<uint16_t>& h, unsigned seed) {
  const char* mode = getenv("TFK_ABL_DATA");
  const int m = !mode ? 0 : !strcmp(mode, "zero") ? 1 : !strcmp(mode, "p0") ? 2 : !strcmp(mode, "real") ? 3 : 0;
  unsigned s = seed;
  for (size_t i = 0; i < h.size(); ++i) {
    s = s * 1664525u + 1013904223u;
    const int q = (int)((i % 192) / 64);  // plane of element i of the tiled array
    uint16_t v = (uint16_t)(0x3c00u + ((s >> 9) & 0x3ffu) + ((s >> 8) & 0x8000u));
    if (m == 1 || (m == 2 && q > 0)) v = 0;
    if (m == 3) v = (uint16_t)(v - q * (8u << 7));  // exponent - 8 q
    h[i] = v;
  }
}
static uint16_t* random_planes(size_t rows, int ld, unsigned seed) {
  std::vector<uint16_t> h(tfk::x3::elems(rows, ld));
  fill_planes(h, seed);
  uint16_t* d;
  hipMalloc(&d, h.size() * 2);
  hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  return d;
}

static int dual(int T, int din, int dout) {
  auto p32 = [](int n) { return (n + 31) & ~31; };
  const int ld_in = p32(din), ld_out = p32(dout);
  uint16_t* dz = random_planes(T, ld_out, 1u);   // [T, d_out]
  uint16_t* w = random_planes(din, ld_out, 2u);  // [d_in, d_out]
  uint16_t* in = random_planes(T, ld_in, 3u);    // [T, d_in]
  float *dA, *dW;
  hipMalloc(&dA, (size_t)T * ld_in * 4); hipMalloc(&dW, (size_t)din * ld_out * 4);
  tfk::GemmArgsB a = {}, g = {};
  a.A = dz; a.B = w; a.C = dA; a.M = T; a.N = din; a.K = dout; a.lda = ld_out; a.ldb = ld_out; a.ldc = ld_in;
  g.A = in; g.B = dz; g.C = dW; g.M = din; g.N = dout; g.K = T; g.lda = ld_in; g.ldb = ld_out; g.ldc = ld_out;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i)
    if (tfk::gemm_bf16x3_dual(a, g, 0) != 0) { printf("dual launch not eligible\n"); return 1; }
  const int iters = getenv("TFK_ABL_ITERS") ? atoi(getenv("TFK_ABL_ITERS")) : 30;
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) tfk::gemm_bf16x3_dual(a, g, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("ABL=%d dual (dA + dW) frames %d, %d x %d: %7.1f us  %7.1f TF fp32-equivalent\n", TFKB_ABL, T, din, dout, ms * 1e3,
         4.0 * T * din * dout / ms / 1e9);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 5) return 1;
  const int layout = atoi(argv[1]), M = atoi(argv[2]), N = atoi(argv[3]), K = atoi(argv[4]);
  if (layout == 3) return dual(M, N, K);
  auto p8 = [](int n) { return (n + 31) & ~31; };  // (rows of whole interleave blocks, as the engine's twins)
  const int a_rows = layout == 2 ? K : M, a_cols = layout == 2 ? M : K;
  const int b_rows = layout == 1 ? N : K, b_cols = layout == 1 ? K : N;
  const int lda = p8(a_cols), ldb = p8(b_cols), ldc = (N + 3) & ~3;
  std::vector<uint16_t> ha(tfk::x3::elems(a_rows, lda)), hb(tfk::x3::elems(b_rows, ldb));
  fill_planes(ha, 12345u);
  fill_planes(hb, 54321u);
  uint16_t *dA, *dB;
  float* dC;
  hipMalloc(&dA, ha.size() * 2); hipMalloc(&dB, hb.size() * 2); hipMalloc(&dC, (size_t)M * ldc * 4);
  hipMemcpy(dA, ha.data(), ha.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(dB, hb.data(), hb.size() * 2, hipMemcpyHostToDevice);
  tfk::GemmArgsB g = {};
  g.A = dA; g.B = dB; g.C = dC; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  // the split-K form where the shape is eligible (TFK_BF16X3_CFG=0 / 1: never)
  if (const size_t need = tfk::gemm_bf16x3_splitk_floats((tfk::GemmLayout)layout, M, N, K)) {
    hipMalloc(&g.splitk_ws, need * 4);
    hipMemset(g.splitk_ws, 0, need * 4);
    g.splitk_ws_floats = need;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) tfk::gemm_bf16x3((tfk::GemmLayout)layout, g, 0);
  const int iters = getenv("TFK_ABL_ITERS") ? atoi(getenv("TFK_ABL_ITERS")) : 30;  // (thousands: long enough to sample power)
  hipEventRecord(e0, 0);
  for (int i = 0; i < iters; ++i) tfk::gemm_bf16x3((tfk::GemmLayout)layout, g, 0);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  printf("ABL=%d %s layout %d %dx%dx%d: %7.1f us  %7.1f TF fp32-equivalent\n", TFKB_ABL, g.splitk_ws ? "split-K" : "       ", layout, M, N, K, ms * 1e3, 2.0 * M * N * K / ms / 1e9);
  return 0;
}
