// Diagnostic harness: runs the LIBRARY's f6_from_sb_kernel and gemm_f16f6_kernel (csrc/gemm_f16f6.hip, included as source) on
// synthetic data with exactly sized buffers, synchronising and checking after each launch.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I include tools/proto/f6_lib_check.hip -o tools/proto/f6_lib_check.bin
#include "../../tf-kaldi-speaker_amd/csrc/gemm_f16f6.hip"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static uint16_t f2h(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }

int main() {
  using namespace xv;
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int CIN = 512, N = 512, W = 7, rows_in = 1041, M = rows_in - (W - 1), slack = 512;
  srand(5);
  auto rnd = [] { float s = 0.f; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
  std::vector<float> A((size_t)rows_in * CIN), Wt((size_t)W * CIN * N);
  for (auto& v : A) { const float x = rnd(); v = x > 0.f ? x : 0.f; }
  for (auto& v : Wt) v = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.0383f;
  // SB(f16) input rows
  std::vector<uint16_t> sb((size_t)(rows_in + slack) * CIN * 2, 0);
  for (int r = 0; r < rows_in; ++r)
    for (int c = 0; c < CIN; ++c) {
      const float x = A[(size_t)r * CIN + c];
      const uint16_t h = f2h(x);
      const size_t blk = ((size_t)r * (CIN / 32) + c / 32) * 64;
      sb[blk + (c & 31)] = h;
      sb[blk + 32 + (c & 31)] = f2h(x - h2f(h));
    }
  printf("host packing of the weights is the library's job: this harness only checks that the kernels run (no reference)\n");
  // weights: zeros are enough to exercise every access (main 16 KB x ncb per 32-channel block, cross likewise)
  const int ncb = CIN / 32;
  const size_t wm_bytes = (size_t)(N / 32) * ncb * 8 * 2 * 1024, wx_bytes = (size_t)(N / 32) * ncb * 2 * 2 * (64 * 16 * 3 + 64 * 4);
  char *dsb, *df6, *dwm, *dwx; float *dy, *dscale, *dshift;
  CHECK(hipMalloc(&dsb, sb.size() * 2)); CHECK(hipMalloc(&df6, sb.size() * 2)); CHECK(hipMalloc(&dwm, wm_bytes)); CHECK(hipMalloc(&dwx, wx_bytes));
  CHECK(hipMalloc(&dy, (size_t)M * N * 4)); CHECK(hipMalloc(&dscale, N * 4)); CHECK(hipMalloc(&dshift, N * 4));
  CHECK(hipMemcpy(dsb, sb.data(), sb.size() * 2, hipMemcpyHostToDevice));
  CHECK(hipMemset(df6, 0, sb.size() * 2)); CHECK(hipMemset(dwm, 0, wm_bytes)); CHECK(hipMemset(dwx, 0, wx_bytes));
  std::vector<float> ones(N, 1.f), zeros(N, 0.f);
  CHECK(hipMemcpy(dscale, ones.data(), N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dshift, zeros.data(), N * 4, hipMemcpyHostToDevice));
  CHECK(launch_f6_from_sb(dsb, df6, rows_in, CIN / 32, 0));
  CHECK(hipDeviceSynchronize());
  printf("converter ok\n");
  GemmArgs a{};
  a.cin = CIN; a.M = M; a.K = W * CIN; a.N = N; a.Kpad = W * CIN; a.Npad = N;
  a.scale = dscale; a.shift = dshift; a.act = ACT_RELU; a.Y = dy; a.ldy = N;
  a.Xsb = df6; a.ldsbx = CIN; a.Wfr = dwm; a.Wx6 = dwx; a.f16 = 1;
  CHECK(launch_gemm_f16f6(a, 0));
  CHECK(hipDeviceSynchronize());
  printf("gemm ok (fp32 output)\n");
  std::vector<float> y((size_t)M * N);
  CHECK(hipMemcpy(y.data(), dy, y.size() * 4, hipMemcpyDeviceToHost));
  double s = 0.0; for (float v : y) s += std::fabs(v);
  printf("sum |y| = %g (zero weights -> 0)\n", s);
  return 0;
}
