// Probe of v_mfma_scale_f32_16x16x128_f8f6f4 with fp6 (e2m3) operands on gfx950: which K elements does a lane hold, how are the 32
// 6-bit values packed into its 6 VGPRs, and what does the per-lane E8M0 scale apply to?  Hypotheses are checked against a CPU
// product on random representable values.  (Groundwork for the two-unit split of DESIGN.md section 8.)
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_scale_probe.hip -o tools/mfma_scale_probe.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void probe(const v8i* a, const v8i* b, const int* sa, const int* sb, f32x4* d) {
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 2, 2, 0, sa[threadIdx.x], 0, sb[threadIdx.x]);
  d[threadIdx.x] = c;
}

// e2m3: 1 sign, 2 exponent (bias 1), 3 mantissa; subnormal when exponent field is 0
static float e2m3(int code) {
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

int main() {
  srand(7);
  std::vector<int> ca(16 * 128), cb(128 * 16);        // fp6 codes: A[i][k], B[k][j]
  for (auto& c : ca) c = rand() & 63;
  for (auto& c : cb) c = rand() & 63;
  std::vector<int> ea(64), eb(64);                    // E8M0 scale per lane
  for (int l = 0; l < 64; ++l) { ea[l] = 125 + rand() % 5; eb[l] = 125 + rand() % 5; }
  // hypothesis: lane l holds row / column (l & 15), K group (l >> 4): k = 32 * (l >> 4) + t, t = 0..31, packed little-endian 6 bits each
  std::vector<v8i> ha(64), hb(64);
  for (int l = 0; l < 64; ++l) {
    unsigned char bytes_a[32] = {0}, bytes_b[32] = {0};
    for (int t = 0; t < 32; ++t) {
      const int k = 32 * (l >> 4) + t, bit = 6 * t;
      const int va = ca[(l & 15) * 128 + k], vb = cb[k * 16 + (l & 15)];
      for (int q = 0; q < 6; ++q) {
        if ((va >> q) & 1) bytes_a[(bit + q) >> 3] |= 1 << ((bit + q) & 7);
        if ((vb >> q) & 1) bytes_b[(bit + q) >> 3] |= 1 << ((bit + q) & 7);
      }
    }
    memcpy(&ha[l], bytes_a, 32);
    memcpy(&hb[l], bytes_b, 32);
  }
  v8i *da, *db; int *dsa, *dsb; f32x4* dd;
  CHECK(hipMalloc(&da, 64 * 32)); CHECK(hipMalloc(&db, 64 * 32)); CHECK(hipMalloc(&dsa, 256)); CHECK(hipMalloc(&dsb, 256)); CHECK(hipMalloc(&dd, 64 * 16));
  CHECK(hipMemcpy(da, ha.data(), 64 * 32, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(db, hb.data(), 64 * 32, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dsa, ea.data(), 256, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dsb, eb.data(), 256, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dd);
  std::vector<float> hd(64 * 4);
  CHECK(hipMemcpy(hd.data(), dd, 64 * 16, hipMemcpyDeviceToHost));
  // reference: D[i][j] = sum_k A[i][k] * 2^(ea[lane(i, k / 32)] - 127) * B[k][j] * 2^(eb[lane(j, k / 32)] - 127); C/D map: col = lane & 15,
  // row = 4 * (lane >> 4) + reg
  double worst = 0.0, scale_free_worst = 0.0;
  for (int l = 0; l < 64; ++l)
    for (int r = 0; r < 4; ++r) {
      const int i = 4 * (l >> 4) + r, j = l & 15;
      double ref = 0.0, ref_noscale = 0.0;
      for (int k = 0; k < 128; ++k) {
        const double p = (double)e2m3(ca[i * 128 + k]) * e2m3(cb[k * 16 + j]);
        ref += p * std::ldexp(1.0, ea[i + 16 * (k / 32)] - 127) * std::ldexp(1.0, eb[j + 16 * (k / 32)] - 127);
        ref_noscale += p;
      }
      worst = std::fmax(worst, std::fabs(ref - hd[l * 4 + r]));
      scale_free_worst = std::fmax(scale_free_worst, std::fabs(ref_noscale - hd[l * 4 + r]));
    }
  printf("hypothesis lane = (row|col l&15, K group l>>4), 32 x 6 bits little-endian, scale per lane = per (row, K group): max |D - ref| = %.3g"
         "  (ignoring the scales: %.3g)\n", worst, scale_free_worst);
  printf(worst < 1e-3 ? "LAYOUT CONFIRMED\n" : "layout NOT confirmed\n");
  return 0;
}
