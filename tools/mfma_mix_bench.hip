// Microbenchmark: how fast would a 2-unit split (f16 hi*hi + two block-scaled fp6 cross terms) issue against today's 3 x bf16
// split, on random operands, at the product kernel's occupancy (3 workgroups of 4 waves per CU, 16 accumulator tiles per wave)?
//   mix A (today)   : per K = 32 step and 16x16 tile   3 x v_mfma_f32_16x16x32_bf16                       (48 per wave-step)
//   mix B (proposed): per K = 128 step and tile        4 x v_mfma_f32_16x16x32_f16 + 2 x v_mfma_scale_f32_16x16x128_f8f6f4 (fp6 e2m3)
//   mix C           : the same with fp8 e4m3 cross terms
// (mix B / C hold 56-64 weight registers per K = 128 step: 134-162 VGPRs in this loop, still three workgroups per CU.)
// Operands stay in registers (variant 0) or the activation-side fragments are re-read from LDS for every frame tile (variant 1), as
// the product kernel does.  Prints microseconds per K = 128 of one wave-tile set and the ratio.  tests/analysis/f16f8_error_model.py has the
// accuracy side (1.2e-5 on the TDNN x-vector with e2m3 cross terms; bar 1e-4).
// build: hipcc -O3 --offload-arch=gfx950 tools/mfma_mix_bench.hip -o tools/mfma_mix_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int LDSREAD>
__global__ __launch_bounds__(256, 3) void mix_a(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ bf16x8 lds[2 * 8 * 64];
  const int lane = threadIdx.x & 63;
  bf16x8 W[4];
  for (int i = 0; i < 4; ++i) W[i] = src[(threadIdx.x * 4 + i) & 4095];
  for (int i = threadIdx.x; i < 2 * 8 * 64; i += 256) lds[i] = src[(i * 7 + blockIdx.x) & 4095];
  __syncthreads();
  bf16x8 fh = src[(threadIdx.x + 1111) & 4095], fl = src[(threadIdx.x + 2222) & 4095];
  f32x4 acc[8][2];
  for (int g = 0; g < 8; ++g) for (int c = 0; c < 2; ++c) acc[g][c] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (LDSREAD) { fh = lds[g * 64 + lane]; fl = lds[(8 + g) * 64 + lane]; }
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[0], fl, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[1], fl, acc[g][1], 0, 0, 0);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[2], fh, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[3], fh, acc[g][1], 0, 0, 0);
      acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[0], fh, acc[g][0], 0, 0, 0);
      acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(W[1], fh, acc[g][1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int g = 0; g < 8; ++g) for (int c = 0; c < 2; ++c) s += acc[g][c][0] + acc[g][c][1] + acc[g][c][2] + acc[g][c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// FMT: 2 = fp6 e2m3 (6 VGPRs per operand), 0 = fp8 e4m3 (8 VGPRs)
template <int LDSREAD, int FMT, int OCC = 3>
__global__ __launch_bounds__(256, OCC) void mix_b(const f16x8* __restrict__ src, const v8i* __restrict__ src8, float* __restrict__ out,
                                                int iters) {
  __shared__ f16x8 lds[4 * 2 * 64];      // two frame tiles' worth, re-read alternately (keeps three workgroups per CU)
  __shared__ v8i lds8[2 * 2 * 64];
  __shared__ int occ_pad[OCC == 2 ? 14000 : 1];   // OCC == 2: 56 KB more, so that only two workgroups fit a CU
  if (iters < 0) occ_pad[threadIdx.x] = iters;
  const int lane = threadIdx.x & 63;
  f16x8 W[2][4];
  v8i W8h[2], W8l[2];
  for (int c = 0; c < 2; ++c) {
    for (int k = 0; k < 4; ++k) W[c][k] = src[(threadIdx.x * 8 + c * 4 + k) & 4095];
    W8h[c] = src8[(threadIdx.x * 4 + c) & 4095];
    W8l[c] = src8[(threadIdx.x * 4 + 2 + c) & 4095];
  }
  for (int i = threadIdx.x; i < 4 * 2 * 64; i += 256) lds[i] = src[(i * 7 + blockIdx.x) & 4095];
  for (int i = threadIdx.x; i < 2 * 2 * 64; i += 256) lds8[i] = src8[(i * 5 + blockIdx.x) & 4095];
  __syncthreads();
  f16x8 f[4];
  for (int k = 0; k < 4; ++k) f[k] = src[(threadIdx.x + 1111 * (k + 1)) & 4095];
  v8i f8h = src8[(threadIdx.x + 777) & 4095], f8l = src8[(threadIdx.x + 999) & 4095];
  const int sc = 127;                       // E8M0 scale 2^0
  f32x4 acc[8][2];
  for (int g = 0; g < 8; ++g) for (int c = 0; c < 2; ++c) acc[g][c] = f32x4{0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      if (LDSREAD) {
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = lds[(k * 2 + (g & 1)) * 64 + lane];
        f8h = lds8[(g & 1) * 64 + lane];
        f8l = lds8[(2 + (g & 1)) * 64 + lane];
      }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W8h[c], f8l, acc[g][c], FMT, FMT, 0, sc, 0, sc);
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(W8l[c], f8h, acc[g][c], FMT, FMT, 0, sc, 0, sc);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[g][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[0][k], f[k], acc[g][0], 0, 0, 0);
        acc[g][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[1][k], f[k], acc[g][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
  for (int g = 0; g < 8; ++g) for (int c = 0; c < 2; ++c) s += acc[g][c][0] + acc[g][c][1] + acc[g][c][2] + acc[g][c][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short f2bf(float x) { unsigned u; memcpy(&u, &x, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }
static float rnd() { float s = 0.f; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; }

template <typename F>
static double time_it(F launch, double seconds) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  // keep the chip busy for `seconds`, then time the last batch of launches (clock has settled)
  float ms = 0.f, total = 0.f;
  do {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    total += ms;
  } while (total < seconds * 1e3);
  return ms / 10.0;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  const int n = 4096, wgs = 768;
  std::vector<unsigned short> hb(n * 8), hf(n * 8);
  std::vector<int> h8(n * 8);
  for (int i = 0; i < n * 8; ++i) { const float x = rnd(); hb[i] = f2bf(x); _Float16 h = (_Float16)x; memcpy(&hf[i], &h, 2); h8[i] = (rand() << 16) ^ rand(); }
  void *db, *df, *d8; float* out;
  CHECK(hipMalloc(&db, n * 16)); CHECK(hipMalloc(&df, n * 16)); CHECK(hipMalloc(&d8, n * 32)); CHECK(hipMalloc(&out, wgs * 256 * 4));
  CHECK(hipMemcpy(db, hb.data(), n * 16, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(df, hf.data(), n * 16, hipMemcpyHostToDevice));
  CHECK(hipMemcpy(d8, h8.data(), n * 32, hipMemcpyHostToDevice));
  const int itA = 4000, itB = 1000;            // the same K: 4000 x 32 = 1000 x 128
  for (int v = 0; v < 2; ++v) {
    double a = v ? time_it([&] { hipLaunchKernelGGL(mix_a<1>, dim3(wgs), dim3(256), 0, 0, (const bf16x8*)db, out, itA); }, seconds)
                 : time_it([&] { hipLaunchKernelGGL(mix_a<0>, dim3(wgs), dim3(256), 0, 0, (const bf16x8*)db, out, itA); }, seconds);
    double b6 = v ? time_it([&] { hipLaunchKernelGGL((mix_b<1, 2>), dim3(wgs), dim3(256), 0, 0, (const f16x8*)df, (const v8i*)d8, out, itB); }, seconds)
                  : time_it([&] { hipLaunchKernelGGL((mix_b<0, 2>), dim3(wgs), dim3(256), 0, 0, (const f16x8*)df, (const v8i*)d8, out, itB); }, seconds);
    double b8 = v ? time_it([&] { hipLaunchKernelGGL((mix_b<1, 0>), dim3(wgs), dim3(256), 0, 0, (const f16x8*)df, (const v8i*)d8, out, itB); }, seconds)
                  : time_it([&] { hipLaunchKernelGGL((mix_b<0, 0>), dim3(wgs), dim3(256), 0, 0, (const f16x8*)df, (const v8i*)d8, out, itB); }, seconds);
    // algorithmic FLOPs of one launch: 768 WGs x 4 waves x 16 tiles x (16 x 16 x 128 x 2) x 1000
    const double flop = 768.0 * 4 * 16 * (16.0 * 16 * 128 * 2) * 1000;
    printf("%s: 3 x bf16 %.3f ms = %.0f TF algorithmic | f16 + 2 x fp6 %.3f ms = %.0f TF (x%.2f) | f16 + 2 x fp8 %.3f ms = %.0f TF (x%.2f)\n",
           v ? "fragments from LDS " : "operands in registers", a, flop / a / 1e9, b6, flop / b6 / 1e9, a / b6, b8, flop / b8 / 1e9, a / b8);
  }
  {   // the proposed mix at TWO workgroups per CU (a simpler kernel could double-buffer its weights in 256 VGPRs)
    double b6 = time_it([&] { hipLaunchKernelGGL((mix_b<1, 2, 2>), dim3(wgs), dim3(256), 0, 0, (const f16x8*)df, (const v8i*)d8, out, itB); }, seconds);
    const double flop = 768.0 * 4 * 16 * (16.0 * 16 * 128 * 2) * 1000;
    printf("fragments from LDS, two workgroups per CU: f16 + 2 x fp6 %.3f ms = %.0f TF\n", b6, flop / b6 / 1e9);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
