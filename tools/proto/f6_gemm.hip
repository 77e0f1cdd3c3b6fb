// PROTOTYPE (not part of the library): the two-unit split of DESIGN.md section 8 on the shape of tdnn3_conv.
//   out[m][n] = sum_{j < 7} sum_{c < 512} A[m + j][c] * W[j][c][n],   M = 74 752 rows, N = 512
//   a * w ~ f16(a) * f16(w)                                   v_mfma_f32_16x16x32_f16
//         + q6(f16(a)) * q6(w - f16(w)) + q6(a - f16(a)) * q6(f16(w))       2 x v_mfma_scale_f32_16x16x128_f8f6f4 (fp6 e2m3,
//                                                                          one E8M0 scale per 32 channels), K = 4 taps x 32 channels
// Activation block of one (row, 32 channels), 128 bytes: chunks 0-3 f16 hi | chunk 4 / 5 fp6(hi) / fp6(lo) bytes 0-15 | chunk 6
// their bytes 16-23 | chunk 7 the two scale bytes.  Operands are converted on the host; the kernel is the K loop (slab staging by
// LDS-DMA as in the product, weights double-buffered in registers per macro step [M M M M X], two workgroups per CU) with a raw
// fp32 store.  Prints accuracy against a float64 product on sampled outputs and the rate.
// build: hipcc -O3 --offload-arch=gfx950 tools/proto/f6_gemm.hip -o tools/proto/f6_gemm.bin
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

constexpr int BM = 128, BN = 128, DROW = 128, DA_ROWS = 136, DA_BYTES = DA_ROWS * DROW;
constexpr int CIN = 512, NCB = CIN / 32, TAPS = 7, TAP8 = 8, NQ = 2;

// weights: main [N/32][NCB][TAP8][2 ct][64 lanes][16 B]; cross [N/32][NCB][NQ][2 ct] x {hi_a 64x16, lo_a 64x16, tails 64x16 (hi 8 | lo 8), scales 64x4}
constexpr size_t WMAIN_CT = 64 * 16, WX_CT = 64 * 16 * 3 + 64 * 4;

__global__ __launch_bounds__(256, 2) void f6_gemm_kernel(const char* __restrict__ A, const char* __restrict__ Wmain,
                                                        const char* __restrict__ Wx, float* __restrict__ out, int M, int N, int nMt,
                                                        int nNt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c16 = lane & 15, g4 = lane >> 4;
  const int q8 = blockIdx.x & 7, iq = blockIdx.x >> 3, per = (nMt * nNt) >> 3;      // XCD-contiguous tile order (grid % 8 == 0)
  const int tile = q8 * per + iq;
  const int mt = tile / nNt, nt = tile - mt * nNt;
  const int m0 = mt * BM, n0 = nt * BN;
  const int lrow = lane >> 3, lpc = lane & 7;
  const int64_t a_row_bytes = (int64_t)CIN * 4;                                      // 16 blocks of 128 B
  const char* Ag = A + (int64_t)(m0 + lrow) * a_row_bytes;
  const uint32_t as_lds = (uint32_t)(uintptr_t)(lptr_t)smem;
  auto dma_a = [&](int cb, int buf, int g) {                                         // 8 rows x 128 B of channel block cb
    const int c = lpc ^ ((4 * g + (lrow >> 1)) & 7);
    const char* src = Ag + (int64_t)(8 * g) * a_row_bytes + (int64_t)cb * 128 + c * 16;
    const uint32_t dst = __builtin_amdgcn_readfirstlane(as_lds + buf * DA_BYTES + g * 1024);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(src), "s"(dst) : "memory");
  };
  const char* Wm_g = Wmain + ((int64_t)((n0 >> 5) + wave) * NCB * TAP8 * 2) * WMAIN_CT + lane * 16;
  const char* Wx_g = Wx + ((int64_t)((n0 >> 5) + wave) * NCB * NQ * 2) * WX_CT;

  f32x4 acc[8][2];
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) acc[g][c] = f32x4{0.f, 0.f, 0.f, 0.f};

  f16x8 Wm[2][4][2];          // main weights [buffer][tap of the macro step][channel tile], one macro step ahead
  v4i Xh[2], Xl[2];           // cross weights of the CURRENT macro step: fp6(hi) / fp6(lo) bytes 0-15 per channel tile; reloaded for the
  v2i Xth[2], Xtl[2];         // next step right after the cross pass, while the main pass runs
  int Xs[2];                  // scale bytes: 0 = fp6(hi), 1 = fp6(lo)
#define GLD16(dst, ptr, OFF) asm volatile("global_load_dwordx4 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
#define GLD8(dst, ptr, OFF) asm volatile("global_load_dwordx2 %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
#define GLD4(dst, ptr, OFF) asm volatile("global_load_dword %0, %1, off offset:" #OFF : "=v"(dst) : "v"(ptr))
  auto load_wm = [&](int step, int buf) __attribute__((always_inline)) {      // step = cb * NQ + q; 8 KB per wave and step
    const char* pm = Wm_g + (int64_t)step * (4 * 2) * WMAIN_CT;
    GLD16(Wm[buf][0][0], pm, 0);    GLD16(Wm[buf][0][1], pm, 1024); GLD16(Wm[buf][1][0], pm, 2048); GLD16(Wm[buf][1][1], pm, 3072);
    const char* pm2 = pm + 4096;
    GLD16(Wm[buf][2][0], pm2, 0);   GLD16(Wm[buf][2][1], pm2, 1024); GLD16(Wm[buf][3][0], pm2, 2048); GLD16(Wm[buf][3][1], pm2, 3072);
  };
  auto load_x = [&](int step) __attribute__((always_inline)) {
    // per channel tile 3 328 bytes: [64 x 16 hi | 64 x 16 lo | 64 x (8 hi tail | 8 lo tail) | 64 x 4 scales]
    const char* p0 = Wx_g + (int64_t)step * 2 * WX_CT + lane * 16;
    GLD16(Xh[0], p0, 0); GLD16(Xl[0], p0, 1024); GLD8(Xth[0], p0, 2048); GLD8(Xtl[0], p0, 2056);
    const char* p1 = p0 + WX_CT;
    GLD16(Xh[1], p1, 0); GLD16(Xl[1], p1, 1024); GLD8(Xth[1], p1, 2048); GLD8(Xtl[1], p1, 2056);
    const char* ps = Wx_g + (int64_t)step * 2 * WX_CT + 3072 + lane * 4;
    GLD4(Xs[0], ps, 0);
    const char* ps1 = ps + WX_CT;
    GLD4(Xs[1], ps1, 0);
  };
  const int nsteps = NCB * NQ;
  // prologue: slab 0, weights of step 0
  for (int g = wave; g < 17; g += 4) dma_a(0, 0, g);
  load_wm(0, 0);
  load_x(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // LDS offsets of this lane's fragments inside a slab, per macro step type q (the swizzle (row >> 1) & 7 does not depend on
  // the frame tile: 16 g rows further is + g * 2048 bytes, an immediate)
  int xo[2][4], mo[2][4];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int rx = c16 + 4 * q + g4, sx = (rx >> 1) & 7;       // cross: K group g4 = tap 4q + g4
    xo[q][0] = rx * DROW + ((4 ^ sx) << 4);
    xo[q][1] = rx * DROW + ((5 ^ sx) << 4);
    xo[q][2] = rx * DROW + ((6 ^ sx) << 4);
    xo[q][3] = rx * DROW + ((7 ^ sx) << 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = c16 + 4 * q + j;
      mo[q][j] = r * DROW + ((g4 ^ ((r >> 1) & 7)) << 4);
    }
  }
  // cross pass: the two block-scaled MFMAs of every tile (this lane's K group is tap 4q + g4 -> slab row 16g + c16 + 4q + g4)
  auto cross_pass = [&](int cb, int q) __attribute__((always_inline)) {
    const char* slab = smem + (cb & 1) * DA_BYTES;
    v4i fh[2], fl[2];
    v2i fth[2], ftl[2];
    int fs[2];
    auto read_cross = [&](int g, int slot) __attribute__((always_inline)) {
      fh[slot] = *reinterpret_cast<const v4i*>(slab + xo[q][0] + g * 2048);
      fl[slot] = *reinterpret_cast<const v4i*>(slab + xo[q][1] + g * 2048);
      fth[slot] = *reinterpret_cast<const v2i*>(slab + xo[q][2] + g * 2048);
      ftl[slot] = *reinterpret_cast<const v2i*>(slab + xo[q][2] + g * 2048 + 8);
      fs[slot] = *reinterpret_cast<const int*>(slab + xo[q][3] + g * 2048);
    };
    read_cross(0, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int sl = g & 1;
      if (g + 1 < 8) read_cross(g + 1, sl ^ 1);
      const v8i a_hi6 = {fh[sl][0], fh[sl][1], fh[sl][2], fh[sl][3], fth[sl][0], fth[sl][1], 0, 0};
      const v8i a_lo6 = {fl[sl][0], fl[sl][1], fl[sl][2], fl[sl][3], ftl[sl][0], ftl[sl][1], 0, 0};
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const v8i w_hi6 = {Xh[c][0], Xh[c][1], Xh[c][2], Xh[c][3], Xth[c][0], Xth[c][1], 0, 0};
        const v8i w_lo6 = {Xl[c][0], Xl[c][1], Xl[c][2], Xl[c][3], Xtl[c][0], Xtl[c][1], 0, 0};
        // w_lo * a_hi (weight scale byte 1, activation scale byte 0), then w_hi * a_lo (bytes 0 / 1)
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w_lo6, a_hi6, acc[g][c], 2, 2, 1, Xs[c], 0, fs[sl]);
        acc[g][c] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(w_hi6, a_lo6, acc[g][c], 2, 2, 0, Xs[c], 1, fs[sl]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // main pass: hi * hi of the four taps of the macro step
  auto main_pass = [&](int cb, int q, int buf) __attribute__((always_inline)) {
    const char* slab = smem + (cb & 1) * DA_BYTES;
    f16x8 fm[2][4];
    auto read_main = [&](int g, int slot) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) fm[slot][j] = *reinterpret_cast<const f16x8*>(slab + mo[q][j] + g * 2048);
    };
    read_main(0, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int sl = g & 1;
      if (g + 1 < 8) read_main(g + 1, sl ^ 1);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int c = 0; c < 2; ++c) acc[g][c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Wm[buf][j][c], fm[sl][j], acc[g][c], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  for (int step = 0; step < nsteps; step += 2) {        // two macro steps = one channel block; main weight buffers alternate statically
    const int cb = step >> 1;
    // ---- q = 0: main weights of (cb, 1) and the first half of slab cb + 1 go out first
    load_wm(step + 1, 1);
    if (cb + 1 < NCB) for (int g = wave; g < 9; g += 4) dma_a(cb + 1, (cb + 1) & 1, g);
    __builtin_amdgcn_sched_barrier(0);
    cross_pass(cb, 0);
    __builtin_amdgcn_sched_barrier(0);
    load_x(step + 1);                                   // cross weights of (cb, 1) land while the main pass runs
    __builtin_amdgcn_sched_barrier(0);
    main_pass(cb, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // ---- q = 1
    load_wm(step + 2 < nsteps ? step + 2 : 0, 0);
    if (cb + 1 < NCB) for (int g = 9 + wave; g < 17; g += 4) dma_a(cb + 1, (cb + 1) & 1, g);
    __builtin_amdgcn_sched_barrier(0);
    cross_pass(cb, 1);
    __builtin_amdgcn_sched_barrier(0);
    load_x(step + 2 < nsteps ? step + 2 : 0);
    __builtin_amdgcn_sched_barrier(0);
    main_pass(cb, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // slab cb + 1 complete; every wave is done reading slab cb
  }
  // raw store: lane holds channels 4 g4 .. 4 g4 + 3 of tile ct for frame c16 of frame tile g
#pragma unroll
  for (int g = 0; g < 8; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int m = m0 + 16 * g + c16, n = n0 + wave * 32 + c * 16 + 4 * g4;
      if (m < M) *reinterpret_cast<f32x4*>(out + (int64_t)m * N + n) = acc[g][c];
    }
}

// ---------------------------------------------------------------- host side
static uint16_t f2h(float x) { _Float16 h = (_Float16)x; uint16_t u; memcpy(&u, &h, 2); return u; }
static float h2f(uint16_t u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static float e2m3_val(int code) {
  const int s = (code >> 5) & 1, e = (code >> 3) & 3, m = code & 7;
  const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
  return s ? -v : v;
}
static int e2m3_enc(float v) {               // round to nearest representable, |v| <= 7.5 after scaling
  int best = 0;
  float bd = 1e30f;
  for (int c = 0; c < 64; ++c) { const float d = std::fabs(e2m3_val(c) - v); if (d < bd) { bd = d; best = c; } }
  return best;
}
// 32 values -> 24 bytes of fp6 codes (little-endian 6-bit fields) + E8M0 scale byte; returns the dequantised values in deq
static void quant_block(const float* v, unsigned char* codes24, unsigned char* scale, float* deq) {
  float amax = 0.f;
  for (int i = 0; i < 32; ++i) amax = std::fmax(amax, std::fabs(v[i]));
  int e = amax > 0.f ? (int)std::ceil(std::log2(amax / 7.5f)) : -127;
  if (e < -127) e = -127;
  if (e > 127) e = 127;
  const float s = std::ldexp(1.f, e);
  memset(codes24, 0, 24);
  for (int i = 0; i < 32; ++i) {
    const int c = e2m3_enc(v[i] / s);
    deq[i] = e2m3_val(c) * s;
    const int bit = 6 * i;
    for (int q = 0; q < 6; ++q) if ((c >> q) & 1) codes24[(bit + q) >> 3] |= 1 << ((bit + q) & 7);
  }
  *scale = (unsigned char)(127 + e);
}

int main(int argc, char** argv) {
  const int M = 74752, N = 512, rowsA = M + 8 + 128;          // slack rows: the last tile stages 136 rows
  const int nMt = M / BM, nNt = N / BN;
  srand(3);
  auto rnd = [] { float s = 0.f; for (int i = 0; i < 12; ++i) s += rand() / (float)RAND_MAX; return s - 6.f; };
  printf("building operands on the host ...\n");
  std::vector<float> A((size_t)rowsA * CIN), W((size_t)TAPS * CIN * N);
  for (auto& v : A) { const float x = rnd(); v = x > 0.f ? x : 0.f; }        // post-ReLU activations
  for (auto& v : W) v = (rand() / (float)RAND_MAX * 2.f - 1.f) * 0.0383f;     // Glorot range of a 3584 -> 512 layer
  // activations -> blocks
  std::vector<unsigned char> Ab((size_t)rowsA * CIN * 4, 0);
  std::vector<float> dq(32), hi(32), lo(32);
  for (int r = 0; r < rowsA; ++r)
    for (int cb = 0; cb < NCB; ++cb) {
      unsigned char* blk = &Ab[((size_t)r * NCB + cb) * 128];
      const float* x = &A[(size_t)r * CIN + cb * 32];
      for (int i = 0; i < 32; ++i) { const uint16_t h = f2h(x[i]); memcpy(blk + 2 * i, &h, 2); hi[i] = h2f(h); lo[i] = x[i] - hi[i]; }
      unsigned char c24[24], sc;
      quant_block(hi.data(), c24, &sc, dq.data()); memcpy(blk + 64, c24, 16); memcpy(blk + 96, c24 + 16, 8); blk[112] = sc;
      quant_block(lo.data(), c24, &sc, dq.data()); memcpy(blk + 80, c24, 16); memcpy(blk + 104, c24 + 16, 8); blk[113] = sc;
    }
  // weights
  std::vector<unsigned char> Wm((size_t)(N / 32) * NCB * TAP8 * 2 * WMAIN_CT, 0), Wxv((size_t)(N / 32) * NCB * NQ * 2 * WX_CT, 0);
  for (int n = 0; n < N; ++n) {
    const int nb = n >> 5, ct = (n >> 4) & 1, r16 = n & 15;
    for (int cb = 0; cb < NCB; ++cb)
      for (int j = 0; j < TAP8; ++j) {
        float w32[32], whi[32], wlo[32];
        for (int t = 0; t < 32; ++t) {
          w32[t] = j < TAPS ? W[((size_t)j * CIN + cb * 32 + t) * N + n] : 0.f;
          whi[t] = h2f(f2h(w32[t]));
          wlo[t] = w32[t] - whi[t];
        }
        // main: lane = 16 * k-chunk + row, 8 halfs of k = 8 chunk .. 8 chunk + 7
        unsigned char* pm = &Wm[((((size_t)nb * NCB + cb) * TAP8 + j) * 2 + ct) * WMAIN_CT];
        for (int kc = 0; kc < 4; ++kc)
          for (int e = 0; e < 8; ++e) { const uint16_t h = f2h(w32[8 * kc + e]); memcpy(pm + (16 * kc + r16) * 16 + 2 * e, &h, 2); }
        // cross: K group of tap j inside macro step q = j / 4 is g4 = j % 4 -> lane = 16 * (j & 3) + row
        unsigned char* px = &Wxv[((((size_t)nb * NCB + cb) * NQ + (j >> 2)) * 2 + ct) * WX_CT];
        const int ln = 16 * (j & 3) + r16;
        unsigned char c24[24], sc;
        quant_block(whi, c24, &sc, dq.data()); memcpy(px + ln * 16, c24, 16); memcpy(px + 2048 + ln * 16, c24 + 16, 8); px[3072 + ln * 4] = sc;
        quant_block(wlo, c24, &sc, dq.data()); memcpy(px + 1024 + ln * 16, c24, 16); memcpy(px + 2048 + ln * 16 + 8, c24 + 16, 8); px[3072 + ln * 4 + 1] = sc;
      }
  }
  char *dA, *dWm, *dWx; float* dOut;
  CHECK(hipMalloc(&dA, Ab.size())); CHECK(hipMalloc(&dWm, Wm.size())); CHECK(hipMalloc(&dWx, Wxv.size())); CHECK(hipMalloc(&dOut, (size_t)M * N * 4));
  CHECK(hipMemcpy(dA, Ab.data(), Ab.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dWm, Wm.data(), Wm.size(), hipMemcpyHostToDevice));
  CHECK(hipMemcpy(dWx, Wxv.data(), Wxv.size(), hipMemcpyHostToDevice));
  const size_t smem = 2 * DA_BYTES;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(f6_gemm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  auto launch = [&] { hipLaunchKernelGGL(f6_gemm_kernel, dim3(nMt * nNt), dim3(256), smem, 0, dA, dWm, dWx, dOut, M, N, nMt, nNt); };
  launch();
  CHECK(hipDeviceSynchronize());
  std::vector<float> out((size_t)M * N);
  CHECK(hipMemcpy(out.data(), dOut, out.size() * 4, hipMemcpyDeviceToHost));
  // accuracy on sampled rows against the float64 product of the fp32 operands
  double num = 0.0, den = 0.0;
  for (int s = 0; s < 48; ++s) {
    const int m = (int)((rand() / (double)RAND_MAX) * (M - 1));
    for (int n = 0; n < N; n += 7) {
      double ref = 0.0;
      for (int j = 0; j < TAPS; ++j)
        for (int c = 0; c < CIN; ++c) ref += (double)A[(size_t)(m + j) * CIN + c] * W[((size_t)j * CIN + c) * N + n];
      const double d = out[(size_t)m * N + n] - ref;
      num += d * d;
      den += ref * ref;
    }
  }
  printf("relative L2 error vs float64 on sampled outputs: %.3e\n", std::sqrt(num / den));
  const double seconds = argc > 1 ? atof(argv[1]) : 1.5;
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  float ms = 0.f, total = 0.f;
  do {
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < 10; ++i) launch();
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    total += ms;
  } while (total < seconds * 1e3);
  const double flop = 2.0 * M * N * TAPS * CIN;
  printf("%.4f ms per launch = %.0f TF algorithmic (tdnn3_conv shape; shipped bf16x3 kernel: 0.513 ms = 523 TF)\n", ms / 10, flop / (ms / 10) / 1e9);
  return 0;
}
