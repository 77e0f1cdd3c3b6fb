// HBM-bound kernels of the x-vector path: statistics pooling, attentive pooling, row maps,
// l2 scaling and small elementwise stages.  All are wavefront(64)-shaped reductions:
// coalesced 16-byte loads, DPP/shuffle reduction inside a wave, one LDS hop across waves.
#include "xv_epilogue.h"

namespace xv {


__device__ __forceinline__ float act_apply(float v, int act, float alpha) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_LRELU: return fmaxf(v, kLreluAlpha * v);
    case ACT_PRELU: return fmaxf(v, 0.0f) + alpha * (v - fabsf(v)) * 0.5f;
    case ACT_TANH: return tanhf(v);
    default: return v;
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ------------------------------------------------------------------------------ row map
__global__ void build_rowmap_kernel(const int32_t* __restrict__ off0, int B, int ctx_in, int w,
                                    int32_t* __restrict__ rowmap, int M) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  int lo = 0, hi = B - 1;            // largest b with in_off[b] <= r
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off0[mid] - mid * ctx_in <= r) lo = mid; else hi = mid - 1;
  }
  const int b = lo;
  const int in0 = off0[b] - b * ctx_in;
  const int len = (off0[b + 1] - (b + 1) * ctx_in) - in0;
  const int t = r - in0;
  rowmap[r] = (t < len - (w - 1)) ? (off0[b] - b * (ctx_in + w - 1) + t) : -1;
}

hipError_t launch_build_rowmap(const int32_t* off0, int B, int ctx_in, int w, int32_t* rowmap, int M,
                               hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(build_rowmap_kernel, dim3((M + 255) / 256), dim3(256), 0, s, off0, B, ctx_in, w,
                     rowmap, M);
  return hipGetLastError();
}

__global__ void build_row2utt_kernel(const int32_t* __restrict__ off0, int B, int ctx, int32_t* __restrict__ row2utt,
                                     int M) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= M) return;
  int lo = 0, hi = B - 1;            // largest b with first row <= r
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off0[mid] - mid * ctx <= r) lo = mid; else hi = mid - 1;
  }
  row2utt[r] = (r < off0[lo + 1] - (lo + 1) * ctx) ? lo : -1;
}

hipError_t launch_build_row2utt(const int32_t* off0, int B, int ctx, int32_t* row2utt, int M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(build_row2utt_kernel, dim3((M + 255) / 256), dim3(256), 0, s, off0, B, ctx, row2utt, M);
  return hipGetLastError();
}

// Fused statistics pooling, second half: merge the per-segment statistics of one utterance in
// slot order (deterministic).  Segment i has n_i frames, sum s_i and M2_i = sum (x - s_i/n_i)^2:
//   mean = sum s_i / L,   M2 = sum [ M2_i + n_i (s_i/n_i - mean)^2 ]   (Chan et al. pairwise merge)
// which equals the reference's two-pass mean / squared-difference variance (model/pooling.py:41-48)
// up to fp32 rounding.  grid (ceil(C/256), B).
__global__ __launch_bounds__(256) void pool_finalize_kernel(const float* __restrict__ part, int C,
                                                            const int32_t* __restrict__ off0, int ctx,
                                                            const int32_t* __restrict__ slotbase,
                                                            float* __restrict__ out, int64_t ldo) {
  const int b = blockIdx.y;
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= C) return;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  const int t0 = r0 >> 6, t1 = (r1 - 1) >> 6;
  const int64_t base = (int64_t)slotbase[b];
  float s = 0.f;
  for (int t = t0; t <= t1; ++t) s += part[((base + t) * 2) * C + n];
  const float mean = s / (float)(r1 - r0);
  float m2 = 0.f;
  for (int t = t0; t <= t1; ++t) {
    const int ni = min(r1, (t + 1) << 6) - max(r0, t << 6);
    const float d = part[((base + t) * 2) * C + n] / (float)ni - mean;
    m2 += part[((base + t) * 2 + 1) * C + n] + (float)ni * d * d;
  }
  float var = m2 / (float)(r1 - r0);
  var = var <= kVarFloor ? kVarFloor : var;                     // model/pooling.py:46-48
  out[(int64_t)b * ldo + n] = mean;
  out[(int64_t)b * ldo + C + n] = sqrtf(var);
}

hipError_t launch_pool_finalize(const float* part, int C, const int32_t* off0, int B, int ctx,
                                const int32_t* slotbase, float* out, int64_t ldo, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(pool_finalize_kernel, dim3((C + 255) / 256, B), dim3(256), 0, s, part, C, off0, ctx, slotbase,
                     out, ldo);
  return hipGetLastError();
}

// ------------------------------------------------------------------ statistics pooling
// grid (colChunks, B); block 256 = 16 row slots x 16 lanes x float4 (64 columns per block).
// Pass 1: mean.  Pass 2: population variance as mean of squared differences to that mean
// (the two-pass form of model/pooling.py:41-42; the second sweep hits L2).  Floor, sqrt, concat.
template <bool VEC>
__global__ __launch_bounds__(256) void stat_pool_kernel(const float* __restrict__ x, int64_t ldx, int C,
                                                        const int32_t* __restrict__ off0, int ctx,
                                                        float* __restrict__ out, int64_t ldo) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = tid & 15, rs = tid >> 4;
  const int col = blockIdx.x * 64 + cl * 4;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  const int L = r1 - r0;
  const float invL = 1.0f / (float)L;

  auto load4 = [&](int r) -> f32x4 {
    const float* p = x + (int64_t)r * ldx + col;
    if (VEC) {
      if (col < C) return *reinterpret_cast<const f32x4*>(p);
      return f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (col + e < C) ? p[e] : 0.f;
    return v;
  };
  auto block_sum = [&](f32x4 v) -> f32x4 {
#pragma unroll
    for (int e = 0; e < 4; ++e) {          // row slots of this wave: lanes 16 apart
      v[e] += __shfl_xor(v[e], 16, 64);
      v[e] += __shfl_xor(v[e], 32, 64);
    }
    __syncthreads();
    if (lane < 16) {
#pragma unroll
      for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = v[e];
    }
    __syncthreads();
    f32x4 t;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      t[e] = red[0][cl * 4 + e] + red[1][cl * 4 + e] + red[2][cl * 4 + e] + red[3][cl * 4 + e];
    return t;
  };

  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rs; r < r1; r += 16) s += load4(r);
  f32x4 mean = block_sum(s) * invL;
  f32x4 q = {0.f, 0.f, 0.f, 0.f};
  for (int r = r0 + rs; r < r1; r += 16) {
    f32x4 d = load4(r) - mean;
    q += d * d;
  }
  f32x4 var = block_sum(q) * invL;
  if (rs == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      if (col + e < C) {
        const float v = var[e] <= kVarFloor ? kVarFloor : var[e];   // model/pooling.py:46-48
        out[(int64_t)b * ldo + col + e] = mean[e];
        out[(int64_t)b * ldo + C + col + e] = sqrtf(v);
      }
    }
  }
}

hipError_t launch_stat_pool(const float* x, int64_t ldx, int C, const int32_t* off0, int B, int ctx,
                            float* out, int64_t ldo, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  dim3 grid((C + 63) / 64, B), block(256);
  const bool vec = (C % 4 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  if (vec)
    hipLaunchKernelGGL(stat_pool_kernel<true>, grid, block, 0, s, x, ldx, C, off0, ctx, out, ldo);
  else
    hipLaunchKernelGGL(stat_pool_kernel<false>, grid, block, 0, s, x, ldx, C, off0, ctx, out, ldo);
  return hipGetLastError();
}

// ------------------------------------------------------------------- attentive pooling
// scores[r, h] = scale * <key[r, head slice], query[h]>; one wave per row.
__global__ __launch_bounds__(256) void att_scores_kernel(const float* __restrict__ key, int64_t ldk,
                                                         int64_t rows, const float* __restrict__ query,
                                                         int H, int dk_h, int split_key, float scale,
                                                         float* __restrict__ scores) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  for (int h = 0; h < H; ++h) {
    const float* kp = key + r * ldk + (split_key ? h * dk_h : 0);
    const float* qp = query + (int64_t)h * dk_h;
    float s = 0.f;
    for (int d = lane; d < dk_h; d += 64) s = fmaf(kp[d], qp[d], s);
    s = wave_sum(s);
    if (lane == 0) scores[r * H + h] = s * scale;
  }
}

hipError_t launch_att_scores(const float* key, int64_t ldk, int64_t rows, const float* query, int H,
                             int dk_h, int split_key, float scale, float* scores, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_scores_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, key, ldk, rows,
                     query, H, dk_h, split_key, scale, scores);
  return hipGetLastError();
}

// softmax over the frames of one utterance for one head (tf.nn.softmax over the last axis of
// [b,h,l], model/pooling.py:197), in place.  grid (H, B).
__global__ __launch_bounds__(256) void att_softmax_kernel(float* __restrict__ scores, int H,
                                                          const int32_t* __restrict__ off0, int ctx) {
  __shared__ float red[4];
  const int h = blockIdx.x, b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  float m = -INFINITY;
  for (int r = r0 + tid; r < r1; r += 256) m = fmaxf(m, scores[(int64_t)r * H + h]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  float s = 0.f;
  for (int r = r0 + tid; r < r1; r += 256) {
    const float e = expf(scores[(int64_t)r * H + h] - m);
    scores[(int64_t)r * H + h] = e;
    s += e;
  }
  s = wave_sum(s);
  if (lane == 0) red[wave] = s;
  __syncthreads();
  const float inv = 1.0f / (red[0] + red[1] + red[2] + red[3]);
  for (int r = r0 + tid; r < r1; r += 256) scores[(int64_t)r * H + h] *= inv;
}

hipError_t launch_att_softmax(float* scores, int H, const int32_t* off0, int B, int ctx, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_softmax_kernel, dim3(H, B), dim3(256), 0, s, scores, H, off0, ctx);
  return hipGetLastError();
}

// weighted mean and weighted variance around that mean (model/pooling.py:201-218).
// grid (colChunks of 64, B, split ? 1 : H); block 256 = 4 row slots (waves) x 64 columns.
__global__ __launch_bounds__(256) void att_pool_kernel(const float* __restrict__ value, int64_t ldv, int dv,
                                                       const float* __restrict__ w, int H, int split_value,
                                                       const int32_t* __restrict__ off0, int ctx,
                                                       float* __restrict__ out, int64_t ldo) {
  __shared__ float red[4][64];
  const int b = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = blockIdx.x * 64 + lane;
  const bool cok = col < dv;
  const int dvh = split_value ? dv / H : dv;
  const int head = split_value ? (cok ? col / dvh : 0) : (int)blockIdx.z;
  const int ocol = split_value ? col : head * dv + col;
  const int odim = split_value ? dv : dv * H;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;

  auto block_sum = [&](float v) -> float {
    __syncthreads();
    red[wave][lane] = v;
    __syncthreads();
    return red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
  };
  float s = 0.f;
  for (int r = r0 + wave; r < r1; r += 4) {
    const float x = cok ? value[(int64_t)r * ldv + col] : 0.f;
    s = fmaf(x, w[(int64_t)r * H + head], s);
  }
  const float mean = block_sum(s);
  float q = 0.f;
  for (int r = r0 + wave; r < r1; r += 4) {
    const float x = cok ? value[(int64_t)r * ldv + col] : 0.f;
    const float d = x - mean;
    q = fmaf(d * d, w[(int64_t)r * H + head], q);
  }
  float var = block_sum(q);
  if (wave == 0 && cok) {
    var = var <= kVarFloor ? kVarFloor : var;
    out[(int64_t)b * ldo + ocol] = mean;
    out[(int64_t)b * ldo + odim + ocol] = sqrtf(var);
  }
}

hipError_t launch_att_pool(const float* value, int64_t ldv, int dv, const float* weights, int H,
                           int split_value, const int32_t* off0, int B, int ctx, float* out, int64_t ldo,
                           hipStream_t s) {
  if (B <= 0) return hipSuccess;
  dim3 grid((dv + 63) / 64, B, split_value ? 1 : H);
  hipLaunchKernelGGL(att_pool_kernel, grid, dim3(256), 0, s, value, ldv, dv, weights, H, split_value, off0,
                     ctx, out, ldo);
  return hipGetLastError();
}

// ---- fused attentive pooling: the three small kernels around the attention epilogue of the GEMM (xv_epilogue.h)
// scores[r, h] = scale * sum over the 32-channel blocks (ascending) of the partial dot products the key layer stored
__global__ __launch_bounds__(256) void att_scores_reduce_kernel(const float* __restrict__ part, int64_t ld, int nblk, int H,
                                                                int64_t rows, float scale, float* __restrict__ scores) {
  const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int b = 0; b < nblk; ++b) s += part[((int64_t)b * H + h) * ld + r];     // coalesced across r
    scores[r * H + h] = s * scale;
  }
}

hipError_t launch_att_scores_reduce(const float* part, int64_t ld, int nblk, int H, int64_t rows, float scale,
                                    float* scores, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_scores_reduce_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, s, part, ld, nblk, H, rows,
                     scale, scores);
  return hipGetLastError();
}

// s0[slot * H + h]: sum of the (normalised) weights of utterance b inside each 64-row tile; one thread per slot and head
__global__ __launch_bounds__(64) void att_slot_sums_kernel(const float* __restrict__ w, int H, const int32_t* __restrict__ off0,
                                                           int ctx, const int32_t* __restrict__ slotbase, float* __restrict__ s0) {
  const int b = blockIdx.x, h = blockIdx.y;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  const int t0 = r0 >> 6, t1 = (r1 - 1) >> 6;
  for (int t = t0 + (int)threadIdx.x; t <= t1; t += 64) {
    const int a = max(r0, t << 6), e = min(r1, (t + 1) << 6);
    float s = 0.f;
    for (int r = a; r < e; ++r) s += w[(int64_t)r * H + h];
    s0[((int64_t)slotbase[b] + t) * H + h] = s;
  }
}

hipError_t launch_att_slot_sums(const float* weights, int H, const int32_t* off0, int B, int ctx, const int32_t* slotbase,
                                float* s0, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_slot_sums_kernel, dim3(B, H), dim3(64), 0, s, weights, H, off0, ctx, slotbase, s0);
  return hipGetLastError();
}

// Weighted Chan merge of the per-slot moments of one utterance, in slot order (deterministic):
//   mean = sum_s s1_s            (the weights of an utterance sum to 1: model/pooling.py:197,203)
//   var  = sum_s [ m2_s + s0_s (s1_s / s0_s - mean)^2 ]      (= sum_t w_t (x_t - mean)^2, :204-206)
// grid (ceil(odim / 256), B); column oc belongs to head oc / dvh (split value) or oc / (odim / H).
__global__ __launch_bounds__(256) void att_pool_finalize_kernel(const float* __restrict__ part, const float* __restrict__ s0,
                                                                int odim, int H, int dvh, int split,
                                                                const int32_t* __restrict__ off0, int ctx,
                                                                const int32_t* __restrict__ slotbase, float* __restrict__ out,
                                                                int64_t ldo) {
  const int b = blockIdx.y;
  const int oc = blockIdx.x * 256 + threadIdx.x;
  if (oc >= odim) return;
  const int h = min(split ? oc / dvh : oc / (odim / H), H - 1);
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  const int t0 = r0 >> 6, t1 = (r1 - 1) >> 6;
  const int64_t base = (int64_t)slotbase[b];
  float mean = 0.f;
  for (int t = t0; t <= t1; ++t) mean += part[((base + t) * 2) * odim + oc];
  float var = 0.f;
  for (int t = t0; t <= t1; ++t) {
    const float w = s0[(base + t) * H + h];
    const float d = w > 0.f ? part[((base + t) * 2) * odim + oc] / w - mean : 0.f;
    var += part[((base + t) * 2 + 1) * odim + oc] + w * d * d;
  }
  var = var <= kVarFloor ? kVarFloor : var;                     // model/pooling.py:215-216
  out[(int64_t)b * ldo + oc] = mean;
  out[(int64_t)b * ldo + odim + oc] = sqrtf(var);
}

hipError_t launch_att_pool_finalize(const float* part, const float* s0, int odim, int H, int dvh, int split,
                                    const int32_t* off0, int B, int ctx, const int32_t* slotbase, float* out, int64_t ldo,
                                    hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_pool_finalize_kernel, dim3((odim + 255) / 256, B), dim3(256), 0, s, part, s0, odim, H, dvh, split,
                     off0, ctx, slotbase, out, ldo);
  return hipGetLastError();
}

// endpoints["attention_weights"] [b,h,l] for a uniform-length batch (model/pooling.py:198)
__global__ void att_weights_out_kernel(const float* __restrict__ scores, int H,
                                       const int32_t* __restrict__ off0, int ctx, float* __restrict__ out) {
  const int h = blockIdx.x, b = blockIdx.y;
  const int r0 = off0[b] - b * ctx, r1 = off0[b + 1] - (b + 1) * ctx;
  const int L = r1 - r0;
  for (int t = threadIdx.x; t < L; t += blockDim.x)
    out[((int64_t)b * H + h) * L + t] = scores[(int64_t)(r0 + t) * H + h];
}

hipError_t launch_att_weights_out(const float* scores, int H, const int32_t* off0, int B, int ctx,
                                  float* out, hipStream_t s) {
  if (B <= 0) return hipSuccess;
  hipLaunchKernelGGL(att_weights_out_kernel, dim3(H, B), dim3(256), 0, s, scores, H, off0, ctx, out);
  return hipGetLastError();
}

// ------------------------------------------------------------ first-layer im2col -> SB
// One thread per (row, pair of k): the 30-dim input is tiny (36 KB/utterance), so materialising
// the w*cin-wide rows once lets the first layer run on the bf16x3 MFMA kernel as a dense layer.
__global__ void im2col_sb_kernel(const float* __restrict__ x, int64_t ldx, int cin, int K, int64_t rows,
                                 char* __restrict__ out, int ldsb, int f16, int* __restrict__ ovf) {
  const int pairs = ldsb >> 1;
  const int64_t total = rows * pairs;
  float mx = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t m = i / pairs;
    const int k = (int)(i - m * pairs) * 2;
    float v[2];
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = k + e;
      const int dr = kk / cin;
      v[e] = kk < K ? x[(m + dr) * ldx + (kk - dr * cin)] : 0.f;
    }
    uint32_t hi, lo;
    split2(v[0], v[1], hi, lo, f16);                 // hi/lo split in the format the GEMM consumes (xv_epilogue.h)
    if (f16) ovf_report(ovf, fmaxf(fabsf(v[0]), fabsf(v[1])));      // a feature beyond the fp16 range
    mx = fmaxf(mx, fmaxf(fabsf(v[0]), fabsf(v[1])));
    char* blk = out + m * (int64_t)ldsb * 4 + (k >> 5) * 128 + (k & 31) * 2;
    *reinterpret_cast<uint32_t*>(blk) = hi;
    *reinterpret_cast<uint32_t*>(blk + 64) = lo;
  }
  // fp16 split: the other end of the range.  Hidden activations are kept at a per-layer power-of-two scale (GemmArgs::sb_mul), the
  // features are whatever the caller sends: every workgroup leaves the largest magnitude it staged in its own word behind the flag
  // words (no atomics: 19 200 same-address atomics per batch took 0.2 ms, four times the layer this staging feeds -- and a version
  // that skipped the atomic below the current maximum paid them again after every reset); flags_snapshot_kernel / xv_check_overflow
  // reduce the words, and the host refuses a batch whose features all sit below 2^-8, where the low halves are subnormal and even
  // the largest value keeps fewer than 17 bits.
  if (f16 && ovf) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if ((threadIdx.x & 63) == 0)           // one word per wave: no barrier, no atomic
      reinterpret_cast<float*>(ovf + kFlagWords)[blockIdx.x * 4 + (threadIdx.x >> 6)] = (mx <= 3.0e38f) ? mx : 0.f;
  }
}

hipError_t launch_im2col_sb(const float* x, int64_t ldx, int cin, int w, int64_t rows, void* out_sb, int ldsb, int f16,
                            int* ovf, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  const int64_t total = rows * (ldsb >> 1);
  const int blocks = (int)((total + 255) / 256 > kFeatMaxSlots / 4 ? kFeatMaxSlots / 4 : (total + 255) / 256);
  hipLaunchKernelGGL(im2col_sb_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, cin, w * cin, rows,
                     static_cast<char*>(out_sb), ldsb, f16, ovf);
  return hipGetLastError();
}

// --------------------------------------------------------------------- small elementwise
__global__ void affine_act_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int C,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const float* __restrict__ alpha, int act, float* __restrict__ y,
                                  int64_t ldy) {
  const int64_t n = rows * C;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / C;
    const int c = (int)(i - r * C);
    const float v = fmaf(x[r * ldx + c], scale ? scale[c] : 1.f, shift ? shift[c] : 0.f);
    y[r * ldy + c] = act_apply(v, act, alpha ? alpha[c] : 0.f);
  }
}

hipError_t launch_affine_act(const float* x, int64_t ldx, int64_t rows, int C, const float* scale,
                             const float* shift, const float* alpha, int act, float* y, int64_t ldy,
                             hipStream_t s) {
  const int64_t n = rows * C;
  if (n <= 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(affine_act_kernel, dim3(blocks), dim3(256), 0, s, x, ldx, rows, C, scale, shift, alpha,
                     act, y, ldy);
  return hipGetLastError();
}

// l2_scaling: one wave per row
__global__ __launch_bounds__(256) void l2_scale_kernel(const float* __restrict__ x, int64_t rows, int C,
                                                       float factor, float* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  float s = 0.f;
  for (int c = lane; c < C; c += 64) {
    const float v = x[r * C + c];
    s = fmaf(v, v, s);
  }
  s = wave_sum(s);
  const float inv = rsqrtf(fmaxf(s, 1e-12f)) * factor;   // model/common.py:56
  for (int c = lane; c < C; c += 64) y[r * C + c] = x[r * C + c] * inv;
}

hipError_t launch_l2_scale(const float* x, int64_t rows, int C, float factor, float* y, hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(l2_scale_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, rows, C, factor, y);
  return hipGetLastError();
}

// fp16 range flags -> pinned host memory, and cleared for the next forward: ONE small kernel in stream order.  (A memcpy D2H plus a
// memset are two engine switches between two forwards: together with the result copy they cost 0.15 ms of device time per 1.27 ms
// batch -- tools/pipeline_probe.py.)
__global__ void flags_snapshot_kernel(int* __restrict__ dev, int* __restrict__ host) {
  float* part = reinterpret_cast<float*>(dev + kFlagWords);          // per-wave feature maxima of the staging kernel
  __shared__ float wmax[16];
  float mx = 0.f;
  for (int i = threadIdx.x; i < kFeatMaxSlots; i += 1024) {
    mx = fmaxf(mx, part[i]);
    part[i] = 0.f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 16; ++i) mx = fmaxf(mx, wmax[i]);
    host[0] = dev[0];
    host[1] = __float_as_int(fmaxf(mx, __int_as_float(dev[1])));
    dev[0] = 0;
    dev[1] = 0;
    __threadfence_system();
  }
}

hipError_t launch_flags_snapshot(int* dev, int* host, hipStream_t s) {
  hipLaunchKernelGGL(flags_snapshot_kernel, dim3(1), dim3(1024), 0, s, dev, host);
  return hipGetLastError();
}

__global__ void copy2d_kernel(const float* __restrict__ src, int64_t lds, float* __restrict__ dst,
                              int64_t ldd, int64_t rows, int cols) {
  const int64_t n = rows * cols;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / cols;
    const int c = (int)(i - r * cols);
    dst[r * ldd + c] = src[r * lds + c];
  }
}

hipError_t launch_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols,
                         hipStream_t s) {
  const int64_t n = rows * cols;
  if (n <= 0) return hipSuccess;
  const int blocks = (int)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256);
  hipLaunchKernelGGL(copy2d_kernel, dim3(blocks), dim3(256), 0, s, src, lds, dst, ldd, rows, cols);
  return hipGetLastError();
}

}  // namespace xv
