// Index and layout helpers for the ResNet-18 encoder (model/resnet.py:152-351).
//
// 2-D activations live on a zero-bordered NHWC "grid": utterance b owns (L_b + 2) time rows of S positions
// of C channels starting at position (off0[b] + 2b) * S; position 0 of a time row is the left border, positions
// 1..F the frequency bins.  The pitch S is F + 1 -- the right border of a row IS the left border of the next row --
// unless a consumer reads the value with a frequency stride of 2, which needs an even pitch: then S = F + 2.
// The border stays zero, so TensorFlow's 'same' padding of the 3x3 convolutions needs no bounds checks and every
// kernel row of a window is one contiguous run of 3*C values -- the convolution becomes the overlapping-row GEMM
// of xv_kernels.h (taps = kernel rows, tap stride = one padded time row).  The GEMM rows of a layer enumerate the
// positions of its INPUT (all of them, border included); rows that do not produce an output bin land on border
// positions of the output, and their row-map entry says "write zeros there" (xv_epilogue.h out_row): the border
// is re-zeroed by the epilogue, not by a memset of the whole value.
//
// Compact form (split-precision path, default): the GEMM rows enumerate only the OUTPUT BINS of a layer; row m reads its
// window through a second index (GemmArgs::arow: grid position of the window's top-left corner) and no MFMA work is
// spent on border rows -- 1 / (F + 1) of the positions, 17 % in stage 4 -- nor, under resnet_time_stride, on the input
// rows between two output frames.  The border of the output is then zeroed by grid_zero_border_kernel, which writes
// the border positions only.
#include "xv_epilogue.h"

namespace xv {

namespace {

// utterance whose block [start_b, start_{b+1}) contains r, with start_b = (off0[b] + 2b) * unit
__device__ __forceinline__ int find_utt(const int32_t* off0, int B, int64_t unit, int64_t r) {
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int64_t)(off0[mid] + 2 * mid) * unit <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// GEMM row m = (utterance b, padded time row t, j-th position of that row at the layer's frequency stride); output
// position (t + 1, j + 1) of the output grid (pitch So): an output bin when t < L_b and j < Fout, else a border
// position -- zeroed through this row when `cover` (the rows of a time row cover the whole output row: rows_per_t == So)
__global__ void rowmap_grid_kernel(const int32_t* __restrict__ off0, int B, int rows_per_t, int Fout, int So, int cover,
                                   int32_t* __restrict__ rowmap, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = find_utt(off0, B, rows_per_t, m);
  const int64_t local = m - (int64_t)(off0[b] + 2 * b) * rows_per_t;
  const int t = (int)(local / rows_per_t), j = (int)(local - (int64_t)t * rows_per_t);
  const int L = off0[b + 1] - off0[b];
  const int32_t pos = (int32_t)((int64_t)(off0[b] + 2 * b) * So + (int64_t)(t + 1) * So + j + 1);
  rowmap[m] = (t < L && j < Fout) ? pos : (cover ? -pos - 2 : -1);
}

// The same for a convolution that also strides TIME by 2 (resnet_time_stride, model/resnet.py:187).  The GEMM rows still
// enumerate every padded input time row t (at frequency stride 2); a row produces output frame t_o when it is the top row
// of that frame's window under TensorFlow's 'same' padding, which depends on the parity of the utterance's length:
//   3 x 3 (ktime = 3): L even pads (0, 1) -> window rows 2 t_o .. 2 t_o + 2 of the input = grid rows t = 2 t_o + 1 .. :
//                      t odd, t_o = (t - 1) / 2;   L odd pads (1, 1) -> grid rows t = 2 t_o .. : t even, t_o = t / 2
//   1 x 1 (ktime = 1): no padding, input row 2 t_o (the A operand already points at the window centre): t even, t_o = t / 2
// The other rows are skipped (-1): the caller zeroes the output as a whole.
__global__ void rowmap_grid_ts_kernel(const int32_t* __restrict__ off_in, const int32_t* __restrict__ off_out, int B,
                                      int rows_per_t, int Fout, int So, int ktime, int32_t* __restrict__ rowmap, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = find_utt(off_in, B, rows_per_t, m);
  const int64_t local = m - (int64_t)(off_in[b] + 2 * b) * rows_per_t;
  const int t = (int)(local / rows_per_t), j = (int)(local - (int64_t)t * rows_per_t);
  const int Lin = off_in[b + 1] - off_in[b], Lout = off_out[b + 1] - off_out[b];
  const int first = (ktime == 3 && (Lin & 1) == 0) ? 1 : 0;       // grid row of the window top of output frame 0
  const int to = (t - first) >> 1;
  const bool valid = t >= first && ((t - first) & 1) == 0 && to < Lout && j < Fout;
  rowmap[m] = valid ? (int32_t)((int64_t)(off_out[b] + 2 * b) * So + (int64_t)(to + 1) * So + j + 1) : -1;
}

// utterance whose block [off[b] * unit, off[b + 1] * unit) contains r (no border rows in this enumeration)
__device__ __forceinline__ int find_utt0(const int32_t* off, int B, int64_t unit, int64_t r) {
  int lo = 0, hi = B - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if ((int64_t)off[mid] * unit <= r) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Compact enumeration: GEMM row m = (utterance b, output frame t_o, output bin f_o).  arow[m] = input grid position
// (pitch Sin) of the top-left corner of the window, rowmap[m] = output grid position (pitch So) of the bin.
//   frequency: stride 1 -> padded column f_o;  stride 2 (Fin even: TF 'same' pads (0, 1)) -> 2 f_o + 1 for the 3x3 window,
//              2 f_o for the 1x1 shortcut (whose A operand is offset to the window centre, a_off = (Sin + 1) * cin)
//   time:      stride 1 -> padded row t_o;  stride 2 -> 2 t_o + first, first as in rowmap_grid_ts_kernel below
// Entries M .. Mpad - 1 of arow are position 0 (a tile always stages 128 rows).
__global__ void rowmap_grid_compact_kernel(const int32_t* __restrict__ off_in, const int32_t* __restrict__ off_out, int B,
                                           int Sin, int So, int Fout, int sw, int st, int ktime,
                                           int32_t* __restrict__ arow, int32_t* __restrict__ rowmap, int64_t M, int64_t Mpad) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= Mpad) return;
  if (m >= M) { arow[m] = 0; return; }
  const int b = find_utt0(off_out, B, Fout, m);
  const int64_t local = m - (int64_t)off_out[b] * Fout;
  const int to = (int)(local / Fout), fo = (int)(local - (int64_t)to * Fout);
  const int Lin = off_in[b + 1] - off_in[b];
  const int first = (st == 2 && ktime == 3 && (Lin & 1) == 0) ? 1 : 0;
  const int trow = st == 2 ? 2 * to + first : to;
  const int col = sw == 2 ? 2 * fo + (ktime == 3 ? 1 : 0) : fo;
  arow[m] = (int32_t)((int64_t)(off_in[b] + 2 * b) * Sin + (int64_t)trow * Sin + col);
  rowmap[m] = (int32_t)((int64_t)(off_out[b] + 2 * b) * So + (int64_t)(to + 1) * So + fo + 1);
}

// Zero the border positions of a grid value (pitch S, F bins; utterance b: top row, L_b x (S - F) side positions, bottom
// row) and the S positions behind its last row (with the shared border column the window of the last bin of the last
// utterance ends one position behind the grid).  One thread per (border position, 16-byte chunk); `y` (fp32 rows of
// chunks_y chunks) and / or `ysb` (split-blocked rows of chunks_sb chunks) may be null.
__global__ void grid_zero_border_kernel(const int32_t* __restrict__ off0, int B, int F, int S, int chunks_y, int chunks_sb,
                                        char* __restrict__ y, char* __restrict__ ysb, int64_t nborder) {
  const int side = S - F;
  const int cmax = chunks_y > chunks_sb ? chunks_y : chunks_sb;
  const int64_t total = nborder * cmax;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t k = i / cmax;
    const int c = (int)(i - k * cmax);
    const int64_t end = (int64_t)off0[B] * side + (int64_t)2 * S * B;      // border positions inside the grid
    int64_t pos;
    if (k >= end) {
      pos = (int64_t)(off0[B] + 2 * B) * S + (k - end);
    } else {
      int lo = 0, hi = B - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if ((int64_t)off0[mid] * side + (int64_t)2 * S * mid <= k) lo = mid; else hi = mid - 1;
      }
      const int b = lo, L = off0[b + 1] - off0[b];
      const int64_t local = k - ((int64_t)off0[b] * side + (int64_t)2 * S * b);
      const int64_t base = (int64_t)(off0[b] + 2 * b) * S;
      if (local < S) {
        pos = base + local;
      } else if (local < S + (int64_t)L * side) {
        const int64_t q = local - S;
        const int t = (int)(q / side), e = (int)(q - (int64_t)t * side);
        pos = base + (int64_t)(t + 1) * S + (e == 0 ? 0 : S - 1);
      } else {
        pos = base + (int64_t)(L + 1) * S + (local - S - (int64_t)L * side);
      }
    }
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    if (y && c < chunks_y) *reinterpret_cast<f32x4*>(y + (pos * chunks_y + c) * 16) = z;
    if (ysb && c < chunks_sb) *reinterpret_cast<f32x4*>(ysb + (pos * chunks_sb + c) * 16) = z;
  }
}

__global__ void rowmap_rows_kernel(const int32_t* __restrict__ off0, int B, int32_t* __restrict__ rowmap, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = find_utt(off0, B, 1, m);
  const int t = (int)(m - (off0[b] + 2 * b));
  const int L = off0[b + 1] - off0[b];
  rowmap[m] = (t >= 1 && t <= L) ? off0[b] + t - 1 : -1;
}

// Two-unit form of the stride-1 3 x 3 convolutions (gemm_f6v2_kernel, three taps along time): GEMM row m = padded time row m of the
// input grid = the top row of the windows of output frame t = m - (off0[b] + 2 b); the kernel's frequency bin j writes output row
// rowmap[m] + j of a value whose base pointer is advanced by one position, i.e. grid position (t + 1, j + 1).  The two bottom
// rows of an utterance start no window.
__global__ void rowmap_trows_kernel(const int32_t* __restrict__ off0, int B, int So, int32_t* __restrict__ rowmap, int64_t M) {
  const int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= M) return;
  const int b = find_utt(off0, B, 1, m);
  const int t = (int)(m - (off0[b] + 2 * b));
  const int L = off0[b + 1] - off0[b];
  rowmap[m] = t < L ? (int32_t)((m + 1) * So) : -1;
}

// conv0: the GEMM rows ARE the output positions (pitch S); border positions are written as zeros
__global__ void rowmap_interior_kernel(const int32_t* __restrict__ off0, int B, int F, int S, int32_t* __restrict__ rowmap,
                                       int64_t M) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= M) return;
  const int b = find_utt(off0, B, S, p);
  const int64_t local = p - (int64_t)(off0[b] + 2 * b) * S;
  const int t = (int)(local / S), f = (int)(local - (int64_t)t * S);
  const int L = off0[b + 1] - off0[b];
  rowmap[p] = (t >= 1 && t <= L && f >= 1 && f <= F) ? (int32_t)p : -(int32_t)p - 2;
}

// conv0 (3x3, cin = 1): one thread per (output grid position, pair of taps); taps 9..31 are zero padding
template <bool SB>
__global__ void im2col2d_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ off0, int B, int F,
                                int S, int64_t P, char* __restrict__ out, int f16, int* __restrict__ ovf) {
  const int64_t total = P * 16;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i >> 4;
    const int k = (int)(i & 15) * 2;
    const int b = find_utt(off0, B, S, p);
    const int64_t local = p - (int64_t)(off0[b] + 2 * b) * S;
    const int tp = (int)(local / S), fp = (int)(local - (int64_t)tp * S);
    const int L = off0[b + 1] - off0[b];
    float v[2] = {0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int kk = k + e;
      if (kk < 9) {
        const int t = tp + kk / 3 - 2, f = fp + kk % 3 - 2;       // output (tp-1, fp-1), tap (kh-1, kw-1)
        if (t >= 0 && t < L && f >= 0 && f < F) v[e] = x[(int64_t)(off0[b] + t) * ldx + f];
      }
    }
    if (SB) {
      uint32_t hi, lo;
      split2(v[0], v[1], hi, lo, f16);
      if (f16) ovf_report(ovf, fmaxf(fabsf(v[0]), fabsf(v[1])));
      char* blk = out + p * 128 + k * 2;
      *reinterpret_cast<uint32_t*>(blk) = hi;
      *reinterpret_cast<uint32_t*>(blk + 64) = lo;
    } else {
      float* row = reinterpret_cast<float*>(out) + p * 32 + k;
      row[0] = v[0];
      row[1] = v[1];
    }
  }
}

// conv0 directly (split precisions): one workgroup per padded time row of the batch, one thread per (position of the row, 8
// channels).  The workgroup finds its utterance once, stages the three feature rows of the window (zero padded: TF 'same',
// stride 1, one row / column each side) and the 9 x C kernel + BN scale / shift in LDS; an interior position sums its 3x3 window
// in fp32, applies scale / shift and the activation and stores 8 fp32 values and / or the 16-byte hi and lo chunks of its
// split-blocked row; border positions are written as zeros.  K = 9: the matrix pipe has nothing to do here -- as im2col + GEMM
// the layer was one K step of prologue and epilogue around a 100 MB detour.
constexpr int kConv0MaxC = 256, kConv0MaxF = 126;
__global__ __launch_bounds__(256) void conv0_direct_kernel(const float* __restrict__ x, int64_t ldx, const int32_t* __restrict__ off0,
                                                          int B, int F, int S, int C, const float* __restrict__ wdir, int act,
                                                          const float* __restrict__ alpha, float* __restrict__ y,
                                                          char* __restrict__ ysb, int ldsb, int f16, int* __restrict__ ovf, float sb_mul) {
  __shared__ float xs[3][kConv0MaxF + 2];
  __shared__ float wsm[11 * kConv0MaxC];
  const int64_t r = blockIdx.x;                               // padded time row of the batch
  const int b = find_utt(off0, B, 1, r);
  const int tp = (int)(r - (off0[b] + 2 * b));
  const int L = off0[b + 1] - off0[b];
  const bool inside = tp >= 1 && tp <= L;
  if (inside) {
    for (int i = threadIdx.x; i < 11 * C; i += 256) wsm[i] = wdir[i];
    for (int i = threadIdx.x; i < 3 * (F + 2); i += 256) {
      const int kh = i / (F + 2), c = i - kh * (F + 2);
      const int tt = tp - 2 + kh, ff = c - 1;                 // feature row / bin of window row kh, column c
      xs[kh][c] = (tt >= 0 && tt < L && ff >= 0 && ff < F) ? x[(int64_t)(off0[b] + tt) * ldx + ff] : 0.f;
    }
  }
  __syncthreads();
  const int groups = C >> 3;
  for (int item = threadIdx.x; item < S * groups; item += 256) {
    const int fp = item / groups, c0 = (item - fp * groups) * 8;
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (inside && fp >= 1 && fp <= F) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const float xv = xs[kh][fp - 1 + kw];
          const float* w = wsm + (kh * 3 + kw) * C + c0;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = fmaf(xv, w[e], v[e]);
        }
#pragma unroll
      for (int e = 0; e < 8; ++e)
        v[e] = apply_act(fmaf(v[e], wsm[9 * C + c0 + e], wsm[10 * C + c0 + e]), act, alpha ? alpha[c0 + e] : 0.f);
    }
    const int64_t p = r * S + fp;
    if (y) {
      float* o = y + p * C + c0;
      *reinterpret_cast<f32x4*>(o) = f32x4{v[0], v[1], v[2], v[3]};
      *reinterpret_cast<f32x4*>(o + 4) = f32x4{v[4], v[5], v[6], v[7]};
    }
    if (ysb) {
      uint32_t hi[4], lo[4];
      float m = 0.f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float a0 = v[2 * e] * sb_mul, a1 = v[2 * e + 1] * sb_mul;      // the split copy is kept at the layer's power-of-two scale
        split2(a0, a1, hi[e], lo[e], f16);
        m = fmaxf(m, fmaxf(fabsf(a0), fabsf(a1)));
      }
      if (f16) ovf_report(ovf, m);
      char* blk = ysb + p * (int64_t)ldsb * 4 + (c0 >> 5) * 128 + (c0 & 31) * 2;
      *reinterpret_cast<uint4*>(blk) = uint4{hi[0], hi[1], hi[2], hi[3]};
      *reinterpret_cast<uint4*>(blk + 64) = uint4{lo[0], lo[1], lo[2], lo[3]};
    }
  }
}

// 3x3 stride-1 'same' max-pool on a grid value (model/resnet.py:230-231): one thread per (position, 4 channels).  An
// interior position takes the maximum over its neighbours INSIDE the map (TensorFlow's 'same' max-pool ignores the
// padding, so the zero border must not take part: leaky / parametric ReLU outputs can be negative); a border
// position is written as zero, so the whole output grid is defined without a memset.  Reads fp32, writes fp32 and / or SB.
__global__ void grid_maxpool3x3_kernel(const float* __restrict__ x, const int32_t* __restrict__ off0, int B, int F, int S,
                                       int C, int64_t P, float* __restrict__ y, char* __restrict__ ysb, int ldsb, int f16,
                                       int* __restrict__ ovf, float sb_mul) {
  const int quads = C >> 2;
  const int64_t total = P * quads;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t p = i / quads;
    const int c = (int)(i - p * quads) * 4;
    const int b = find_utt(off0, B, S, p);
    const int64_t local = p - (int64_t)(off0[b] + 2 * b) * S;
    const int t = (int)(local / S), f = (int)(local - (int64_t)t * S);
    const int L = off0[b + 1] - off0[b];
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (t >= 1 && t <= L && f >= 1 && f <= F) {
      v = f32x4{-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      for (int dt = -1; dt <= 1; ++dt) {
        if (t + dt < 1 || t + dt > L) continue;
        for (int df = -1; df <= 1; ++df) {
          if (f + df < 1 || f + df > F) continue;
          const f32x4 u = *reinterpret_cast<const f32x4*>(x + (p + (int64_t)dt * S + df) * C + c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], u[e]);
        }
      }
    }
    if (y) *reinterpret_cast<f32x4*>(y + p * C + c) = v;
    if (ysb) {
      uint32_t h01, l01, h23, l23;
      v *= sb_mul;
      split2(v[0], v[1], h01, l01, f16);
      split2(v[2], v[3], h23, l23, f16);
      if (f16) ovf_report(ovf, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
      char* blk = ysb + p * (int64_t)ldsb * 4 + (c >> 5) * 128 + (c & 31) * 2;
      *reinterpret_cast<uint2*>(blk) = make_uint2(h01, h23);
      *reinterpret_cast<uint2*>(blk + 64) = make_uint2(l01, l23);
    }
  }
}

__global__ void grid_unpad_kernel(const float* __restrict__ grid, const int32_t* __restrict__ off0, int B, int F, int S, int C,
                                  float* __restrict__ out, int64_t total) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const int64_t pos = i / C;                 // dense position: frame * F + f
    const int f = (int)(pos % F);
    const int64_t frame = pos / F;
    int lo = 0, hi = B - 1;                    // utterance of this frame
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (off0[mid] <= frame) lo = mid; else hi = mid - 1;
    }
    const int t = (int)(frame - off0[lo]);
    const int64_t p = (int64_t)(off0[lo] + 2 * lo) * S + (int64_t)(t + 1) * S + f + 1;
    out[i] = grid[p * C + c];
  }
}

int launch_blocks(int64_t total) { return (int)((total + 255) / 256 > 8192 ? 8192 : (total + 255) / 256); }

}  // namespace

hipError_t launch_build_rowmap_grid(const int32_t* off0, int B, int rows_per_t, int Fout, int So, int cover, int32_t* rowmap,
                                    int64_t M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_grid_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, off0, B, rows_per_t, Fout, So,
                     cover, rowmap, M);
  return hipGetLastError();
}

hipError_t launch_build_rowmap_grid_ts(const int32_t* off_in, const int32_t* off_out, int B, int rows_per_t, int Fout, int So,
                                       int ktime, int32_t* rowmap, int64_t M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_grid_ts_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, off_in, off_out, B, rows_per_t,
                     Fout, So, ktime, rowmap, M);
  return hipGetLastError();
}

hipError_t launch_build_rowmap_grid_compact(const int32_t* off_in, const int32_t* off_out, int B, int Sin, int So, int Fout,
                                            int sw, int st, int ktime, int32_t* arow, int32_t* rowmap, int64_t M, int64_t Mpad,
                                            hipStream_t s) {
  if (Mpad <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_grid_compact_kernel, dim3((unsigned)((Mpad + 255) / 256)), dim3(256), 0, s, off_in, off_out, B, Sin,
                     So, Fout, sw, st, ktime, arow, rowmap, M, Mpad);
  return hipGetLastError();
}

hipError_t launch_grid_zero_border(const int32_t* off0, int B, int64_t frames, int F, int S, int chunks_y, int chunks_sb,
                                   float* y, void* ysb, hipStream_t s) {
  const int64_t nborder = frames * (S - F) + (int64_t)2 * S * B + S;
  const int cmax = chunks_y > chunks_sb ? chunks_y : chunks_sb;
  if ((!y && !ysb) || nborder * cmax <= 0) return hipSuccess;
  const int64_t blocks = (nborder * cmax + 255) / 256;
  hipLaunchKernelGGL(grid_zero_border_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, off0, B, F, S,
                     y ? chunks_y : 0, ysb ? chunks_sb : 0, reinterpret_cast<char*>(y), static_cast<char*>(ysb), nborder);
  return hipGetLastError();
}

hipError_t launch_conv0_direct(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int C, int64_t P,
                               const float* wdir, int act, const float* alpha, float* y, void* ysb, int ldsb, int f16, int* ovf,
                               float sb_mul, hipStream_t s) {
  const int64_t rows = S > 0 ? P / S : 0;                     // padded time rows of the batch
  if (rows <= 0) return hipSuccess;
  if (C > kConv0MaxC || F > kConv0MaxF || (C & 7) || rows * S != P) return hipErrorInvalidValue;
  hipLaunchKernelGGL(conv0_direct_kernel, dim3((unsigned)rows), dim3(256), 0, s, x, ldx, off0, B, F, S, C, wdir, act, alpha, y,
                     static_cast<char*>(ysb), ldsb, f16, ovf, sb_mul);
  return hipGetLastError();
}

hipError_t launch_build_rowmap_trows(const int32_t* off0, int B, int So, int32_t* rowmap, int64_t M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_trows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, off0, B, So, rowmap, M);
  return hipGetLastError();
}

hipError_t launch_build_rowmap_rows(const int32_t* off0, int B, int32_t* rowmap, int64_t M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_rows_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, off0, B, rowmap, M);
  return hipGetLastError();
}

hipError_t launch_build_rowmap_interior(const int32_t* off0, int B, int F, int S, int32_t* rowmap, int64_t M, hipStream_t s) {
  if (M <= 0) return hipSuccess;
  hipLaunchKernelGGL(rowmap_interior_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, s, off0, B, F, S, rowmap, M);
  return hipGetLastError();
}

hipError_t launch_im2col2d_sb(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int64_t P, void* out_sb,
                              int f16, int* ovf, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(im2col2d_kernel<true>, dim3(launch_blocks(P * 16)), dim3(256), 0, s, x, ldx, off0, B, F, S, P,
                     static_cast<char*>(out_sb), f16, ovf);
  return hipGetLastError();
}

hipError_t launch_im2col2d_f32(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int64_t P, float* out,
                               hipStream_t s) {
  if (P <= 0) return hipSuccess;
  hipLaunchKernelGGL(im2col2d_kernel<false>, dim3(launch_blocks(P * 16)), dim3(256), 0, s, x, ldx, off0, B, F, S, P,
                     reinterpret_cast<char*>(out), 0, nullptr);
  return hipGetLastError();
}

hipError_t launch_grid_maxpool3x3(const float* x, const int32_t* off0, int B, int F, int S, int C, int64_t P, float* y,
                                  void* ysb, int ldsb, int f16, int* ovf, float sb_mul, hipStream_t s) {
  if (P <= 0) return hipSuccess;
  if (C & 3) return hipErrorInvalidValue;
  hipLaunchKernelGGL(grid_maxpool3x3_kernel, dim3(launch_blocks(P * (C >> 2))), dim3(256), 0, s, x, off0, B, F, S, C, P, y,
                     static_cast<char*>(ysb), ldsb, f16, ovf, sb_mul);
  return hipGetLastError();
}

hipError_t launch_grid_unpad_n(const float* grid, const int32_t* off0, int B, int F, int S, int C, int64_t frames, float* out,
                               hipStream_t s) {
  const int64_t total = frames * F * C;
  if (total <= 0) return hipSuccess;
  hipLaunchKernelGGL(grid_unpad_kernel, dim3(launch_blocks(total)), dim3(256), 0, s, grid, off0, B, F, S, C, out, total);
  return hipGetLastError();
}

}  // namespace xv
