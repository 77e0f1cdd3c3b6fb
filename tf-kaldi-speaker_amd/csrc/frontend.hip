// On-GPU feature front-end: sliding-window cepstral mean normalisation + voiced-frame selection.
//
// The reference feeds extract.py through two Kaldi binaries (egs/voxceleb/v1/nnet/run_extract_embeddings.sh:47):
//   apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 scp:feats.scp ark:- |
//   select-voiced-frames ark:- scp,s,cs:vad.scp ark:- |
// Kaldi itself is not part of the reference tree; this restates the published algorithm of Kaldi's
// SlidingWindowCmn (feature-functions.cc: centred window [t - W/2, t - W/2 + W) shifted inside [0, T), mean
// over the window subtracted, arithmetic in double) and of select-voiced-frames (keep frame t iff vad[t] != 0).
// **parity unpinned** (no Kaldi binary or fixture available); checked against oracle/ref_frontend.py.
//
// Two kernels: per-utterance exclusive prefix sums over time (double, one thread per feature dimension), then
// one thread per (selected frame, dimension): window bounds -> mean from two prefix rows -> x - mean.
#include "xv_kernels.h"

namespace xv {

// prefix[(off[b] + b + t) * dim + d] = sum_{u < t} x[(off[b] + u) * ld + d], t = 0..T_b   (T_b + 1 rows per utterance)
__global__ void cmn_prefix_kernel(const float* __restrict__ x, int64_t ld, int dim, const int32_t* __restrict__ off,
                                  double* __restrict__ prefix) {
  const int b = blockIdx.x;
  const int d = threadIdx.x;
  if (d >= dim) return;
  const int r0 = off[b], r1 = off[b + 1];
  double s = 0.0;
  double* p = prefix + ((int64_t)r0 + b) * dim + d;
  for (int r = r0; r < r1; ++r) {
    *p = s;
    s += (double)x[(int64_t)r * ld + d];
    p += dim;
  }
  *p = s;
}

// out[r, d] = x[src[r], d] - mean over the Kaldi window of frame t = src[r] - off[b]
__global__ void cmn_select_kernel(const float* __restrict__ x, int64_t ld, int dim, const int32_t* __restrict__ off, int B,
                                  const double* __restrict__ prefix, const int32_t* __restrict__ src, int64_t out_rows,
                                  int window, int center, int min_window, float* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= out_rows * dim) return;
  const int64_t r = i / dim;
  const int d = (int)(i - r * dim);
  const int row = src[r];
  int lo = 0, hi = B - 1;                       // utterance of the source frame
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (off[mid] <= row) lo = mid; else hi = mid - 1;
  }
  if (window <= 0) {                             // selection only
    out[r * dim + d] = x[(int64_t)row * ld + d];
    return;
  }
  const int b = lo, T = off[b + 1] - off[b], t = row - off[b];
  int ws, we;                                    // Kaldi SlidingWindowCmnInternal window [ws, we)
  if (center) { ws = t - window / 2; we = ws + window; } else { ws = t - window; we = t + 1; }
  if (ws < 0) { we -= ws; ws = 0; }
  if (!center && we > t) we = max(t + 1, min_window);
  if (we > T) { ws -= (we - T); we = T; if (ws < 0) ws = 0; }
  const double* p = prefix + ((int64_t)off[b] + b) * dim + d;
  const double mean = (p[(int64_t)we * dim] - p[(int64_t)ws * dim]) / (double)(we - ws);
  out[r * dim + d] = (float)((double)x[(int64_t)row * ld + d] - mean);
}

hipError_t launch_cmn_select(const float* x, int64_t ld, int dim, const int32_t* off, int B, double* prefix,
                             const int32_t* src, int64_t out_rows, int window, int center, int min_window, float* out,
                             hipStream_t s) {
  if (B <= 0 || out_rows <= 0) return hipSuccess;
  const int threads = (dim + 63) / 64 * 64;
  hipError_t e = hipSuccess;
  if (window > 0) {
    hipLaunchKernelGGL(cmn_prefix_kernel, dim3(B), dim3(threads), 0, s, x, ld, dim, off, prefix);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int64_t total = out_rows * dim;
  hipLaunchKernelGGL(cmn_select_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ld, dim, off, B, prefix,
                     src, out_rows, window, center, min_window, out);
  return hipGetLastError();
}

}  // namespace xv
