// Post-step of the extraction recipe on the GPU: length normalisation and per-speaker means.
//
// The reference runs two Kaldi binaries behind extract.py (egs/voxceleb/v1/nnet/run_extract_embeddings.sh:80-103):
//   ivector-normalize-length --scaleup=false      v -> v / ||v||_2   (zero vectors are left as they are)
//   ivector-mean ark:spk2utt ...                  per speaker: sum of its utterances' vectors, scaled by 1/count
// Kaldi is not part of the reference tree; this restates the published algorithm of ivector-normalize-length.cc
// (ratio = norm, or norm / sqrt(dim) with --scaleup=true; ratio == 0 -> unchanged) and ivector-mean.cc
// (Vector<float> accumulation in spk2utt order, Scale(1.0 / utt_count)).  **parity unpinned** (no Kaldi binary or
// fixture available); checked against oracle/ref_post.py.
//
// Both are HBM-bound row sweeps of the [N, dim] embedding matrix (2 KB per x-vector): one wave per vector for the
// norm (coalesced float loads, DPP/shuffle reduction), one thread per (speaker, column) for the mean with the
// additions in spk2utt order, so the result is bit-identical to the sequential float32 sum of the oracle.
#include "xv_kernels.h"

namespace xv {

__global__ __launch_bounds__(256) void length_norm_kernel(const float* __restrict__ x, int64_t ldx, int64_t rows, int dim,
                                                          int scaleup, float* __restrict__ y, int64_t ldy) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* xr = x + r * ldx;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) s = fmaf(xr[c], xr[c], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  float ratio = sqrtf(s);
  if (scaleup) ratio = ratio / sqrtf((float)dim);
  const float inv = ratio == 0.f ? 1.f : 1.0f / ratio;           // "Zero iVector": written unchanged
  for (int c = lane; c < dim; c += 64) y[r * ldy + c] = xr[c] * inv;
}

hipError_t launch_length_norm(const float* x, int64_t ldx, int64_t rows, int dim, int scaleup, float* y, int64_t ldy,
                              hipStream_t s) {
  if (rows <= 0) return hipSuccess;
  hipLaunchKernelGGL(length_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, ldx, rows, dim, scaleup, y, ldy);
  return hipGetLastError();
}

// out[s, c] = (sum_{i in [spk_off[s], spk_off[s+1])} x[utt[i], c]) * (1 / count); speakers without utterances -> 0
__global__ __launch_bounds__(256) void speaker_mean_kernel(const float* __restrict__ x, int64_t ldx, int dim,
                                                           const int32_t* __restrict__ spk_off, const int32_t* __restrict__ utt,
                                                           int64_t num_spk, float* __restrict__ out, int64_t ldo) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= num_spk * dim) return;
  const int64_t s = i / dim;
  const int c = (int)(i - s * dim);
  const int b = spk_off[s], e = spk_off[s + 1];
  float acc = 0.f;
  for (int k = b; k < e; ++k) acc += x[(int64_t)utt[k] * ldx + c];
  out[s * ldo + c] = e > b ? acc * (float)(1.0 / (double)(e - b)) : 0.f;
}

hipError_t launch_speaker_mean(const float* x, int64_t ldx, int dim, const int32_t* spk_off, const int32_t* utt,
                               int64_t num_spk, float* out, int64_t ldo, hipStream_t s) {
  if (num_spk <= 0) return hipSuccess;
  const int64_t total = num_spk * dim;
  hipLaunchKernelGGL(speaker_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, dim, spk_off, utt,
                     num_spk, out, ldo);
  return hipGetLastError();
}

}  // namespace xv
