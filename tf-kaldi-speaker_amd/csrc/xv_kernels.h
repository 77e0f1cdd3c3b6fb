// Internal launch interface between the C-ABI layer (xvec_api.hip) and the gfx950 kernels.
// Host-side structs only; nothing here is exported from the shared library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xv {

// Epilogue activation applied after  v = acc * scale[n] + shift[n].
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_LRELU = 2, ACT_PRELU = 3, ACT_TANH = 4 };

constexpr float kLreluAlpha = 0.2f;       // tf.nn.leaky_relu default (model/tdnn.py:33)
// fp16 range guard buffer of a handle: kFlagWords ints ([0] overflow flag, [1] bits of a feature maximum carried over), then
// kFeatMaxSlots floats: the largest feature magnitude each wave of the feature staging kernel saw since the last read-out
constexpr int kFlagWords = 4, kFeatMaxSlots = 32768;      // 8192 workgroups x 4 waves
constexpr float kVarFloor = 1e-12f;       // VAR2STD_EPSILON (model/pooling.py:6)

// One "overlapping-row" GEMM:  Y[rowmap[m], n] = act((sum_k A[m,k] * Wt[n,k]) * scale[n] + shift[n])
//   A[m,k] = X[(m + k / cin) * ldx + k % cin]      (== X[m*ldx + k] when ldx == cin)
// A dense layer has K == cin; a temporal convolution of width w has K == w*cin, so that row m
// of A is the w consecutive input frames starting at frame m (model/tdnn.py:42-47).
struct GemmArgs {
  const float* X;         // activations, fp32
  int64_t ldx;            // row stride of X in elements
  int cin;                // channels per input frame
  int M;                  // rows of A to compute (input rows minus w-1)
  int K;                  // w * cin
  int N;                  // output channels
  const float* Wt;        // packed weights [Npad][Kpad], k contiguous, zero padded
  int Kpad;               // multiple of 32
  int Npad;               // multiple of 128
  const float* scale;     // [N]
  const float* shift;     // [N]
  const float* alpha;     // [N] PReLU slopes (ACT_PRELU) or nullptr
  int act;
  const int32_t* rowmap;  // [M] output row or -1 (skip); nullptr = identity
  float* Y;               // fp32 output (may be nullptr when only the split planes are wanted)
  int64_t ldy;
  // bf16x3 path: activations consumed by a split GEMM are carried in the split-blocked (SB)
  // format of xv_epilogue.h (per row: blocks of [32 x bf16 hi | 32 x bf16 lo], 128 bytes).
  const void* Xsb;        // SB input (bf16x3 kernel), row stride ldsbx elements (x4 bytes)
  int64_t ldsbx;          // multiple of 32; == cin for convolutions
  void* Ysb;              // SB output or nullptr
  int ldsb;               // channels per SB output row incl. zero padding (multiple of 32)
  const void* Wsb;        // packed weights in SB format [Npad][Kpad/32][128 bytes]
  const void* Wfr = nullptr;   // weights, MFMA-fragment-major (gemm_bf16x3_wreg_kernel)
  const void* Wx6 = nullptr;   // XV_PREC_F16F6 layers: block-scaled fp6 cross-term weights (gemm_f16f6.hip; Wfr then holds the f16 main weights)
  int ysb_f6 = 0;              // gemm_f16f6 kernel: write Ysb in ITS activation block format (f16 hi | fp6 hi / lo codes | scales) -- the
                               // consumer is another two-unit layer and no conversion pass is needed
  int slab3 = 1;               // one-tap layers: three slab buffers / slabs two steps ahead (gemm_bf16x3_w1p3_kernel); 0 = the
                               // two-buffer kernel (same arithmetic, bit-identical results; xv_set_option "slab3")
  int* ovf = nullptr;          // f16 only: set to 1 when a value beyond the fp16 range was converted (checked by the host)
  int nbin = 0;                // gemm_f6v2_kernel, ResNet form: frequency bins per time row (a second tile dimension), and the byte
  int64_t bin_x_bytes = 0;     // offset of a bin's windows inside a row of A; bin b writes output rows rowmap[m] + b
  float sb_mul = 1.f;          // fp16 split formats: the split-blocked / fp6-block output holds y * sb_mul, a per-layer power of two that
                               // keeps the activations' low halves normal (|y * sb_mul| ~ 2^4 rms; the reader's per-channel scale
                               // carries 1 / sb_mul exactly: xvec_api.hip, act_exponent).  The fp32 output Y is never scaled.
  int f16 = 0;                 // split format of both operands: 0 = bf16 hi/lo (bf16x3), 1 = fp16 hi/lo (f16x3: same layout and
                               // MFMA rate, 11 + 11 significand bits instead of 8 + 8; values beyond +-65504 overflow)
  long long* trace = nullptr;  // debug: per-workgroup phase timestamps (XVEC_TRACE_K), 4 per workgroup
  // split-K (fp32 kernel, small-M segment layers): slice s of `ksplit` accumulates K tiles
  // [s*kper, (s+1)*kper) and writes RAW accumulators to partial[s][M][Npad]; a reduce kernel
  // sums the slices in order (deterministic) and applies the epilogue.
  int ksplit;             // 0/1 = off
  int raw;                // epilogue: store acc unchanged (scale 1, shift 0, no activation)
  float* partial;         // [ksplit][M][Npad] fp32 workspace (bf16x3 tail form: [ksplit][tail_mt * 128][Npad])
  int tail_mt = 0;        // bf16x3 kernel: the last `tail_mt` M tiles are computed K-split in `ksplit` slices (see
                          // gemm_bf16x3_tail_plan) and finished by a reduce kernel
  // fused statistics pooling (dense layer feeding statistics_pooling, model/pooling.py:27-52):
  // instead of storing the activations, every 64-row wave tile writes, per channel and per
  // utterance segment inside the tile, sum(x) and sum((x - segment mean)^2) to
  //   pool_part[(slot * 2 + {0,1}) * N + n],  slot = pool_slotbase[b] + (row >> 6)
  // (each slot is written exactly once -> deterministic); launch_pool_finalize merges the slots.
  float* pool_part;
  const int32_t* pool_row2utt;    // [M] utterance index of each row
  const int32_t* pool_slotbase;   // [B]
  // generalised A addressing (2-D convolutions of the ResNet encoder on a zero-bordered NHWC grid,
  // model/resnet.py:25-79,217-261).  A[m, tap*ktap + kk] = X[a_off + m*a_pitch + tap*tap_stride + kk]
  // (element units).  a_pitch == 0 selects the default 1-D form above.  All of a_off, a_pitch,
  // tap_stride, ktap are multiples of 32 on the bf16x3 path (whole SB blocks).
  int64_t a_pitch;
  int64_t a_off;
  int64_t tap_stride;
  int ntaps;              // K == ntaps * ktap
  int ktap;
  // Gathered grid form (bf16x3 one-tap kernel only): GEMM row m reads the A row that starts at grid position arow[m]
  // (units of a_pitch elements) instead of position m, so the rows can enumerate just the output bins of a layer --
  // no MFMA work on border positions (csrc/grid.hip, rowmap_grid_compact_kernel).  arow is padded to a multiple of
  // 128 rows with position 0.
  const int32_t* arow;
  // residual shortcut added before the activation (resnet.py:84-85,147-148): fp32 [rows, N]
  const float* R;
  int64_t ldr;
  // ---- fused attentive pooling (model/pooling.py:189-217), bf16x3 kernel with the attention epilogue (EPI = 1):
  // (a) last key layer: instead of storing the key, every wave stores the partial scores of its 32 channels,
  //       att_part[((n / 32) * att_heads + h) * att_ld + m] = sum_{n' in block} act(...)[m, n'] * att_q[h * Npad + n']
  //     (att_q = the query expanded per head over the padded key width, zero outside the head's slice; the scale of
  //     :193-194 and the ordered sum over the blocks are applied by launch_att_scores_reduce);
  const float* att_q = nullptr;
  float* att_part = nullptr;
  int64_t att_ld = 0;
  int att_heads = 0;
  // (b) value layer: instead of storing the value, per 64-frame tile and utterance segment the weighted moments
  //       pool_part[(slot * 2 + 0) * pool_odim + oc] = sum_t w[t, h] x[t, c]
  //       pool_part[(slot * 2 + 1) * pool_odim + oc] = sum_t w[t, h] (x[t, c] - s1 / s0)^2,   s0 = sum_t w[t, h]
  //     with pool_w = the softmax output [rows, pool_heads]; split value (:151-157): h = c / pool_dvh, oc = c; else
  //     every head pools every channel, oc = h * N + c.  launch_att_pool_finalize merges the slots (weighted Chan).
  const float* pool_w = nullptr;
  int pool_heads = 0;
  int pool_split = 0;
  int pool_dvh = 0;
  int pool_odim = 0;
};

// fp32 MFMA (v_mfma_f32_32x32x2_f32) path.  aligned: ldx == cin (or K == cin), ldx % 4 == 0,
// K % 4 == 0, X 16-byte aligned and followed by >= 160 rows of readable slack.
hipError_t launch_gemm_f32(const GemmArgs& a, bool aligned, hipStream_t s);

// number of K slices launch_gemm_f32 will use for this shape (1 = no split); the caller provides
// `partial` of ksplit * M * Npad floats when > 1.
int gemm_f32_ksplit(int M, int Kpad, int Npad);

// Tail balancing of the bf16x3 kernel (1-D layers): when the last round of 128x128 tiles over the chip's 768
// workgroup slots would be nearly empty, its M tiles are split along K into `*splits` slices each (deterministic:
// raw partials + an ordered reduce).  Returns the partial workspace in bytes (0 = no tail handling).
// (cb_quant: a slice must hold a multiple of this many channel blocks -- the two-unit kernel takes them in pairs (5 taps) or quads (7))
int64_t gemm_bf16x3_tail_plan(int M, int Kpad, int Npad, int w, int* tail_mt, int* splits, int cb_quant = 1);

// fp32 frames -> split-blocked im2col rows for a small-cin first layer:
//   out row m, k < w*cin: x[(m + k / cin) * ldx + k % cin]; zero padded to ldsb columns.
hipError_t launch_im2col_sb(const float* x, int64_t ldx, int cin, int w, int64_t rows, void* out_sb, int ldsb, int f16,
                            int* ovf, hipStream_t s);

// bf16x3 / f16x3 split path (3x v_mfma_f32_16x16x32_{bf16,f16} per product tile).
hipError_t launch_gemm_bf16x3(const GemmArgs& a, hipStream_t s);
// two-unit split of the multi-tap convolutions (XV_PREC_F16F6: f16 hi*hi + two block-scaled fp6 cross terms), gemm_f16f6.hip
hipError_t launch_gemm_f16f6(const GemmArgs& a, hipStream_t s);
// reduce of a K-split tail whose slices hold raw sums (a.partial / a.tail_mt / a.ksplit in {2, 4, 8}; rows from M tile nMain on):
// BN scale / shift + activation, then the split-blocked rows or (a.ysb_f6) the two-unit block format
hipError_t launch_f6v2_tail_reduce(const GemmArgs& a, int nMain, hipStream_t s);
// split-blocked f16 rows -> the activation block format of gemm_f16f6.hip (same 128 bytes per (row, 32 channels))
hipError_t launch_f6_from_sb(const void* sb, void* out, int64_t rows, int nblk, hipStream_t s);

// rowmap for a valid convolution of width w over packed utterances:
//   in_off[b] = off0[b] - b*ctx_in  (rows of utterance b in the layer's input)
//   row r of utterance b at local frame t maps to out_off[b] + t if t < len_b - (w-1), else -1.
hipError_t launch_build_rowmap(const int32_t* off0, int B, int ctx_in, int w, int32_t* rowmap, int M,
                               hipStream_t s);

// statistics pooling (model/pooling.py:27-52):  out[b] = [mean_t x, sqrt(max-floor(var_t x))]
//   rows of utterance b: [off0[b] - b*ctx, off0[b+1] - (b+1)*ctx)
hipError_t launch_stat_pool(const float* x, int64_t ldx, int C, const int32_t* off0, int B, int ctx,
                            float* out, int64_t ldo, hipStream_t s);

// row -> utterance map for the fused pooling epilogue (rows of utterance b: [off0[b]-b*ctx, off0[b+1]-(b+1)*ctx))
hipError_t launch_build_row2utt(const int32_t* off0, int B, int ctx, int32_t* row2utt, int M, hipStream_t s);
// finalize of the fused statistics pooling: merge the per-segment (sum, M2) pairs of each utterance
hipError_t launch_pool_finalize(const float* part, int C, const int32_t* off0, int B, int ctx,
                                const int32_t* slotbase, float* out, int64_t ldo, hipStream_t s);

// ---- ResNet grid helpers (csrc/grid.hip).  A "grid" value holds, per utterance b, (L_b + 2) time rows of S positions
// of C channels with a zero border (S = F + 1, or F + 2 when a consumer reads it at frequency stride 2); utterance b
// starts at position (off0[b] + 2b) * S.
// rowmap of a conv whose GEMM rows enumerate (b, t', j): t' in [0, L_b + 2), j in [0, rows_per_t):
//   output bin iff t' < L_b and j < Fout, at position (off0[b]+2b)*So + (t'+1)*So + j + 1; the other rows fall on border
//   positions: encoded "write zeros" (-pos - 2) when cover (rows_per_t == So), else -1 (the caller zeroes the value)
hipError_t launch_build_rowmap_grid(const int32_t* off0, int B, int rows_per_t, int Fout, int So, int cover, int32_t* rowmap,
                                    int64_t M, hipStream_t s);
// the same when the convolution also strides time by 2 (off_in / off_out: frame offsets at the input / output time level;
// ktime 3 = 3x3 'same', 1 = 1x1 shortcut); rows that produce no output frame are skipped (-1)
hipError_t launch_build_rowmap_grid_ts(const int32_t* off_in, const int32_t* off_out, int B, int rows_per_t, int Fout, int So,
                                       int ktime, int32_t* rowmap, int64_t M, hipStream_t s);
// rowmap of conv5 (1 x F valid): rows enumerate padded time rows; valid iff 1 <= t' <= L_b -> frame off0[b]+t'-1
// compact enumeration of a grid convolution (rows = output bins only): arow[m] = window position, rowmap[m] = output position
hipError_t launch_build_rowmap_grid_compact(const int32_t* off_in, const int32_t* off_out, int B, int Sin, int So, int Fout,
                                            int sw, int st, int ktime, int32_t* arow, int32_t* rowmap, int64_t M, int64_t Mpad,
                                            hipStream_t s);
// zero the border positions (and one pitch behind the last row) of a grid value; chunks = 16-byte chunks per position
hipError_t launch_grid_zero_border(const int32_t* off0, int B, int64_t frames, int F, int S, int chunks_y, int chunks_sb,
                                   float* y, void* ysb, hipStream_t s);
hipError_t launch_build_rowmap_rows(const int32_t* off0, int B, int32_t* rowmap, int64_t M, hipStream_t s);
hipError_t launch_build_rowmap_trows(const int32_t* off0, int B, int So, int32_t* rowmap, int64_t M, hipStream_t s);
// rowmap of conv0 (rows = grid positions, pitch S): interior -> same position, border -> zeros at the same position
hipError_t launch_build_rowmap_interior(const int32_t* off0, int B, int F, int S, int32_t* rowmap, int64_t M, hipStream_t s);
// conv0 im2col: out SB row p (grid position of the OUTPUT), k = kh*3+kw < 9: x[t+kh-1][f+kw-1] or 0
hipError_t launch_im2col2d_sb(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int64_t P, void* out_sb,
                              int f16, int* ovf, hipStream_t s);
hipError_t launch_im2col2d_f32(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int64_t P, float* out,
                               hipStream_t s);
// 3x3 stride-1 'same' max-pool of a grid value [P, C] (fp32 in; fp32 and / or SB out, either may be null); borders -> 0
// conv0 of the ResNet (3x3 'same' on the 1-channel feature map, model/resnet.py:217-228) computed directly: fp32 FMAs, BN scale /
// shift + activation, fp32 and / or split-blocked grid output (pitch S, border positions zero).  wdir = [9][C] kernel, scale[C], shift[C].
hipError_t launch_conv0_direct(const float* x, int64_t ldx, const int32_t* off0, int B, int F, int S, int C, int64_t P,
                               const float* wdir, int act, const float* alpha, float* y, void* ysb, int ldsb, int f16, int* ovf,
                               float sb_mul, hipStream_t s);
hipError_t launch_grid_maxpool3x3(const float* x, const int32_t* off0, int B, int F, int S, int C, int64_t P, float* y,
                                  void* ysb, int ldsb, int f16, int* ovf, float sb_mul, hipStream_t s);
// grid [P, C] -> dense [sum L_b * F, C] (drops the zero border; test / endpoint output only)
hipError_t launch_grid_unpad_n(const float* grid, const int32_t* off0, int B, int F, int S, int C, int64_t frames, float* out,
                               hipStream_t s);

// sliding-window CMN + voiced-frame selection (csrc/frontend.hip); prefix: double [(frames + B) * dim] scratch
hipError_t launch_cmn_select(const float* x, int64_t ld, int dim, const int32_t* off, int B, double* prefix,
                             const int32_t* src, int64_t out_rows, int window, int center, int min_window, float* out,
                             hipStream_t s);

// post-step (csrc/post.hip): ivector-normalize-length / ivector-mean of run_extract_embeddings.sh:80-103
hipError_t launch_length_norm(const float* x, int64_t ldx, int64_t rows, int dim, int scaleup, float* y, int64_t ldy,
                              hipStream_t s);
hipError_t launch_speaker_mean(const float* x, int64_t ldx, int dim, const int32_t* spk_off, const int32_t* utt,
                               int64_t num_spk, float* out, int64_t ldo, hipStream_t s);

// attention scores (model/pooling.py:189-194): score[r, h] = scale * sum_d key[r, h*dk_h + d] * q[h, d]
// (split_key) or sum_d key[r, d] * q[h, d] (no split; dk_h == dk).
hipError_t launch_att_scores(const float* key, int64_t ldk, int64_t rows, const float* query, int H,
                             int dk_h, int split_key, float scale, float* scores, hipStream_t s);
// softmax over time per (utterance, head), in place on scores [rows, H]; also writes the
// [B, H, Lmax]-free packed layout weights_out[h * rows + r] when weights_out != nullptr.
hipError_t launch_att_softmax(float* scores, int H, const int32_t* off0, int B, int ctx, hipStream_t s);
// fused attention (csrc/xv_epilogue.h, attention epilogue of the bf16x3 GEMM):
// scores[r, h] = scale * sum_blk part[(blk * H + h) * ld + r], blocks in ascending order (deterministic)
hipError_t launch_att_scores_reduce(const float* part, int64_t ld, int nblk, int H, int64_t rows, float scale,
                                    float* scores, hipStream_t s);
// per-slot weight sums for the weighted Chan merge: s0[slot * H + h] = sum of weights[r, h] over the rows of utterance b
// inside 64-row tile t, slot = slotbase[b] + t (rows in ascending order)
hipError_t launch_att_slot_sums(const float* weights, int H, const int32_t* off0, int B, int ctx, const int32_t* slotbase,
                                float* s0, hipStream_t s);
// out[b] = [sum_s s1 ..., sqrt(max-floor(sum_s m2_s + s0_s (s1_s / s0_s - mean)^2)) ...] over the slots of utterance b
hipError_t launch_att_pool_finalize(const float* part, const float* s0, int odim, int H, int dvh, int split,
                                    const int32_t* off0, int B, int ctx, const int32_t* slotbase, float* out, int64_t ldo,
                                    hipStream_t s);

// weighted mean / std (model/pooling.py:201-218).  value [rows, dv]; out[b] = [mean(h,d)..., std(h,d)...]
hipError_t launch_att_pool(const float* value, int64_t ldv, int dv, const float* weights, int H,
                           int split_value, const int32_t* off0, int B, int ctx, float* out, int64_t ldo,
                           hipStream_t s);

// y = act(x * scale[c] + shift[c]) on [rows, C] (att_post_bn / att_post_relu, stage replays)
hipError_t launch_affine_act(const float* x, int64_t ldx, int64_t rows, int C, const float* scale,
                             const float* shift, const float* alpha, int act, float* y, int64_t ldy,
                             hipStream_t s);

// l2_scaling (model/common.py:45-58): y = x * s * rsqrt(max(sum x^2, 1e-12)) per row
hipError_t launch_l2_scale(const float* x, int64_t rows, int C, float factor, float* y, hipStream_t s);

// transposed copy [B,H,L]-style attention weights: out[b][h][t] from scores [rows,H] (uniform L only)
hipError_t launch_copy2d(const float* src, int64_t lds, float* dst, int64_t ldd, int64_t rows, int cols,
                         hipStream_t s);
// fp16 range flags (2 ints) -> device-accessible host memory, then cleared; one kernel
hipError_t launch_flags_snapshot(int* dev, int* host, hipStream_t s);
hipError_t launch_att_weights_out(const float* scores, int H, const int32_t* off0, int B, int ctx,
                                  float* out, hipStream_t s);

}  // namespace xv
