// C ABI of libxvec_hip.so (include/xvec_hip.h): model handle, batch plan and the op executor
// that strings the gfx950 kernels into the predict graph of the reference
// (model/tdnn.py:36-181, model/pooling.py:8-240, model/trainer.py:385-405, 886-913).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/xvec_hip.h"
#include "xv_kernels.h"

using namespace xv;

namespace {

constexpr double kBnEps = 1e-3;        // tf.layers.batch_normalization default epsilon
constexpr int kSlackRows = 512;        // readable rows after every activation buffer (GEMM tile / conv window overreach)
constexpr int kAlign = 256;

thread_local std::string g_last_error;  // failures with no handle (xv_create)

struct DevBuf {
  void* p = nullptr;
  size_t bytes = 0;
  hipError_t alloc(size_t n) {
    release();
    bytes = n;
    if (n == 0) return hipSuccess;
    return hipMalloc(&p, n);
  }
  void release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    bytes = 0;
  }
};

struct DeviceGuard {   // leave the caller's current device untouched
  int prev = -1;
  bool ok = true;
  explicit DeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { ok = false; return; }
    if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
  }
  ~DeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};

// ------------------------------------------------------------------------------ graph
enum Stage : int { ST_AFFINE = 0, ST_BN = 1, ST_ACT = 2 };

struct HostTensor {
  std::vector<int64_t> shape;
  std::vector<float> data;
  bool set = false;
};

// An affine layer: temporal convolution (w > 1) or dense (w == 1), optional BN, optional activation.
struct Layer {
  std::string kernel_name, bias_name, bn_scope, alpha_name;
  std::string ep[3];          // endpoint key per stage ("" when the stage does not exist)
  int w = 1, cin = 0, cout = 0;
  bool has_bn = false;
  int act = ACT_NONE;         // activation of the final stage
  // device
  DevBuf wt;                  // fp32 [Npad][Kpad]                      (fp32 MFMA kernel)
  DevBuf wfr;                 // same values, MFMA-fragment-major: [Npad/32][Kpad/32][plane*2+channel tile][64 lanes][16 B]
  DevBuf wsb;                 // split-blocked bf16 hi/lo [Npad][Kpad/32][128 B] (bf16x3 kernel)
  bool use_split = false;     // this layer runs on the bf16x3 kernel
  bool im2col = false;        // first layer in a split mode: its input is the caller's fp32 feature matrix, staged per forward
  int cin_pad = 0;            // > 0: staged as SB rows of cin_pad (= cin rounded up to 32) channels per frame and convolved
                              // like any other w-tap layer (slab reuse across the taps); 0 with im2col: rows materialised
  // ResNet 2-D convolutions on the zero-bordered grid (csrc/grid.hip): 0 = 1-D conv / dense,
  // 1 = 3x3 'same' stride (1,sw), 2 = 1x1 shortcut stride (1,sw), 3 = conv5 (1 x Fin, valid), 4 = conv0 (cin 1)
  int mode = 0;
  int Fin = 0, Fout = 0, sw = 1;
  int st = 1;                 // stride along time of a grid convolution (2: first block of stages 2-4 under resnet_time_stride)
  bool has_bias = true;
  int K() const { return mode == 1 ? 9 * cin : (mode == 3 ? Fin * cin : (mode == 4 ? 9 : w * cin)); }
  DevBuf vec;                 // [bias | bn_scale | bn_shift | alpha | ones] each cout floats
  int in_exp = 0, out_exp = 0; // fp16 split formats: the split copy of the input / output holds value * 2^exp (act_exponent below)
  bool use_f6 = false;        // XV_PREC_F16F6: this multi-tap convolution runs on gemm_f16f6_kernel (input converted to its block format)
  DevBuf wf6m, wf6x;          // its weights: f16 main fragments [N/32][cin/32][8 taps][2][64 lanes][16 B]; fp6 cross operands (gemm_f16f6.hip)
  DevBuf wdir;                // conv0 (mode 4): fp32 [9][cout] kernel, then bn_scale[cout], bn_shift[cout] (direct kernel, csrc/grid.hip)
  int Kpad = 0, Npad = 0;
  int final_stage() const { return act != ACT_NONE ? ST_ACT : (has_bn ? ST_BN : ST_AFFINE); }
  const float* d_bias() const { return static_cast<const float*>(vec.p); }
  const float* d_scale() const { return d_bias() + cout; }
  const float* d_shift() const { return d_bias() + 2 * cout; }
  const float* d_alpha() const { return alpha_name.empty() ? nullptr : d_bias() + 3 * cout; }
  const float* d_ones() const { return d_bias() + 4 * cout; }
};

enum OpKind : int {
  OP_GEMM = 0,        // layer
  OP_STAT_POOL,       // statistics pooling
  OP_ATT_SCORES,      // key . query
  OP_ATT_SOFTMAX,     // in place on the scores buffer (modelled as its own value)
  OP_ATT_POOL,        // weighted mean/std
  OP_AFFINE_ACT,      // att_post_bn / att_post_relu
  OP_L2_SCALE,        // endpoints["output"] with feature_norm
  OP_GRID_MAXPOOL     // 3x3 'same' max-pool on a grid value (resnet_maxpooling)
};

// A value is a matrix produced by an op (or the network input, value 0).
struct Value {
  int grid_F = 0;     // > 0: zero-bordered grid value with grid_F frequency bins (rows = (F0+2B)*grid_S)
  int grid_S = 0;     // its pitch: positions per padded time row (csrc/grid.hip: F + 1, or F + 2 under a stride-2 reader)
  bool frame_level = true;
  int ctx = 0;        // temporal context consumed (frame-level values): rows = F[tlevel] - B*ctx
  int tlevel = 0;     // time resolution: utterance b has ceil(L_b / 2^tlevel) frames (resnet_time_stride; else 0)
  int cols = 0;
  int sb_exp = 0;     // fp16 split formats: the split-blocked copy of this value holds value * 2^sb_exp
};

struct Op {
  int kind = OP_GEMM;
  int layer = -1;     // OP_GEMM
  int in0 = -1, in1 = -1;   // value ids
  int out = -1;       // value id
  int bnvec = -1;     // OP_AFFINE_ACT: index into Model::post vectors
};

struct Node {         // an endpoints[...] key
  std::string name;
  int op = -1;        // producing op
  int stage = -1;     // OP_GEMM: stage to emit; OP_AFFINE_ACT: 1 = bn only, 2 = bn + act
  bool att_weights = false;
};

}  // namespace

struct xv_handle {
  xv_model_desc desc{};
  int device = 0;
  bool finalized = false;
  std::string err;
  std::mutex mu;                                      // graph / weights / options
  std::mutex err_mu;                                  // h->err (any thread may fail)
  std::mutex prof_mu;                                 // profiling records (xv_forward from several threads)
  // options (xv_set_option)
  int opt_pool_fusion = 1;                            // statistics pooling fused into the last frame layer's epilogue
  int opt_tail_split = 1;                             // K-split of the last, nearly empty round of GEMM tiles
  int opt_onetap_f6_notail = 1;                       // (A/B switch of the note in xv_plan_create)
  int opt_grid_f6 = 1;                                // XV_PREC_F16F6: the stride-1 3 x 3 ResNet convolutions of >= 128 channels on the two-unit kernel
  int opt_slab3 = 1;                                  // one-tap GEMM layers on the three-slab-buffer kernel
  int opt_grid_compact = 1;                           // ResNet grid convolutions enumerate output bins only (split precisions)
  int opt_att_fusion = 1;                             // attention scores / weighted moments in the GEMM epilogues
  int opt_profile_dominant = 0;                       // xv_profile_*: bracket only the step with the most FLOPs of a plan
  // device index arrays of destroyed plans, kept for the next plan (no hipMalloc / hipFree per ragged batch)
  std::mutex pool_mu;
  std::vector<DevBuf> pool;
  size_t pool_bytes = 0;
  std::map<std::string, HostTensor> tensors;          // expected variables
  std::vector<Layer> layers;
  std::vector<Value> values;
  std::vector<Op> ops;
  std::vector<Node> nodes;
  // attention extras
  DevBuf query;                 // [H, dk_h]
  DevBuf ovf_flag;              // fp16 split formats, 4 ints: [0] set by any kernel that converted a value beyond the fp16 range;
                                // [1] bits of the largest feature magnitude staged since the last reset (underflow guard)
  DevBuf query_eff;             // [H, Npad of the last key layer]: the query of head h over the padded key width, zero
                                // outside the head's slice (fused score epilogue)
  int key_npad = 0;
  int att_dk_h = 0, att_dk = 0, att_dv = 0;
  int final_ctx = 14;           // temporal context of the pooled frames (tdnn 14, etdnn 22)
  std::string post_bn_scope, post_alpha_name;
  DevBuf post_vec;              // [scale | shift | alpha] each pool_dim floats
  int pool_dim = 0;
  // optional per-op event profiling (xv_profile_begin / xv_profile_end)
  struct ProfRec { hipEvent_t e0, e1; const xv_plan* plan; int step; };
  bool profiling = false;
  std::vector<hipEvent_t> prof_pool;
  size_t prof_next = 0;
  std::vector<ProfRec> prof_recs;
  int prof_forwards = 0;
};

struct PlanStep {
  int op = -1;
  int stage = -1;               // stage override for the target op, else the op's final stage
  bool to_out = false;          // writes the user's output buffer
  int64_t out_off = -1;         // workspace byte offset of the fp32 output (-1: none / user buffer)
  int64_t out_sb_off = -1;      // workspace byte offset of the split-blocked output (-1: none)
  int64_t in0_off = -1, in1_off = -1;   // fp32 inputs; -2 = network input
  int64_t in0_sb_off = -1;      // split-blocked input
  int64_t rows_in = 0, rows_out = 0;
  int M = 0;                    // GEMM rows to compute
  int rowmap = -1;              // index into plan rowmaps (conv layers)
  int64_t scratch_off = -1;     // per-step scratch (im2col rows / split-K partials), released after the step
  int64_t scratch2_off = -1;    // two-unit layers: the K-split partials of the tail tiles (scratch_off holds the converted input)
  int ksplit = 1;               // split-K slices of a small-M fp32 GEMM, or of the tail M tiles of a bf16x3 GEMM
  int tail_mt = 0;              // bf16x3: M tiles computed K-split (gemm_bf16x3_tail_plan)
  bool fuse_pool = false;       // GEMM: emit pooling partials instead of activations; STAT_POOL: finalize only
  int lvl_in = 0, lvl_out = 0;  // time level of the input / output value
  int64_t frames_out = 0;       // total frames of the batch at the output's time level
  bool grid_cover = false;      // grid-valued output whose border is re-zeroed by zero-writing GEMM rows (no memset)
  bool out_f6 = false;          // XV_PREC_F16F6: this layer writes its split-blocked output in the block format of gemm_f16f6.hip ...
  bool trows = false;           // two-unit form of a stride-1 3 x 3 grid convolution: GEMM rows = padded time rows x frequency bins
  bool in_f6 = false;           // ... because its only reader is this kind of layer, which then needs no conversion pass
  bool compact = false;         // grid convolution whose GEMM rows are the output bins only (csrc/grid.hip, compact form)
  int arow = -1;                // compact: index into plan rowmaps of the window positions (GemmArgs::arow)
  int fuse_att = 0;             // GEMM: 1 = score partials instead of the key, 2 = weighted moments instead of the value;
                                // ATT_SCORES / ATT_SOFTMAX / ATT_POOL: 1 = the fused form of that op
  int64_t att_w_off = -1;       // fuse_att 2: workspace offset of the softmax output (weights [rows, H])
  int64_t att_s0_off = -1;      // workspace offset of the per-slot weight sums [pool_slots, H]
  int64_t att_ld = 0;           // fuse_att 1: row stride of the partial-score planes
  bool unpad_to_out = false;    // grid-valued target node: GEMM writes the padded grid, then it is unpadded into `out`
  int64_t flops = 0, bytes = 0;
};

struct xv_plan {
  xv_handle* h = nullptr;
  xv_plan_info info{};
  std::vector<int32_t> offsets;     // host copy
  std::vector<int32_t> offsets_slotbase;
  DevBuf d_offsets;                 // [B+1]
  std::vector<int32_t> lvl_offsets[4];   // frame offsets per time level (level 0 = offsets); levels > 0 only under resnet_time_stride
  DevBuf d_lvl[4];                  // device copies of levels 1..3 ([0] unused: level 0 is d_offsets)
  const int32_t* dev_offsets(int level) const {
    return static_cast<const int32_t*>(level > 0 ? d_lvl[level].p : d_offsets.p);
  }
  DevBuf d_rowmaps;                 // concatenated row maps
  std::vector<int64_t> rowmap_off;  // element offsets into d_rowmaps
  std::vector<PlanStep> steps;
  DevBuf d_row2utt;                 // fused pooling: utterance of each pooled row
  DevBuf d_slotbase;                // fused pooling: [B] slot base per utterance
  int64_t pool_slots = 0;
  bool uniform_len = true;
  int uniform_L = 0;
  int dominant_step = 0;            // index of the step with the most algorithmic FLOPs
};

namespace {

int fail(xv_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (h) {
    std::lock_guard<std::mutex> lk(h->err_mu);
    h->err = buf;
  }
  g_last_error = buf;
  return code;
}

#define XV_HIP(h, expr)                                                                              \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return fail((h), XV_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                  __LINE__);                                                                         \
  } while (0)

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

// Plan index arrays come from / go back to a per-handle pool: a ragged ark stream creates one plan per batch, and
// hipMalloc / hipFree (device-synchronising) per batch would serialise the host with the GPU.
constexpr size_t kPoolMaxBuffers = 96, kPoolMaxBytes = (size_t)1 << 30;

hipError_t pool_take(xv_handle* h, size_t bytes, DevBuf& out) {
  out = DevBuf();
  if (bytes == 0) return hipSuccess;
  {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    int best = -1;
    for (size_t i = 0; i < h->pool.size(); ++i)
      if (h->pool[i].bytes >= bytes && h->pool[i].bytes <= 4 * bytes + 4096 &&
          (best < 0 || h->pool[i].bytes < h->pool[best].bytes))
        best = (int)i;
    if (best >= 0) {
      out = h->pool[best];
      h->pool_bytes -= out.bytes;
      h->pool.erase(h->pool.begin() + best);
      return hipSuccess;
    }
  }
  const size_t cap = (bytes + 4095) / 4096 * 4096;
  const hipError_t e = hipMalloc(&out.p, cap);
  if (e == hipSuccess) out.bytes = cap; else out = DevBuf();
  return e;
}

void pool_give(xv_handle* h, DevBuf& b) {
  if (!b.p) return;
  {
    std::lock_guard<std::mutex> lk(h->pool_mu);
    if (h->pool.size() < kPoolMaxBuffers && h->pool_bytes + b.bytes <= kPoolMaxBytes) {
      h->pool.push_back(b);
      h->pool_bytes += b.bytes;
      b = DevBuf();
      return;
    }
  }
  b.release();
}

uint16_t f32_to_bf16_rn(float f) {   // round to nearest even; inputs are finite weights
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
float bf16_to_f32(uint16_t b) {
  uint32_t u = (uint32_t)b << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
uint16_t f32_to_f16_rn(float f) {    // IEEE binary16, round to nearest even (subnormals kept, overflow -> inf)
  const _Float16 h = (_Float16)f;
  uint16_t u;
  memcpy(&u, &h, 2);
  return u;
}
float f16_to_f32(uint16_t u) {
  _Float16 h;
  memcpy(&h, &u, 2);
  return (float)h;
}

void expect(xv_handle* h, const std::string& name, std::vector<int64_t> shape) {
  HostTensor t;
  t.shape = std::move(shape);
  h->tensors[name] = std::move(t);
}

void expect_bn(xv_handle* h, const std::string& scope, int n) {
  for (const char* s : {"gamma", "beta", "moving_mean", "moving_variance"}) expect(h, scope + "/" + s, {n});
}

int act_of(const xv_model_desc& d) {
  return d.relu_type == XV_ACT_PRELU ? ACT_PRELU : (d.relu_type == XV_ACT_LRELU ? ACT_LRELU : ACT_RELU);
}

// adds layer + op + value + nodes; returns the output value id
int add_layer(xv_handle* h, const std::string& var_scope, const std::string& ep_prefix, bool conv, int w,
              int cin, int cout, bool has_bn, int act, int in_value, bool frame_level, int ctx_out,
              int kernel_rank = 4) {
  Layer L;
  const std::string kind = conv ? "_conv" : "_dense";
  L.kernel_name = var_scope + kind + "/kernel";
  L.bias_name = var_scope + kind + "/bias";
  L.w = w; L.cin = cin; L.cout = cout; L.has_bn = has_bn; L.act = act;
  if (conv && kernel_rank == 4) expect(h, L.kernel_name, {1, w, cin, cout});      // tf.layers.conv2d (1,w): HWIO
  else if (conv) expect(h, L.kernel_name, {w, cin, cout});                         // tf.layers.conv1d
  else expect(h, L.kernel_name, {cin, cout});
  expect(h, L.bias_name, {cout});
  L.ep[ST_AFFINE] = ep_prefix + kind;
  if (has_bn) {
    L.bn_scope = var_scope + "_bn";
    expect_bn(h, L.bn_scope, cout);
    L.ep[ST_BN] = ep_prefix + "_bn";
  }
  if (act != ACT_NONE) {
    L.ep[ST_ACT] = ep_prefix + (act == ACT_TANH ? "_tanh" : "_relu");
    if (act == ACT_PRELU) {
      L.alpha_name = var_scope + "_relu/alpha";
      expect(h, L.alpha_name, {cout});
    }
  }
  const int li = (int)h->layers.size();
  h->layers.push_back(std::move(L));
  Value v; v.frame_level = frame_level; v.ctx = ctx_out; v.cols = cout;
  const int vid = (int)h->values.size();
  h->values.push_back(v);
  Op op; op.kind = OP_GEMM; op.layer = li; op.in0 = in_value; op.out = vid;
  const int oi = (int)h->ops.size();
  h->ops.push_back(op);
  for (int s = 0; s < 3; ++s) {
    const std::string& e = h->layers[li].ep[s];
    if (!e.empty()) { Node n; n.name = e; n.op = oi; n.stage = s; h->nodes.push_back(n); }
  }
  return vid;
}

int add_simple_op(xv_handle* h, int kind, int in0, int in1, bool frame_level, int ctx, int cols) {
  Value v; v.frame_level = frame_level; v.ctx = ctx; v.cols = cols;
  const int vid = (int)h->values.size();
  h->values.push_back(v);
  Op op; op.kind = kind; op.in0 = in0; op.in1 = in1; op.out = vid;
  h->ops.push_back(op);
  return vid;
}

void add_node(xv_handle* h, const std::string& name, int op, int stage, bool attw = false) {
  Node n; n.name = name; n.op = op; n.stage = stage; n.att_weights = attw;
  h->nodes.push_back(n);
}

// one ResNet convolution (+BN, + optional activation / residual); returns the output value id
int add_conv2d(xv_handle* h, const std::string& var, const std::string& bn, const std::string& relu, int mode, int cin,
               int cout, int Fin, int Fout, int sw, int act, int in_value, int residual_value, const char* node_name,
               int st = 1) {
  Layer L;
  L.mode = mode; L.cin = cin; L.cout = cout; L.Fin = Fin; L.Fout = Fout; L.sw = sw; L.w = 1; L.st = st;
  L.kernel_name = var + "/kernel";
  L.has_bias = (mode == 3);
  if (mode == 1 || mode == 4) expect(h, L.kernel_name, {3, 3, cin, cout});
  else if (mode == 2) expect(h, L.kernel_name, {1, 1, cin, cout});
  else expect(h, L.kernel_name, {1, Fin, cin, cout});
  if (L.has_bias) { L.bias_name = var + "/bias"; expect(h, L.bias_name, {cout}); }
  L.has_bn = true;
  L.bn_scope = bn;
  expect_bn(h, bn, cout);
  L.act = act;
  if (act == ACT_PRELU) { L.alpha_name = relu + "/alpha"; expect(h, L.alpha_name, {cout}); }
  L.ep[act != ACT_NONE ? ST_ACT : ST_BN] = node_name ? node_name : "";
  const int li = (int)h->layers.size();
  h->layers.push_back(std::move(L));
  Value v; v.cols = cout;
  v.tlevel = h->values[in_value].tlevel + (st == 2 ? 1 : 0);
  if (mode == 3) { v.frame_level = true; v.ctx = 0; } else { v.grid_F = Fout; }
  const int vid = (int)h->values.size();
  h->values.push_back(v);
  Op op; op.kind = OP_GEMM; op.layer = li; op.in0 = in_value; op.in1 = residual_value; op.out = vid;
  h->ops.push_back(op);
  if (node_name) add_node(h, node_name, (int)h->ops.size() - 1, act != ACT_NONE ? ST_ACT : ST_BN);
  return vid;
}

// model/resnet.py:152-351 (resnet_18): conv0, four stages of [conv_block, identity_block...], conv5,
// dense1, dense2, pooling, tdnn6, tdnn7.  `channels` = width of stage 1 (64 in the reference).
// Block outputs are exposed as extra nodes (conv0_relu, conv1a, ..., conv5_relu, dense*_relu) for
// tests only; the reference registers just pooling / tdnn6_* / tdnn7_* / output.
int build_resnet(xv_handle* h) {
  const xv_model_desc& d = h->desc;
  const int act = act_of(d);
  const std::string sc = "resnet_18/";
  if (d.feat_dim != 40) return fail(h, XV_ERR_INVALID, "resnet_18 needs 40-dim features (model/resnet.py:190)");
  if (d.pooling_type != XV_POOL_STATISTICS)
    return fail(h, XV_ERR_UNSUPPORTED, "resnet_18 registers no frame-level endpoints: only statistics_pooling is possible");
  h->values.clear();
  Value in; in.frame_level = true; in.ctx = 0; in.cols = 40;
  h->values.push_back(in);
  int F = 40, cin = d.channels;
  int v = add_conv2d(h, sc + "conv0_1", sc + "conv0_bn", sc + "conv0_relu", 4, 1, cin, F, F, 1, act, 0, -1, "conv0_relu");
  if (d.resnet_maxpooling) {                       // model/resnet.py:230-231
    Value pv; pv.grid_F = F; pv.cols = cin;
    const int vid = (int)h->values.size();
    h->values.push_back(pv);
    Op op; op.kind = OP_GRID_MAXPOOL; op.in0 = v; op.out = vid;
    h->ops.push_back(op);
    add_node(h, "conv0_max", (int)h->ops.size() - 1, -1);
    v = vid;
  }
  for (int stage = 1; stage <= 4; ++stage) {
    const int nf = d.channels << (stage - 1);
    const int sw = stage == 1 ? 1 : 2;
    const int Fo = F / sw;
    for (int bi = 0; bi < d.resnet_blocks[stage - 1]; ++bi) {
      char nm[32];
      if (bi == 0) snprintf(nm, sizeof(nm), "conv%da", stage); else snprintf(nm, sizeof(nm), "conv%db_%d", stage, bi - 1);
      const std::string b = sc + nm;
      const int s_w = bi == 0 ? sw : 1, Fi = bi == 0 ? F : Fo;
      const int s_t = (bi == 0 && stage > 1 && d.resnet_time_stride) ? 2 : 1;       // model/resnet.py:187,239,244,249
      const int c0 = add_conv2d(h, b + "_conv0", b + "_bn0", b + "_relu0", 1, cin, nf, Fi, Fo, s_w, act, v, -1, nullptr, s_t);
      int shortcut = v;
      if (bi == 0)      // projection shortcut: 1x1 conv + BN (model/resnet.py:71-83)
        shortcut = add_conv2d(h, b + "_conv_short", b + "_bn_short", "", 2, cin, nf, Fi, Fo, s_w, ACT_NONE, v, -1, nullptr, s_t);
      v = add_conv2d(h, b + "_conv1", b + "_bn1", b + "_relu_final", 1, nf, nf, Fo, Fo, 1, act, c0, shortcut, nm);
      cin = nf;
    }
    F = Fo;
  }
  v = add_conv2d(h, sc + "conv5", sc + "conv5_bn", sc + "conv5_relu", 3, cin, cin, F, 1, 1, act, v, -1, "conv5_relu");
  h->final_ctx = 0;
  // dense1 / dense2 (model/resnet.py:270-290): variables "<name>/kernel", BN "<name>_bn", PReLU "<name>_relu/alpha"
  auto dense_named = [&](const char* name, int ci, int co, int in_v) {
    Layer L;
    L.kernel_name = sc + name + "/kernel"; L.bias_name = sc + name + "/bias";
    L.w = 1; L.cin = ci; L.cout = co; L.has_bn = true; L.act = act;
    expect(h, L.kernel_name, {ci, co}); expect(h, L.bias_name, {co});
    L.bn_scope = sc + name + "_bn"; expect_bn(h, L.bn_scope, co);
    if (act == ACT_PRELU) { L.alpha_name = sc + name + "_relu/alpha"; expect(h, L.alpha_name, {co}); }
    L.ep[ST_ACT] = std::string(name) + "_relu";
    const int li = (int)h->layers.size();
    h->layers.push_back(std::move(L));
    Value vv; vv.frame_level = true; vv.ctx = 0; vv.cols = co; vv.tlevel = h->values[in_v].tlevel;
    const int vid = (int)h->values.size();
    h->values.push_back(vv);
    Op op; op.kind = OP_GEMM; op.layer = li; op.in0 = in_v; op.out = vid;
    h->ops.push_back(op);
    add_node(h, std::string(name) + "_relu", (int)h->ops.size() - 1, ST_ACT);
    return vid;
  };
  v = dense_named("dense1", cin, cin, v);
  v = dense_named("dense2", cin, d.num_nodes_pooling_layer, v);
  h->pool_dim = 2 * d.num_nodes_pooling_layer;
  const int pooled = add_simple_op(h, OP_STAT_POOL, v, -1, false, 0, h->pool_dim);
  add_node(h, "pooling", (int)h->ops.size() - 1, -1);
  v = add_layer(h, sc + "tdnn6", "tdnn6", false, 1, h->pool_dim, cin, true, act, pooled, false, 0);
  v = add_layer(h, sc + "tdnn7", "tdnn7", false, 1, cin, d.num_nodes_last_layer, !d.last_layer_no_bn,
                d.last_layer_linear ? ACT_NONE : act, v, false, 0);
  if (d.feature_norm) {
    add_simple_op(h, OP_L2_SCALE, v, -1, false, 0, d.num_nodes_last_layer);
    add_node(h, "output", (int)h->ops.size() - 1, -1);
  } else {
    const int last = (int)h->ops.size() - 1;
    add_node(h, "output", last, h->layers[h->ops[last].layer].final_stage());
  }
  // Pitch of the grid values, one per stage (= per number of frequency bins, so that a layer's input, output and
  // residual share it): F + 1 (shared border column) unless a value of the stage is read at frequency stride 2, which
  // needs an even pitch -> F + 2.  With the default blocks: 42, 22, 12 for stages 1-3 and 6 for stage 4.
  std::vector<int> wide;
  for (const Op& op : h->ops) {
    if (op.kind != OP_GEMM) continue;
    const Layer& L = h->layers[op.layer];
    if ((L.mode == 1 || L.mode == 2) && L.sw == 2 && op.in0 > 0) wide.push_back(h->values[op.in0].grid_F);
  }
  for (Value& v : h->values)
    if (v.grid_F > 0) v.grid_S = v.grid_F + (std::find(wide.begin(), wide.end(), v.grid_F) != wide.end() ? 2 : 1);
  return XV_OK;
}

// Build the predict graph for `desc`: model/tdnn.py:36-181 (tdnn) or :343-591 (etdnn).
int build_graph(xv_handle* h) {
  const xv_model_desc& d = h->desc;
  if (d.network_type == XV_NET_RESNET18) return build_resnet(h);
  const int C = d.channels, act = act_of(d);
  h->values.clear();
  Value in; in.frame_level = true; in.ctx = 0; in.cols = d.feat_dim;
  h->values.push_back(in);   // value 0 = network input
  // frame-level layer table: kernel width per layer (1 = dense); the last one feeds the pooling
  static const int kTdnn[] = {5, 5, 7, 1, 1};
  static const int kEtdnn[] = {5, 1, 5, 1, 7, 1, 9, 1, 1, 1};
  const bool et = d.network_type == XV_NET_ETDNN;
  const int* widths = et ? kEtdnn : kTdnn;
  const int nframe = et ? 10 : 5;
  const std::string scope = et ? "etdnn/" : "tdnn/";
  std::vector<int> frame_value(nframe + 1, -1);   // value id of layer i's output (1-based)
  int v = 0, ctx = 0, cin = d.feat_dim;
  for (int i = 1; i <= nframe; ++i) {
    const int w = widths[i - 1];
    ctx += w - 1;
    const int cout = (i == nframe) ? d.num_nodes_pooling_layer : C;
    char nm[16];
    snprintf(nm, sizeof(nm), "tdnn%d", i);
    v = add_layer(h, scope + nm, nm, w > 1, w, cin, cout, true, act, v, true, ctx, et ? 3 : 4);
    frame_value[i] = v;
    cin = cout;
  }
  h->final_ctx = ctx;
  const int v_last = v;

  int pooled;
  if (d.pooling_type == XV_POOL_STATISTICS) {
    h->pool_dim = 2 * d.num_nodes_pooling_layer;
    pooled = add_simple_op(h, OP_STAT_POOL, v_last, -1, false, 0, h->pool_dim);
    add_node(h, "pooling", (int)h->ops.size() - 1, -1);
  } else if (d.pooling_type == XV_POOL_SELF_ATTENTION) {
    auto pick = [&](int which) {     // a frame layer whose output already has the full temporal context
      if (which < 1 || which > nframe) return -1;
      return h->values[frame_value[which]].ctx == ctx ? frame_value[which] : -1;
    };
    int key = pick(d.att_key_input), val = pick(d.att_value_input);
    if (key < 0 || val < 0)
      return fail(h, XV_ERR_UNSUPPORTED, "att_key_input/att_value_input must be a tdnn<N>_relu at full temporal context");
    if (d.att_num_key_layers < 1 || d.att_num_key_layers > XV_MAX_ATT_LAYERS || d.att_num_value_layers < 0 ||
        d.att_num_value_layers > XV_MAX_ATT_LAYERS)
      return fail(h, XV_ERR_INVALID, "attention: bad number of key/value layers");
    auto kind_to = [&](int kind, bool& bn, int& a) {
      bn = kind == 2;
      a = (kind == 1 || kind == 2) ? act : (kind == 3 ? ACT_TANH : ACT_NONE);
    };
    const std::string base = scope + "attention/";
    for (int i = 0; i < d.att_num_key_layers; ++i) {                       // model/pooling.py:100-116
      bool bn; int a;
      kind_to(i < d.att_num_key_layers - 1 ? 2 : d.att_key_network_type, bn, a);
      char nm[32]; snprintf(nm, sizeof(nm), "att_key%d", i);
      key = add_layer(h, base + nm + "/" + nm, nm, false, 1, h->values[key].cols, d.att_key_num_nodes[i], bn, a, key, true, ctx);
    }
    for (int i = 0; i < d.att_num_value_layers; ++i) {                     // model/pooling.py:119-135
      bool bn; int a;
      kind_to(i < d.att_num_value_layers - 1 ? 2 : d.att_value_network_type, bn, a);
      char nm[32]; snprintf(nm, sizeof(nm), "att_value%d", i);
      val = add_layer(h, base + nm + "/" + nm, nm, false, 1, h->values[val].cols, d.att_value_num_nodes[i], bn, a, val, true, ctx);
    }
    const int H = d.att_num_heads;
    h->att_dk = h->values[key].cols;
    h->att_dv = h->values[val].cols;
    if (H < 1) return fail(h, XV_ERR_INVALID, "att_num_heads must be >= 1");
    if (d.att_split_key && h->att_dk % H) return fail(h, XV_ERR_INVALID, "key dim %d not divisible by %d heads", h->att_dk, H);
    if (d.att_split_value && h->att_dv % H) return fail(h, XV_ERR_INVALID, "value dim %d not divisible by %d heads", h->att_dv, H);
    h->att_dk_h = d.att_split_key ? h->att_dk / H : h->att_dk;
    expect(h, scope + "attention/query", {H, h->att_dk_h});
    const int sc = add_simple_op(h, OP_ATT_SCORES, key, -1, true, ctx, H);
    const int sm = add_simple_op(h, OP_ATT_SOFTMAX, sc, -1, true, ctx, H);
    add_node(h, "attention_weights", (int)h->ops.size() - 1, -1, true);
    h->pool_dim = 2 * (d.att_split_value ? h->att_dv : h->att_dv * H);
    pooled = add_simple_op(h, OP_ATT_POOL, val, sm, false, 0, h->pool_dim);
    add_node(h, "att_output_before_nonlinear", (int)h->ops.size() - 1, -1);
    if (d.att_apply_nonlinear) {                                           // model/pooling.py:222-229
      h->post_bn_scope = scope + "attention/att_post_bn";
      expect_bn(h, h->post_bn_scope, h->pool_dim);
      if (act == ACT_PRELU) {
        h->post_alpha_name = scope + "attention/att_post_relu/alpha";
        expect(h, h->post_alpha_name, {h->pool_dim});
      }
      pooled = add_simple_op(h, OP_AFFINE_ACT, pooled, -1, false, 0, h->pool_dim);
      add_node(h, "att_post_bn", (int)h->ops.size() - 1, 1);
      add_node(h, "att_post_relu", (int)h->ops.size() - 1, 2);
    }
    add_node(h, "pooling", (int)h->ops.size() - 1, d.att_apply_nonlinear ? 2 : -1);
  } else {
    return fail(h, XV_ERR_UNSUPPORTED, "Not implement pooling_type %d", d.pooling_type);
  }
  // segment-level layers: tdnn6/tdnn7 (model/tdnn.py:137-179) or tdnn12/tdnn13 (:547-589)
  char s1[16], s2[16];
  snprintf(s1, sizeof(s1), "tdnn%d", et ? 12 : 6);
  snprintf(s2, sizeof(s2), "tdnn%d", et ? 13 : 7);
  v = add_layer(h, scope + s1, s1, false, 1, h->pool_dim, C, true, act, pooled, false, 0);
  v = add_layer(h, scope + s2, s2, false, 1, C, d.num_nodes_last_layer, !d.last_layer_no_bn,
                d.last_layer_linear ? ACT_NONE : act, v, false, 0);
  if (d.feature_norm) {                                                     // model/trainer.py:400-403
    add_simple_op(h, OP_L2_SCALE, v, -1, false, 0, d.num_nodes_last_layer);
    add_node(h, "output", (int)h->ops.size() - 1, -1);
  } else {
    const int last = (int)h->ops.size() - 1;
    add_node(h, "output", last, h->layers[h->ops[last].layer].final_stage());
  }
  return XV_OK;
}

const HostTensor& T(const xv_handle* h, const std::string& n) { return h->tensors.at(n); }

// s = gamma / sqrt(var + eps), t = beta - mean * s   (inference BN as one multiply-add)
void bn_fold(const xv_handle* h, const std::string& scope, int n, std::vector<double>& s, std::vector<double>& t) {
  const auto& g = T(h, scope + "/gamma").data;
  const auto& b = T(h, scope + "/beta").data;
  const auto& m = T(h, scope + "/moving_mean").data;
  const auto& v = T(h, scope + "/moving_variance").data;
  s.resize(n); t.resize(n);
  for (int i = 0; i < n; ++i) {
    s[i] = (double)g[i] / std::sqrt((double)v[i] + kBnEps);
    t[i] = (double)b[i] - (double)m[i] * s[i];
  }
}

// ---- block-scaled fp6 (e2m3) quantisation of 32 values: the host twin of e2m3_code / e8m0_of in csrc/gemm_f16f6.hip
uint32_t host_e2m3(float x, float inv) {
  float v = std::fmin(std::fabs(x) * inv, 7.5f);
  const bool sub = v < 1.f;
  const float t = sub ? v + 1.f : v;
  uint32_t bits;
  memcpy(&bits, &t, 4);
  uint32_t c = ((bits + 0x80000u) >> 20) - (126u << 3) - (sub ? 8u : 0u);
  c = c > 31u ? 31u : c;
  return c | (x < 0.f ? 32u : 0u);
}
void host_quant32(const float* v, unsigned char* codes24, unsigned char* scale_byte) {
  float amax = 0.f;
  for (int i = 0; i < 32; ++i) amax = std::fmax(amax, std::fabs(v[i]));
  float inv = 1.f;
  uint32_t byte = 0;
  if (amax > 0.f && std::isfinite(amax)) {
    const float r = amax * (1.f / 7.5f);
    uint32_t bits;
    memcpy(&bits, &r, 4);
    int e = (int)((bits + 0x7FFFFFu) >> 23) - 127;
    e = e < -126 ? -126 : (e > 126 ? 126 : e);
    byte = (uint32_t)(127 + e);
    const uint32_t ib = (uint32_t)(127 - e) << 23;
    memcpy(&inv, &ib, 4);
  }
  memset(codes24, 0, 24);
  for (int i = 0; i < 32; ++i) {
    const uint32_t c = host_e2m3(v[i], inv);
    const int bit = 6 * i;
    for (int q = 0; q < 6; ++q)
      if ((c >> q) & 1) codes24[(bit + q) >> 3] |= (unsigned char)(1u << ((bit + q) & 7));
  }
  *scale_byte = (unsigned char)byte;
}

// fp16 split formats (XV_PREC_F16X3 / F16F6): power of two the split copy of a layer's output is kept at.  hi + lo carries 22
// significand bits only while the low half is a normal fp16 number (|x| >= 2^-3); below that it goes subnormal, and a layer whose
// activations sit around 1e-3 would lose precision silently (there is a flag for the other end of the range, none for this one).
// Behind a batch normalisation the pre-activation output of channel c is ~ N(beta_c, gamma_c^2) -- that is what the normalisation
// is for -- so the layer's rms is known from the weights: the split copy holds y * 2^e with rms * 2^e ~ 2^4 (values below 2^-3 are
// then < 1 % of the rms, four orders of magnitude of headroom to 65504 remain), and the reader's per-channel scale takes 2^-e
// (exact).  Layers without a normalisation, and tanh (not homogeneous, bounded anyway), keep e = 0.
int act_exponent(const xv_handle* h, const Layer& L) {
  if (!L.has_bn || L.act == ACT_TANH) return 0;
  const auto& g = T(h, L.bn_scope + "/gamma").data;
  const auto& b = T(h, L.bn_scope + "/beta").data;
  double s = 0.0;
  for (size_t i = 0; i < g.size(); ++i) s += (double)g[i] * g[i] + (double)b[i] * b[i];
  const double rms = std::sqrt(s / std::max<size_t>(g.size(), 1));
  if (!(rms > 0.0) || !std::isfinite(rms)) return 0;
  return (int)std::min(20.0, std::max(-20.0, std::floor(4.5 - std::log2(rms))));
}

int upload_layer(xv_handle* h, Layer& L) {
  const int K = L.K(), N = L.cout;
  // row of the packed weight matrix that holds kernel row k: identity, or tap * cin_pad + channel for a first layer
  // whose frames are padded to whole 32-channel blocks (the padding rows stay zero)
  auto krow = [&](int k) { return L.cin_pad ? (k / L.cin) * L.cin_pad + (k % L.cin) : k; };
  L.Kpad = (int)align_up(L.cin_pad ? L.w * L.cin_pad : K, 32);
  L.Npad = (int)align_up(N, 128);
  const auto& W = T(h, L.kernel_name).data;      // [K][N] (HWIO flattened k-major / [in,out])
  const std::vector<float> no_bias((size_t)N, 0.f);      // resnet convs: use_bias=False (model/resnet.py:31)
  const auto& bias = L.has_bias ? T(h, L.bias_name).data : no_bias;
  std::vector<float> vec((size_t)5 * N, 0.f);
  for (int n = 0; n < N; ++n) { vec[n] = bias[n]; vec[(size_t)4 * N + n] = 1.f; }
  if (L.has_bn) {
    std::vector<double> s, t;
    bn_fold(h, L.bn_scope, N, s, t);
    for (int n = 0; n < N; ++n) {
      vec[(size_t)N + n] = (float)s[n];
      vec[(size_t)2 * N + n] = (float)((double)bias[n] * s[n] + t[n]);
    }
  } else {
    for (int n = 0; n < N; ++n) { vec[(size_t)N + n] = 1.f; vec[(size_t)2 * N + n] = bias[n]; }
  }
  if (!L.alpha_name.empty()) {
    const auto& a = T(h, L.alpha_name).data;
    for (int n = 0; n < N; ++n) vec[(size_t)3 * N + n] = a[n];
  }
  XV_HIP(h, L.vec.alloc(vec.size() * sizeof(float)));
  XV_HIP(h, hipMemcpy(L.vec.p, vec.data(), vec.size() * sizeof(float), hipMemcpyHostToDevice));

  if (L.mode == 4) {                                  // conv0_direct_kernel: the 9 x cout kernel as it is, true BN scale / shift
    std::vector<float> wd((size_t)11 * N);
    for (int k = 0; k < 9; ++k)
      for (int n = 0; n < N; ++n) wd[(size_t)k * N + n] = W[(size_t)k * N + n];
    for (int n = 0; n < N; ++n) { wd[(size_t)9 * N + n] = vec[(size_t)N + n]; wd[(size_t)10 * N + n] = vec[(size_t)2 * N + n]; }
    XV_HIP(h, L.wdir.alloc(wd.size() * sizeof(float)));
    XV_HIP(h, hipMemcpy(L.wdir.p, wd.data(), wd.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  const size_t elems = (size_t)L.Npad * L.Kpad;
  if (!L.use_split) {
    std::vector<float> wt(elems, 0.f);
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) wt[(size_t)n * L.Kpad + k] = W[(size_t)k * N + n];
    XV_HIP(h, L.wt.alloc(elems * sizeof(float)));
    XV_HIP(h, hipMemcpy(L.wt.p, wt.data(), elems * sizeof(float), hipMemcpyHostToDevice));
  } else {
    // split-blocked: row n, block kb: [32 x hi | 32 x lo] for k = 32*kb .. 32*kb+31 (xv_epilogue.h)
    const bool f16 = h->desc.precision == XV_PREC_F16X3 || h->desc.precision == XV_PREC_F16F6;
    float wscale = 1.f;
    if (f16) {
      // fp16 hi/lo keeps 22 significand bits only while the low half stays normal (|w * s| >= 2^-3): scale the layer's
      // weights by a power of two so that the largest lands in [8192, 16384); the epilogue's per-channel scale
      // (and the "ones" vector of the affine-stage endpoints) absorbs 1/s exactly
      float maxabs = 0.f;
      for (size_t i = 0; i < (size_t)K * N; ++i) maxabs = std::max(maxabs, std::fabs(W[i]));
      if (maxabs > 0.f && std::isfinite(maxabs)) {
        int e = 0;
        std::frexp(maxabs, &e);                     // maxabs = m * 2^e, m in [0.5, 1)
        wscale = std::ldexp(1.f, std::min(std::max(14 - e, -24), 24));
      }
    }
    std::vector<uint16_t> sb(elems * 2, 0);
    for (int k = 0; k < K; ++k)
      for (int n = 0; n < N; ++n) {
        const float wv = W[(size_t)k * N + n] * wscale;
        const int kr = krow(k);
        const size_t blk = ((size_t)n * (L.Kpad / 32) + kr / 32) * 64;
        if (f16) {
          const uint16_t a = f32_to_f16_rn(wv);
          sb[blk + (kr & 31)] = a;
          sb[blk + 32 + (kr & 31)] = f32_to_f16_rn(wv - f16_to_f32(a));
        } else {
          const uint16_t a = f32_to_bf16_rn(wv);
          sb[blk + (kr & 31)] = a;
          sb[blk + 32 + (kr & 31)] = f32_to_bf16_rn(wv - bf16_to_f32(a));
        }
      }
    if (wscale != 1.f || L.in_exp != 0) {             // fold 1/s into [bn_scale | ones]; bias / shift are not products
      const float inv = std::ldexp(1.f / wscale, -L.in_exp);          // (and the power of two the input's split copy is kept at)
      for (int n = 0; n < N; ++n) { vec[(size_t)N + n] *= inv; vec[(size_t)4 * N + n] *= inv; }
      XV_HIP(h, hipMemcpy(L.vec.p, vec.data(), vec.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    XV_HIP(h, L.wsb.alloc(elems * 4));
    XV_HIP(h, hipMemcpy(L.wsb.p, sb.data(), elems * 4, hipMemcpyHostToDevice));
    // fragment-major copy for the weights-in-registers kernels (v_mfma_f32_16x16x32 A operand: lane = 16 * k-chunk + row):
    // per (32-channel block, K block) 4 KB = [plane hi/lo][16-channel tile][64 lanes][16 B]; one global_load_dwordx4 of a
    // wave = 1 KB contiguous
    std::vector<uint16_t> fr(elems * 2, 0);
    const size_t nkb = L.Kpad / 32;
    for (size_t n = 0; n < (size_t)L.Npad; ++n)
      for (size_t kb = 0; kb < nkb; ++kb)
        for (int q = 0; q < 8; ++q) {                       // SB chunk q = plane * 4 + k-chunk (8 k values, 16 bytes)
          const int plane = q >> 2, g = q & 3, ct = (int)((n >> 4) & 1);
          const size_t src = (n * nkb + kb) * 64 + (size_t)q * 8;
          const size_t dst = ((((n / 32) * nkb + kb) * 4 + plane * 2 + ct) * 64 + g * 16 + (n & 15)) * 8;
          for (int e = 0; e < 8; ++e) fr[dst + e] = sb[src + e];
        }
    XV_HIP(h, L.wfr.alloc(elems * 4));
    XV_HIP(h, hipMemcpy(L.wfr.p, fr.data(), elems * 4, hipMemcpyHostToDevice));
    if (L.use_f6) {
      // gemm_f6v2_kernel operands (the scaled weights w * wscale, like the f16 halves above):
      //   main  [Npad/32][cin/32][4 NQ tap slots, fw used][2 channel tiles][64 lanes = 16 * k-chunk + channel][8 x f16]      (NQ = ceil(fw / 4))
      //   cross [Npad/32][cin/128][fw macro steps][2 terms: q6(w - f16(w)), q6(f16(w))][2 channel tiles] x { 64 x 16 B codes 0-15 |
      //         64 x 16 B {codes 16-23, scale dword (E8M0 in byte 0), pad} },  lane = 16 * K group + channel
      // (a 3 x 3 grid convolution = three taps along time over the 3 cin contiguous channels of a kernel row: HWIO k-major is
      //  already [kt][kf * cin + c][n])
      const int fw = L.mode == 1 ? 3 : L.w, fcin = L.mode == 1 ? 3 * L.cin : L.cin;
      const int ncb = fcin / 32, NQ = (fw + 3) / 4;
      const size_t main_ct = 64 * 16, cross_ct = 2 * 64 * 16;
      // The cross operands are grouped over QUADS of channel blocks: slot p = fw * (cb & 3) + tap of a quad is K group p & 3 of its macro
      // step p >> 2 -- 4 fw slots = fw macro steps exactly, no zero groups:  [Npad/32][cin/128][fw][2 terms][2 tiles]
      const size_t xsteps = (size_t)(ncb / 4) * fw;
      std::vector<unsigned char> wm((size_t)(L.Npad / 32) * ncb * (4 * NQ) * 2 * main_ct, 0), wx((size_t)(L.Npad / 32) * xsteps * 2 * 2 * cross_ct, 0);
      for (int n = 0; n < L.Npad; ++n) {
        const int nb = n >> 5, ct = (n >> 4) & 1, r16 = n & 15;
        for (int cb = 0; cb < ncb; ++cb)
          for (int j = 0; j < 4 * NQ; ++j) {
            float whi[32], wlo[32];
            uint16_t hh[32];
            for (int t = 0; t < 32; ++t) {
              const float wv = (j < fw && n < N) ? W[((size_t)j * fcin + cb * 32 + t) * N + n] * wscale : 0.f;
              hh[t] = f32_to_f16_rn(wv);
              whi[t] = f16_to_f32(hh[t]);
              wlo[t] = wv - whi[t];
            }
            unsigned char* pm = &wm[((((size_t)nb * ncb + cb) * (4 * NQ) + j) * 2 + ct) * main_ct];
            for (int kc = 0; kc < 4; ++kc) memcpy(pm + (16 * kc + r16) * 16, &hh[8 * kc], 16);
            if (j >= fw) continue;                          // (main weights: tap slots padded to 4 NQ; the cross operands have no padding)
            const int slot = (cb & 3) * fw + j;
            const size_t xstep = (size_t)(cb >> 2) * fw + (slot >> 2);
            const int ln = 16 * (slot & 3) + r16;
            for (int term = 0; term < 2; ++term) {          // term 0 multiplies q6(hi) of the activations, term 1 q6(lo)
              unsigned char* px = &wx[(((((size_t)nb * xsteps + xstep) * 2 + term) * 2) + ct) * cross_ct];
              unsigned char c24[24], sc;
              host_quant32(term == 0 ? wlo : whi, c24, &sc);
              memcpy(px + ln * 16, c24, 16);
              memcpy(px + 1024 + ln * 16, c24 + 16, 8);
              px[1024 + ln * 16 + 8] = sc;
            }
          }
      }
      XV_HIP(h, L.wf6m.alloc(wm.size()));
      XV_HIP(h, hipMemcpy(L.wf6m.p, wm.data(), wm.size(), hipMemcpyHostToDevice));
      XV_HIP(h, L.wf6x.alloc(wx.size()));
      XV_HIP(h, hipMemcpy(L.wf6x.p, wx.data(), wx.size(), hipMemcpyHostToDevice));
    }
  }
  return XV_OK;
}

// Fl[k] = total frames of the batch at time level k (Fl[0] = frame_offsets[B])
int64_t value_rows(const xv_handle* h, int vid, const int64_t* Fl, int B) {
  const Value& v = h->values[vid];
  if (v.grid_F > 0) return (Fl[v.tlevel] + 2 * (int64_t)B) * v.grid_S;
  return v.frame_level ? Fl[v.tlevel] - (int64_t)B * v.ctx : B;
}

int64_t value_bytes(const xv_handle* h, int vid, const int64_t* Fl, int B) {
  const Value& v = h->values[vid];
  return align_up((value_rows(h, vid, Fl, B) + kSlackRows) * (int64_t)v.cols * 4, kAlign);
}

int sb_ld(int cols) { return (int)align_up(cols, 32); }

int64_t value_sb_bytes(const xv_handle* h, int vid, const int64_t* Fl, int B) {
  const Value& v = h->values[vid];
  return align_up((value_rows(h, vid, Fl, B) + kSlackRows) * (int64_t)sb_ld(v.cols) * 4, kAlign);
}

}  // namespace

// ============================================================================ C ABI
extern "C" {

const char* xv_version(void) { return "xvec_hip 0.1 gfx950"; }

const char* xv_last_error(const xv_handle* h) { return h ? h->err.c_str() : g_last_error.c_str(); }

int xv_create(const xv_model_desc* desc, int device, xv_handle** out) {
  if (!desc || !out) return fail(nullptr, XV_ERR_INVALID, "xv_create: null argument");
  *out = nullptr;
  if (desc->struct_size != (int32_t)sizeof(xv_model_desc))
    return fail(nullptr, XV_ERR_INVALID, "xv_create: xv_model_desc size %d != %zu (ABI mismatch)", desc->struct_size,
                sizeof(xv_model_desc));
  if (desc->network_type != XV_NET_TDNN && desc->network_type != XV_NET_ETDNN && desc->network_type != XV_NET_RESNET18)
    return fail(nullptr, XV_ERR_UNSUPPORTED, "Not implement network_type %d (tdnn, extended_tdnn, resnet_18)", desc->network_type);
  if (desc->network_type == XV_NET_RESNET18)
    for (int i = 0; i < 4; ++i)
      if (desc->resnet_blocks[i] < 1 || desc->resnet_blocks[i] > 16)
        return fail(nullptr, XV_ERR_INVALID, "xv_create: resnet_blocks[%d] = %d", i, desc->resnet_blocks[i]);
  if (desc->feat_dim < 1 || desc->channels < 1 || desc->num_nodes_pooling_layer < 1 || desc->num_nodes_last_layer < 1)
    return fail(nullptr, XV_ERR_INVALID, "xv_create: non-positive layer width");
  if (desc->precision != XV_PREC_F32 && desc->precision != XV_PREC_BF16X3 && desc->precision != XV_PREC_F16X3 &&
      desc->precision != XV_PREC_F16F6)
    return fail(nullptr, XV_ERR_INVALID, "xv_create: unknown precision %d", desc->precision);
  if (desc->relu_type < XV_ACT_RELU || desc->relu_type > XV_ACT_PRELU)
    return fail(nullptr, XV_ERR_INVALID, "xv_create: unknown relu_type %d", desc->relu_type);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev)
    return fail(nullptr, XV_ERR_HIP, "xv_create: HIP device %d not available (%d visible)", device, ndev);
  xv_handle* h = new (std::nothrow) xv_handle();
  if (!h) return fail(nullptr, XV_ERR_HIP, "xv_create: out of host memory");
  h->desc = *desc;
  h->device = device;
  const int rc = build_graph(h);
  if (rc != XV_OK) {
    g_last_error = h->err;
    delete h;
    return rc;
  }
  *out = h;
  return XV_OK;
}

int xv_set_tensor(xv_handle* h, const char* tf_name, const float* host, const int64_t* shape, int rank) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_set_tensor: null handle");
  if (!tf_name || !host || !shape || rank < 1) return fail(h, XV_ERR_INVALID, "xv_set_tensor: null argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->finalized) return fail(h, XV_ERR_STATE, "xv_set_tensor after xv_finalize");
  auto it = h->tensors.find(tf_name);
  if (it == h->tensors.end()) return fail(h, XV_ERR_INVALID, "variable '%s' is not part of the predict graph", tf_name);
  HostTensor& t = it->second;
  bool same = (int)t.shape.size() == rank;
  int64_t n = 1;
  for (int i = 0; i < rank; ++i) {
    if (same && t.shape[i] != shape[i]) same = false;
    n *= shape[i];
  }
  if (!same) {
    std::string exp, got;
    for (auto s : t.shape) exp += std::to_string(s) + ",";
    for (int i = 0; i < rank; ++i) got += std::to_string(shape[i]) + ",";
    return fail(h, XV_ERR_INVALID, "variable '%s': expected shape [%s] got [%s]", tf_name, exp.c_str(), got.c_str());
  }
  t.data.assign(host, host + n);
  t.set = true;
  return XV_OK;
}

int xv_finalize(xv_handle* h) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_finalize: null handle");
  std::lock_guard<std::mutex> lk(h->mu);
  if (h->finalized) return XV_OK;
  for (const auto& kv : h->tensors)
    if (!kv.second.set) return fail(h, XV_ERR_MISSING_TENSOR, "variable '%s' was never set", kv.first.c_str());
  DeviceGuard g(h->device);
  if (!g.ok) return fail(h, XV_ERR_HIP, "cannot select HIP device %d", h->device);
  for (const Op& op : h->ops) {      // which layers run on the bf16x3 kernel
    if (op.kind != OP_GEMM) continue;
    Layer& L = h->layers[op.layer];
    const Value& vin = h->values[op.in0];
    const bool bf = h->desc.precision != XV_PREC_F32;       // a split format: bf16x3 or f16x3
    if (L.mode == 0) {
      L.im2col = bf && op.in0 == 0;
      L.cin_pad = (L.im2col && L.w <= 9) ? (int)align_up(L.cin, 32) : 0;
      L.use_split = L.im2col || (bf && vin.frame_level && (L.w == 1 || (L.cin % 32 == 0 && L.w <= 9)));   // slab halo of the split kernel
      // two-unit split: the 5-, 7- and 9-tap layers over whole 32-channel blocks (the first layer, K = 5 x 30, stays on the f16 kernel and writes
      // the block format of its reader: gemm_bf16x3_w14p2_kernel<1, 3, true>)
      // (dense layers stay on three units: a one-tap two-unit kernel needs four slabs per macro step = one workgroup per CU, and
      //  with nothing to overlap its prologue and epilogue it was no faster -- profiles/r03/ab_dense_two_unit.txt)
      L.use_f6 = h->desc.precision == XV_PREC_F16F6 && L.use_split && !L.im2col && (L.w == 5 || L.w == 7 || L.w == 9) &&
                 L.cin % 128 == 0 && L.cout % 4 == 0;      // (the kernel takes channel blocks in quads)
    } else {      // grid convolutions: whole SB blocks per tap; conv0 goes through its own im2col
      L.use_split = bf && (L.mode == 4 || L.cin % 32 == 0);
      // two-unit split of the stride-1 3 x 3 convolutions: three taps along time over the 3 C channels of a kernel row
      // (gemm_f6v2_kernel<3, ...>); whole 128-channel tiles only (stage 1 of the default net, 64 channels, is HBM-bound anyway)
      L.use_f6 = h->desc.precision == XV_PREC_F16F6 && h->opt_grid_f6 && L.mode == 1 && L.use_split && L.sw == 1 && L.st == 1 &&
                 L.cin % 128 == 0 && L.cout % 128 == 0;      // (3 cin / 32 channel blocks, taken in quads)
    }
  }
  if (h->desc.precision == XV_PREC_F16X3 || h->desc.precision == XV_PREC_F16F6) {
    for (const Op& op : h->ops) {      // creation order is topological: a value's exponent is known before its readers
      if (op.kind == OP_GEMM) {
        Layer& L = h->layers[op.layer];
        L.in_exp = (L.use_split && !L.im2col && L.mode != 4 && op.in0 > 0) ? h->values[op.in0].sb_exp : 0;
        L.out_exp = act_exponent(h, L);
        h->values[op.out].sb_exp = L.out_exp;
      } else if (op.kind == OP_GRID_MAXPOOL) {
        h->values[op.out].sb_exp = h->values[op.in0].sb_exp;
      }
    }
  }
  for (auto& L : h->layers) {
    const int rc = upload_layer(h, L);
    if (rc != XV_OK) return rc;
  }
  if (h->desc.pooling_type == XV_POOL_SELF_ATTENTION) {
    const auto& q = T(h, std::string(h->desc.network_type == XV_NET_ETDNN ? "etdnn/" : "tdnn/") + "attention/query").data;
    XV_HIP(h, h->query.alloc(q.size() * sizeof(float)));
    XV_HIP(h, hipMemcpy(h->query.p, q.data(), q.size() * sizeof(float), hipMemcpyHostToDevice));
    {
      const int H = h->desc.att_num_heads, dkh = h->att_dk_h;
      h->key_npad = (int)align_up(h->att_dk, 128);
      std::vector<float> qe((size_t)H * h->key_npad, 0.f);
      for (int hd = 0; hd < H; ++hd)
        for (int d = 0; d < dkh; ++d) {
          const int n = h->desc.att_split_key ? hd * dkh + d : d;
          qe[(size_t)hd * h->key_npad + n] = q[(size_t)hd * dkh + d];
        }
      XV_HIP(h, h->query_eff.alloc(qe.size() * sizeof(float)));
      XV_HIP(h, hipMemcpy(h->query_eff.p, qe.data(), qe.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (h->desc.att_apply_nonlinear) {
      const int n = h->pool_dim;
      std::vector<double> s, t;
      bn_fold(h, h->post_bn_scope, n, s, t);
      std::vector<float> vec((size_t)3 * n, 0.f);
      for (int i = 0; i < n; ++i) { vec[i] = (float)s[i]; vec[(size_t)n + i] = (float)t[i]; }
      if (!h->post_alpha_name.empty()) {
        const auto& a = T(h, h->post_alpha_name).data;
        for (int i = 0; i < n; ++i) vec[(size_t)2 * n + i] = a[i];
      }
      XV_HIP(h, h->post_vec.alloc(vec.size() * sizeof(float)));
      XV_HIP(h, hipMemcpy(h->post_vec.p, vec.data(), vec.size() * sizeof(float), hipMemcpyHostToDevice));
    }
  }
  XV_HIP(h, h->ovf_flag.alloc((kFlagWords + kFeatMaxSlots) * sizeof(int)));
  XV_HIP(h, hipMemset(h->ovf_flag.p, 0, (kFlagWords + kFeatMaxSlots) * sizeof(int)));
  XV_HIP(h, hipDeviceSynchronize());
  for (auto& kv : h->tensors) { kv.second.data.clear(); kv.second.data.shrink_to_fit(); }
  h->finalized = true;
  return XV_OK;
}

int xv_set_option(xv_handle* h, const char* name, int value) {
  if (!h || !name) return fail(h, XV_ERR_INVALID, "xv_set_option: null argument");
  std::lock_guard<std::mutex> lk(h->mu);
  if (!strcmp(name, "pool_fusion")) h->opt_pool_fusion = value != 0;
  else if (!strcmp(name, "tail_split")) h->opt_tail_split = value != 0;
  else if (!strcmp(name, "att_fusion")) h->opt_att_fusion = value != 0;
  else if (!strcmp(name, "slab3")) h->opt_slab3 = value != 0;
  else if (!strcmp(name, "onetap_f6_notail")) h->opt_onetap_f6_notail = value != 0;
  else if (!strcmp(name, "grid_f6")) h->opt_grid_f6 = value != 0;      // (before xv_finalize: it decides the weight formats)
  else if (!strcmp(name, "grid_compact")) h->opt_grid_compact = value != 0;
  else if (!strcmp(name, "profile_dominant")) h->opt_profile_dominant = value != 0;
  else return fail(h, XV_ERR_INVALID, "xv_set_option: unknown option '%s'", name);
  return XV_OK;
}

int xv_check_overflow(xv_handle* h, int reset) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_check_overflow: null handle");
  if (!h->finalized || !h->ovf_flag.p) return 0;
  DeviceGuard g(h->device);
  std::vector<int32_t> w((size_t)kFlagWords + kFeatMaxSlots);
  XV_HIP(h, hipMemcpy(w.data(), h->ovf_flag.p, w.size() * sizeof(int32_t), hipMemcpyDeviceToHost));
  int32_t v[2] = {w[0], w[1]};
  for (int i = 0; i < kFeatMaxSlots; ++i) v[1] = std::max(v[1], w[(size_t)kFlagWords + i]);     // non-negative floats order like ints
  if (reset) XV_HIP(h, hipMemset(h->ovf_flag.p, 0, w.size() * sizeof(int32_t)));
  return xv_flags_decode(v);
}

int xv_flags_decode(const int32_t* flags) {
  if (!flags) return XV_ERR_INVALID;
  if (flags[0]) return 1;
  if (flags[1] > 0) {                    // bits of the largest |feature| staged (0: no feature staged, or all zero)
    float mx;
    memcpy(&mx, &flags[1], 4);
    if (mx < 0.00390625f) return 2;      // 2^-8
  }
  return 0;
}

int xv_flags_async(xv_handle* h, int32_t* host_flags, void* stream) {
  if (!h || !host_flags) return fail(h, XV_ERR_INVALID, "xv_flags_async: null argument");
  if (!h->finalized || !h->ovf_flag.p) { host_flags[0] = host_flags[1] = 0; return XV_OK; }
  DeviceGuard g(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // pinned (device-accessible) host memory: one small kernel writes the words there and clears them; anything else: a copy + a memset
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, host_flags) == hipSuccess && attr.type == hipMemoryTypeHost) {
    XV_HIP(h, launch_flags_snapshot(static_cast<int*>(h->ovf_flag.p), host_flags, s));
    return XV_OK;
  }
  (void)hipGetLastError();               // (an unregistered pointer makes the query fail: not an error of ours)
  // pageable destination: the words are reduced into device word 1 first, then copied
  XV_HIP(h, launch_flags_snapshot(static_cast<int*>(h->ovf_flag.p), static_cast<int*>(h->ovf_flag.p) + 2, s));
  XV_HIP(h, hipMemcpyAsync(host_flags, static_cast<int*>(h->ovf_flag.p) + 2, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
  return XV_OK;
}

int xv_node_id(const xv_handle* h, const char* name) {
  if (!h || !name) return XV_ERR_INVALID;
  for (size_t i = 0; i < h->nodes.size(); ++i)
    if (h->nodes[i].name == name) return (int)i;
  return XV_ERR_INVALID;
}

int xv_node_context(const xv_handle* h, int node_id) {
  if (!h || node_id < 0 || node_id >= (int)h->nodes.size()) return XV_ERR_INVALID;
  const Op& op = h->ops[h->nodes[node_id].op];
  const Value& v = h->values[op.out];
  return v.frame_level ? v.ctx : h->final_ctx;
}

int xv_plan_create(xv_handle* h, const int32_t* frame_offsets, int batch, int node_id, void* stream, xv_plan** out) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_plan_create: null handle");
  if (!out || !frame_offsets) return fail(h, XV_ERR_INVALID, "xv_plan_create: null argument");
  *out = nullptr;
  if (!h->finalized) return fail(h, XV_ERR_STATE, "xv_plan_create before xv_finalize");
  if (batch < 1) return fail(h, XV_ERR_INVALID, "xv_plan_create: batch %d < 1", batch);
  if (node_id < 0 || node_id >= (int)h->nodes.size()) return fail(h, XV_ERR_INVALID, "xv_plan_create: bad node id %d", node_id);
  if (frame_offsets[0] != 0) return fail(h, XV_ERR_INVALID, "frame_offsets[0] must be 0");
  const Node& node = h->nodes[node_id];
  const int need_ctx = xv_node_context(h, node_id);
  bool uniform = true;
  for (int b = 0; b < batch; ++b) {
    const int64_t len = (int64_t)frame_offsets[b + 1] - frame_offsets[b];
    if (len <= need_ctx)
      return fail(h, XV_ERR_TOO_SHORT, "utterance %d has %lld frames; node '%s' needs more than %d", b, (long long)len,
                  node.name.c_str(), need_ctx);
    if (len != (int64_t)frame_offsets[1] - frame_offsets[0]) uniform = false;
  }
  const int64_t F0 = frame_offsets[batch];
  if (F0 > (int64_t)1 << 30) return fail(h, XV_ERR_INVALID, "batch of %lld frames is too large for 32-bit row indices", (long long)F0);
  // frame offsets per time level (tf 'same' with stride 2: ceil(L / 2) frames; model/resnet.py:187)
  int max_level = 0;
  for (const Value& v : h->values) max_level = std::max(max_level, v.tlevel);
  std::vector<int32_t> lvl[4];
  int64_t Fl[4] = {F0, F0, F0, F0};
  lvl[0].assign(frame_offsets, frame_offsets + batch + 1);
  for (int k = 1; k <= max_level && k < 4; ++k) {
    lvl[k].resize(batch + 1);
    lvl[k][0] = 0;
    for (int b = 0; b < batch; ++b) lvl[k][b + 1] = lvl[k][b] + (lvl[k - 1][b + 1] - lvl[k - 1][b] + 1) / 2;
    Fl[k] = lvl[k][batch];
  }

  // ops needed for this node (backward closure), then in topological (= creation) order
  std::vector<char> need(h->ops.size(), 0);
  std::vector<int> producer(h->values.size(), -1);
  for (size_t i = 0; i < h->ops.size(); ++i) producer[h->ops[i].out] = (int)i;
  std::vector<int> stack{node.op};
  while (!stack.empty()) {
    const int o = stack.back();
    stack.pop_back();
    if (need[o]) continue;
    need[o] = 1;
    for (int in : {h->ops[o].in0, h->ops[o].in1})
      if (in > 0 && producer[in] >= 0) stack.push_back(producer[in]);
  }

  xv_plan* p = new (std::nothrow) xv_plan();
  if (!p) return fail(h, XV_ERR_HIP, "out of host memory");
  p->h = h;
  p->offsets.assign(frame_offsets, frame_offsets + batch + 1);
  for (int k = 0; k <= max_level && k < 4; ++k) p->lvl_offsets[k] = lvl[k];
  p->uniform_len = uniform;
  p->uniform_L = frame_offsets[1] - frame_offsets[0];

  std::vector<int> order;
  for (size_t i = 0; i < h->ops.size(); ++i) if (need[i]) order.push_back((int)i);

  // Fused attentive pooling (bf16x3 kernel, attention epilogue): the last key layer emits score partials instead of
  // the key when the scores are its only reader; the value layer emits weighted moments instead of the value when
  // the pooling is its only reader -- it then has to run AFTER the softmax, so it is moved behind it in the order.
  auto readers_of = [&](int v) {
    int n = 0;
    for (int o2 : order)
      if (h->ops[o2].in0 == v || h->ops[o2].in1 == v) ++n;
    return n;
  };
  int key_prod = -1, val_prod = -1, att_pool_value = -1;
  if (h->opt_att_fusion && h->desc.pooling_type == XV_POOL_SELF_ATTENTION && h->desc.att_num_heads <= 8) {
    auto fusable = [&](int v) {
      if (v <= 0 || producer[v] < 0 || producer[v] == node.op) return -1;
      const Op& prod = h->ops[producer[v]];
      if (prod.kind != OP_GEMM || prod.in1 > 0 || readers_of(v) != 1) return -1;
      const Layer& L = h->layers[prod.layer];
      return (L.use_split && !L.im2col && L.mode == 0 && L.w == 1 && (L.cout & 3) == 0) ? producer[v] : -1;
    };
    int softmax_op = -1, pool_op = -1;
    for (int o : order) {
      if (h->ops[o].kind == OP_ATT_SCORES) key_prod = fusable(h->ops[o].in0);
      if (h->ops[o].kind == OP_ATT_SOFTMAX) softmax_op = o;
      if (h->ops[o].kind == OP_ATT_POOL) pool_op = o;
    }
    if (pool_op >= 0 && softmax_op >= 0) {
      const int H = h->desc.att_num_heads, dv = h->att_dv;
      const bool heads_ok = H == 1 || !h->desc.att_split_value || (dv / H) % 32 == 0;
      const int vp = heads_ok ? fusable(h->ops[pool_op].in0) : -1;
      if (vp >= 0) {
        val_prod = vp;
        att_pool_value = h->ops[pool_op].in0;
        order.erase(std::find(order.begin(), order.end(), vp));
        order.insert(std::find(order.begin(), order.end(), pool_op), vp);
      }
    }
  }

  // liveness: last step index that reads each value
  std::vector<int> last_use(h->values.size(), -1);
  for (size_t s = 0; s < order.size(); ++s)
    for (int in : {h->ops[order[s]].in0, h->ops[order[s]].in1})
      if (in > 0) last_use[in] = (int)s;

  // first-fit arena with release after last use
  struct Block { int64_t off, size; };
  std::vector<Block> free_list;
  int64_t arena_top = 0;
  std::vector<int64_t> voff(h->values.size(), -1), vsize(h->values.size(), 0);
  auto arena_alloc = [&](int64_t size) -> int64_t {
    for (size_t i = 0; i < free_list.size(); ++i)
      if (free_list[i].size >= size) {
        const int64_t off = free_list[i].off;
        free_list[i].off += size;
        free_list[i].size -= size;
        if (free_list[i].size == 0) free_list.erase(free_list.begin() + i);
        return off;
      }
    const int64_t off = arena_top;
    arena_top += size;
    return off;
  };
  auto arena_free = [&](int64_t off, int64_t size) {
    free_list.push_back({off, size});
    std::sort(free_list.begin(), free_list.end(), [](const Block& a, const Block& b) { return a.off < b.off; });
    for (size_t i = 0; i + 1 < free_list.size();)
      if (free_list[i].off + free_list[i].size == free_list[i + 1].off) {
        free_list[i].size += free_list[i + 1].size;
        free_list.erase(free_list.begin() + i + 1);
      } else {
        ++i;
      }
  };

  // which formats of each value are read: fp32 by the f32 GEMM / pooling / elementwise kernels,
  // split-blocked by the bf16x3 GEMM
  std::vector<char> want_f32(h->values.size(), 0), want_sb(h->values.size(), 0);
  for (int o : order) {
    const Op& op = h->ops[o];
    const bool split_in = op.kind == OP_GEMM && h->layers[op.layer].use_split;
    if (op.in0 > 0) (split_in ? want_sb : want_f32)[op.in0] = 1;
    if (op.in1 > 0) want_f32[op.in1] = 1;
  }
  std::vector<int64_t> voff_sb(h->values.size(), -1), vsize_sb(h->values.size(), 0);

  // statistics pooling fused into the epilogue of the dense layer that feeds it, when that
  // layer's output has no other reader in this plan
  int fused_value = -1;
  for (int o : order) {
    const Op& op = h->ops[o];
    if (op.kind != OP_STAT_POOL || op.in0 <= 0 || producer[op.in0] < 0) continue;
    const Op& prod = h->ops[producer[op.in0]];
    if (prod.kind != OP_GEMM || producer[op.in0] == node.op) continue;
    const Layer& L = h->layers[prod.layer];
    int readers = 0;
    for (int o2 : order)
      if (h->ops[o2].in0 == op.in0 || h->ops[o2].in1 == op.in0) ++readers;
    if (readers == 1 && L.w == 1 && (L.cout & 3) == 0 && h->opt_pool_fusion) fused_value = op.in0;
  }
  const int slot_value = fused_value >= 0 ? fused_value : att_pool_value;   // the value whose rows are pooled per 64-row slot
  if (slot_value >= 0) {
    const int ctx = h->values[slot_value].ctx;
    const std::vector<int32_t>& so = lvl[h->values[slot_value].tlevel];
    std::vector<int32_t> slotbase(batch);
    int64_t nslots = 0;
    for (int b = 0; b < batch; ++b) {
      const int r0 = so[b] - b * ctx, r1 = so[b + 1] - (b + 1) * ctx;
      const int t0 = r0 >> 6, t1 = (r1 - 1) >> 6;
      slotbase[b] = (int32_t)(nslots - t0);
      nslots += t1 - t0 + 1;
    }
    p->pool_slots = nslots;
    p->offsets_slotbase = std::move(slotbase);
  }

  int64_t total_flops = 0;
  int64_t rowmap_elems = 0;
  for (size_t s = 0; s < order.size(); ++s) {
    const Op& op = h->ops[order[s]];
    PlanStep st;
    int64_t step_scratch = 0, step_scratch2 = 0;            // released once this step's outputs are placed
    st.op = order[s];
    st.to_out = (order[s] == node.op);
    st.rows_in = op.in0 >= 0 ? value_rows(h, op.in0, Fl, batch) : 0;
    st.rows_out = value_rows(h, op.out, Fl, batch);
    st.lvl_in = op.in0 >= 0 ? h->values[op.in0].tlevel : 0;
    st.lvl_out = h->values[op.out].tlevel;
    st.frames_out = Fl[st.lvl_out];
    st.in0_off = op.in0 == 0 ? -2 : (op.in0 > 0 ? voff[op.in0] : -1);
    st.in0_sb_off = op.in0 > 0 ? voff_sb[op.in0] : -1;
    st.in1_off = op.in1 > 0 ? voff[op.in1] : -1;
    if (op.kind == OP_GEMM) {
      const Layer& L = h->layers[op.layer];
      st.stage = st.to_out ? node.stage : L.final_stage();
      const int64_t padded_rows = Fl[st.lvl_in] + 2 * (int64_t)batch;   // input time rows incl. the two border rows per utterance
      if (L.mode == 0) st.M = (int)(st.rows_in - (L.w - 1));
      else if (L.mode == 1 && L.use_f6) {
        st.trows = true;                   // rows = padded time rows, frequency bins = a second tile dimension (csrc/grid.hip)
        st.M = (int)padded_rows;
      }
      else if ((L.mode == 1 || L.mode == 2) && L.use_split && h->opt_grid_compact &&
               st.rows_in * (int64_t)sb_ld(L.cin) * 4 < ((int64_t)1 << 32)) {
        st.compact = true;                 // rows = output bins; 32-bit byte offsets of the window positions
        st.M = (int)(Fl[st.lvl_out] * L.Fout);
      }
      else if (L.mode == 1 || L.mode == 2) st.M = (int)(padded_rows * (h->values[op.in0].grid_S / L.sw));
      else if (L.mode == 3) st.M = (int)padded_rows;
      else st.M = (int)(padded_rows * h->values[op.out].grid_S);     // conv0: one row per output grid position
      // does every border position of the output get a zero-writing GEMM row? (csrc/grid.hip)  If not the value is
      // zeroed as a whole before the layer runs.
      if (st.compact || st.trows) st.grid_cover = false;
      else if (L.mode == 1 || L.mode == 2) st.grid_cover = L.st == 1 && h->values[op.in0].grid_S / L.sw == h->values[op.out].grid_S;
      else if (L.mode == 4) st.grid_cover = true;
      if (L.w > 1 || L.mode != 0) {
        st.rowmap = (int)p->rowmap_off.size();
        p->rowmap_off.push_back(rowmap_elems);
        rowmap_elems += align_up(st.M, 64);
        if (st.compact) {
          st.arow = (int)p->rowmap_off.size();
          p->rowmap_off.push_back(rowmap_elems);
          rowmap_elems += align_up(st.M, 128);
        }
      }
      const int64_t valid_out = L.mode == 0 ? st.rows_out : Fl[st.lvl_out] * (L.mode == 3 ? 1 : L.Fout);
      st.flops = 2 * valid_out * (int64_t)L.cout * L.K();
      st.bytes = 4 * (st.rows_in * L.cin + st.rows_out * L.cout + (int64_t)L.K() * L.cout);
      int64_t scratch = 0;
      if (L.mode == 4) {
        scratch = ((int64_t)st.M + kSlackRows) * 32 * 4;              // conv0 im2col rows (K = 9 padded to 32)
      } else if (L.im2col) {
        scratch = L.cin_pad ? (st.rows_in + kSlackRows) * (int64_t)L.cin_pad * 4 : (st.M + kSlackRows) * (int64_t)L.Kpad * 4;
      } else if (!L.use_split && op.out != fused_value) {
        st.ksplit = gemm_f32_ksplit(st.M, L.Kpad, L.Npad);
        if (st.ksplit > 1) scratch = (int64_t)st.ksplit * st.M * L.Npad * 4;
      } else if (L.use_f6) {
        scratch = (st.rows_in + kSlackRows) * (int64_t)sb_ld(L.cin) * 4;      // the input in the block format of gemm_f16f6.hip
        if (h->opt_tail_split && op.in1 <= 0 && L.mode == 0) {
          const int64_t part = gemm_bf16x3_tail_plan(st.M, L.Kpad, L.Npad, L.w, &st.tail_mt, &st.ksplit, 4);
          if (part > 0) {
            step_scratch2 = align_up(part, kAlign);
            st.scratch2_off = arena_alloc(step_scratch2);
          }
        }
      } else if (L.use_split && L.mode == 0 && op.out != fused_value && op.in1 <= 0 && h->opt_tail_split &&
                 order[s] != key_prod && order[s] != val_prod) {
        scratch = gemm_bf16x3_tail_plan(st.M, L.Kpad, L.Npad, L.w, &st.tail_mt, &st.ksplit);
      }
      if (scratch > 0) {
        step_scratch = align_up(scratch, kAlign);
        st.scratch_off = arena_alloc(step_scratch);
      }
    } else if (op.kind == OP_AFFINE_ACT) {
      st.stage = st.to_out ? node.stage : 2;
      st.bytes = 8 * st.rows_out * h->values[op.out].cols;
    } else if (op.kind == OP_ATT_POOL && val_prod >= 0) {      // finalize only: reads the weighted (s1, m2) slots
      st.fuse_att = 1;
      st.bytes = 4 * (p->pool_slots * (2 * (h->pool_dim / 2) + h->desc.att_num_heads) + st.rows_out * h->values[op.out].cols);
      st.flops = 8 * p->pool_slots * (h->pool_dim / 2);
    } else if (op.kind == OP_STAT_POOL || op.kind == OP_ATT_POOL) {
      st.bytes = 4 * (st.rows_in * h->values[op.in0].cols + st.rows_out * h->values[op.out].cols);
      st.flops = 4 * st.rows_in * h->values[op.in0].cols;
      if (op.kind == OP_STAT_POOL && op.in0 == fused_value) {      // finalize only: reads the (sum, M2) slots
        st.fuse_pool = true;
        st.bytes = 4 * (p->pool_slots * 2 * h->values[op.in0].cols + st.rows_out * h->values[op.out].cols);
        st.flops = 6 * p->pool_slots * h->values[op.in0].cols;
      }
    } else if (op.kind == OP_ATT_SCORES) {
      st.bytes = 4 * st.rows_in * h->values[op.in0].cols;
      st.flops = 2 * st.rows_in * (int64_t)h->att_dk_h * h->desc.att_num_heads;
      if (key_prod >= 0) {               // reduce form: reads the partial planes
        st.fuse_att = 1;
        st.att_ld = align_up(st.rows_in, 64);
        st.bytes = 4 * (st.att_ld * (h->key_npad / 32) + st.rows_in) * h->desc.att_num_heads;
      }
    } else {
      st.bytes = 8 * st.rows_out * h->values[op.out].cols;
    }
    total_flops += st.flops;
    // softmax works in place on the scores buffer; everything else gets its own block(s)
    if (op.kind == OP_ATT_SOFTMAX) {
      voff[op.out] = voff[op.in0];
      vsize[op.out] = vsize[op.in0];
      vsize[op.in0] = 0;                 // ownership moves to the softmax value
      st.out_off = voff[op.out];
      if (val_prod >= 0) {               // the per-slot weight sums live behind the weights (block sized by ATT_SCORES)
        st.fuse_att = 1;
        st.att_s0_off = voff[op.out] + align_up(st.rows_out * (int64_t)h->desc.att_num_heads * 4, kAlign);
      }
    } else if (st.to_out && !node.att_weights && h->values[op.out].grid_F == 0) {
      st.out_off = -1;                   // straight into the caller's output buffer (fp32)
    } else if (st.to_out && !node.att_weights) {
      st.unpad_to_out = true;            // grid-valued node: padded grid in the workspace, then unpad into `out`
      vsize[op.out] = value_bytes(h, op.out, Fl, batch);
      voff[op.out] = arena_alloc(vsize[op.out]);
      st.out_off = voff[op.out];
    } else {
      if (op.kind == OP_GEMM && order[s] == key_prod) {
        st.fuse_att = 1;                    // [Npad / 32][H] planes of partial scores, row stride att_ld
        st.att_ld = align_up(st.rows_out, 64);
        vsize[op.out] = align_up((int64_t)(h->layers[op.layer].Npad / 32) * h->desc.att_num_heads * st.att_ld * 4, kAlign);
        voff[op.out] = arena_alloc(vsize[op.out]);
        st.out_off = voff[op.out];
        st.bytes = 4 * (st.rows_in * h->layers[op.layer].cin + (int64_t)h->layers[op.layer].K() * h->layers[op.layer].cout) +
                   vsize[op.out];
      } else if (op.kind == OP_GEMM && order[s] == val_prod) {
        st.fuse_att = 2;                    // weighted (s1, m2) per slot and output column
        vsize[op.out] = align_up(p->pool_slots * 2 * (int64_t)(h->pool_dim / 2) * 4, kAlign);
        voff[op.out] = arena_alloc(vsize[op.out]);
        st.out_off = voff[op.out];
        st.bytes = 4 * (st.rows_in * h->layers[op.layer].cin + (int64_t)h->layers[op.layer].K() * h->layers[op.layer].cout +
                        st.rows_out * h->desc.att_num_heads) + vsize[op.out];
        for (const PlanStep& prev : p->steps)
          if (h->ops[prev.op].kind == OP_ATT_SOFTMAX) { st.att_w_off = prev.out_off; st.att_s0_off = prev.att_s0_off; }
      } else if (op.out == fused_value) {
        st.fuse_pool = true;
        vsize[op.out] = align_up(p->pool_slots * 2 * (int64_t)h->values[op.out].cols * 4, kAlign);
        voff[op.out] = arena_alloc(vsize[op.out]);
        st.out_off = voff[op.out];
      } else if (want_f32[op.out] || node.att_weights) {
        vsize[op.out] = value_bytes(h, op.out, Fl, batch);
        if (op.kind == OP_ATT_SCORES && val_prod >= 0)       // + the per-slot weight sums written by the softmax step
          vsize[op.out] = align_up(st.rows_out * (int64_t)h->desc.att_num_heads * 4, kAlign) +
                          align_up(p->pool_slots * (int64_t)h->desc.att_num_heads * 4, kAlign) + kAlign;
        voff[op.out] = arena_alloc(vsize[op.out]);
        st.out_off = voff[op.out];
      }
      if (want_sb[op.out]) {
        vsize_sb[op.out] = value_sb_bytes(h, op.out, Fl, batch);
        voff_sb[op.out] = arena_alloc(vsize_sb[op.out]);
        st.out_sb_off = voff_sb[op.out];
      }
    }
    if (op.kind == OP_ATT_POOL && val_prod >= 0)
      for (const PlanStep& prev : p->steps)
        if (h->ops[prev.op].kind == OP_ATT_SOFTMAX) st.att_s0_off = prev.att_s0_off;
    p->steps.push_back(st);
    if (step_scratch > 0) arena_free(st.scratch_off, step_scratch);
    if (step_scratch2 > 0) arena_free(st.scratch2_off, step_scratch2);
    for (int in : {op.in0, op.in1})
      if (in > 0 && last_use[in] == (int)s) {
        if (vsize[in] > 0) { arena_free(voff[in], vsize[in]); vsize[in] = 0; }
        if (vsize_sb[in] > 0) { arena_free(voff_sb[in], vsize_sb[in]); vsize_sb[in] = 0; }
      }
  }

  for (size_t i = 0; i < p->steps.size(); ++i)
    if (p->steps[i].flops > p->steps[p->dominant_step].flops) p->dominant_step = (int)i;

  // two-unit layers back to back: the producer's epilogue writes the consumer's block format directly when nobody else reads
  // the value (no fp32 copy, not the requested node, final stage) -- otherwise the consumer converts the split-blocked rows
  for (PlanStep& cs : p->steps) {
    const Op& cop = h->ops[cs.op];
    if (cop.kind != OP_GEMM || !h->layers[cop.layer].use_f6) continue;
    const bool grid = h->layers[cop.layer].mode == 1;
    // readers of the value's SPLIT copy: every op that takes it as its first input; a second input is the fp32 residual of a ResNet
    // block (another copy of the value) -- in the TDNN graphs it does not occur, and disqualifies as before
    // Several readers are fine when every one of them is a two-unit layer.
    if (cs.in_f6) continue;                // (marked with an earlier reader of the same value)
    int readers = 0;
    bool all_f6 = true;
    PlanStep* prod = nullptr;
    for (PlanStep& os : p->steps) {
      const Op& o = h->ops[os.op];
      if (o.in0 == cop.in0 || (!grid && o.in1 == cop.in0)) {
        ++readers;
        if (o.kind != OP_GEMM || !h->layers[o.layer].use_f6 || o.in0 != cop.in0) all_f6 = false;
      }
      if (o.out == cop.in0) prod = &os;
    }
    if (!prod || !all_f6 || (grid && readers != 1)) continue;
    const Op& pop = h->ops[prod->op];
    if (pop.kind != OP_GEMM) continue;
    const Layer& PL = h->layers[pop.layer];
    if (prod->to_out || prod->out_sb_off < 0 || prod->stage != PL.final_stage()) continue;
    if (grid) {
      // producers: another two-unit grid layer (its residual epilogue also writes the fp32 copy and adds a residual), or a stride-2
      // 3 x 3 convolution of the gathered f16 kernel (EPI = 3: the block format only)
      const bool gather = PL.mode == 1 && !PL.use_f6 && PL.use_split && prod->compact && prod->out_off < 0 && pop.in1 <= 0;
      if (!((PL.mode == 1 && PL.use_f6) || gather)) continue;
    } else {
      // producers that can write the format: another two-unit layer, or a layer of the f16 kernels with one tap (the dense layers
      // between the convolutions of the extended TDNN) or >= 5 taps -- their EPI = 3 forms; a K-split tail of theirs is finished by the
      // two-unit kernel's reduce (launch_f6v2_tail_reduce)
      const bool f16_layer = PL.mode == 0 && PL.use_split && !PL.use_f6 && (PL.w == 1 ? !PL.im2col : PL.w >= 5 && PL.w <= 9) &&
                             (PL.im2col ? PL.cin_pad > 0 : PL.cin % 32 == 0) && prod->fuse_att == 0 && !prod->fuse_pool;
      if (!((PL.mode == 0 && PL.use_f6) || f16_layer) || prod->out_off >= 0 || pop.in1 > 0) continue;
    }
    prod->out_f6 = true;
    for (PlanStep& os : p->steps) {
      const Op& o = h->ops[os.op];
      if (o.kind == OP_GEMM && o.in0 == cop.in0 && h->layers[o.layer].use_f6) os.in_f6 = true;
    }
    // a one-tap producer keeps the three-slab kernel for all its tiles: its K-split tail (raw slices of the two-slab kernel + the
    // block-format reduce) costs more than the third of a round it saves (extended TDNN in-process A/B: 2.192 -> 2.172 ms,
    // profiles/r03/ab_onetap_f6_notail.txt)
    if (!grid && PL.mode == 0 && !PL.use_f6 && PL.w == 1 && h->opt_onetap_f6_notail) { prod->tail_mt = 0; prod->ksplit = 1; }
  }

  // output shape
  const Op& top = h->ops[node.op];
  const Value& vout = h->values[top.out];
  xv_plan_info& I = p->info;
  I.struct_size = (int32_t)sizeof(xv_plan_info);
  I.node_id = node_id;
  I.batch = batch;
  I.in_frames = F0;
  I.flops = total_flops;
  if (node.att_weights) {
    if (!uniform) {
      delete p;
      return fail(h, XV_ERR_INVALID, "attention_weights [b,h,l] needs utterances of equal length");
    }
    I.frame_level = 0;
    I.out_rows = (int64_t)batch * h->desc.att_num_heads;
    I.out_cols = p->uniform_L - vout.ctx;
  } else if (vout.grid_F > 0) {
    I.frame_level = 1;
    I.out_rows = Fl[vout.tlevel] * vout.grid_F;       // [sum L_b, F, C] without the border
    I.out_cols = vout.cols;
  } else {
    I.frame_level = vout.frame_level ? 1 : 0;
    I.out_rows = value_rows(h, top.out, Fl, batch);
    I.out_cols = vout.cols;
  }
  I.workspace_bytes = align_up(arena_top, kAlign) + kAlign;

  // device index arrays
  DeviceGuard g(h->device);
  hipStream_t s = static_cast<hipStream_t>(stream);
  auto bail = [&](hipError_t e, const char* what) {
    const int rc = fail(h, XV_ERR_HIP, "%s failed: %s", what, hipGetErrorString(e));
    xv_plan_destroy(p);
    return rc;
  };
  hipError_t e;
  if ((e = pool_take(h, (size_t)(batch + 1) * 4, p->d_offsets)) != hipSuccess) return bail(e, "hipMalloc(offsets)");
  if ((e = hipMemcpyAsync(p->d_offsets.p, p->offsets.data(), (size_t)(batch + 1) * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
    return bail(e, "hipMemcpyAsync(offsets)");
  for (int k = 1; k <= max_level && k < 4; ++k) {
    if ((e = pool_take(h, (size_t)(batch + 1) * 4, p->d_lvl[k])) != hipSuccess) return bail(e, "hipMalloc(level offsets)");
    if ((e = hipMemcpyAsync(p->d_lvl[k].p, p->lvl_offsets[k].data(), (size_t)(batch + 1) * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
      return bail(e, "hipMemcpyAsync(level offsets)");
  }
  if (rowmap_elems > 0) {
    if ((e = pool_take(h, (size_t)rowmap_elems * 4, p->d_rowmaps)) != hipSuccess) return bail(e, "hipMalloc(rowmaps)");
    for (const PlanStep& st : p->steps) {
      if (st.rowmap < 0) continue;
      const Op& op = h->ops[st.op];
      const Layer& L = h->layers[op.layer];
      const int ctx_in = h->values[op.in0].ctx;
      const int32_t* doff = p->dev_offsets(st.lvl_in);
      int32_t* rm = static_cast<int32_t*>(p->d_rowmaps.p) + p->rowmap_off[st.rowmap];
      if (L.mode == 0) e = launch_build_rowmap(doff, batch, ctx_in, L.w, rm, st.M, s);
      else if (st.trows) e = launch_build_rowmap_trows(doff, batch, h->values[op.out].grid_S, rm, st.M, s);
      else if (st.compact)
        e = launch_build_rowmap_grid_compact(doff, p->dev_offsets(st.lvl_out), batch, h->values[op.in0].grid_S,
                                             h->values[op.out].grid_S, L.Fout, L.sw, L.st, L.mode == 1 ? 3 : 1,
                                             static_cast<int32_t*>(p->d_rowmaps.p) + p->rowmap_off[st.arow], rm, st.M,
                                             align_up(st.M, 128), s);
      else if ((L.mode == 1 || L.mode == 2) && L.st == 2)
        e = launch_build_rowmap_grid_ts(doff, p->dev_offsets(st.lvl_out), batch, h->values[op.in0].grid_S / L.sw, L.Fout,
                                        h->values[op.out].grid_S, L.mode == 1 ? 3 : 1, rm, st.M, s);
      else if (L.mode == 1 || L.mode == 2)
        e = launch_build_rowmap_grid(doff, batch, h->values[op.in0].grid_S / L.sw, L.Fout, h->values[op.out].grid_S,
                                     st.grid_cover ? 1 : 0, rm, st.M, s);
      else if (L.mode == 3) e = launch_build_rowmap_rows(doff, batch, rm, st.M, s);
      else e = launch_build_rowmap_interior(doff, batch, L.Fout, h->values[op.out].grid_S, rm, st.M, s);
      if (e != hipSuccess) return bail(e, "build_rowmap");
    }
  }
  if (slot_value >= 0) {
    const int ctx = h->values[slot_value].ctx;
    const int64_t rows = value_rows(h, slot_value, Fl, batch);
    if ((e = pool_take(h, (size_t)rows * 4, p->d_row2utt)) != hipSuccess) return bail(e, "hipMalloc(row2utt)");
    if ((e = pool_take(h, (size_t)batch * 4, p->d_slotbase)) != hipSuccess) return bail(e, "hipMalloc(slotbase)");
    if ((e = hipMemcpyAsync(p->d_slotbase.p, p->offsets_slotbase.data(), (size_t)batch * 4, hipMemcpyHostToDevice, s)) != hipSuccess)
      return bail(e, "hipMemcpyAsync(slotbase)");
    if ((e = launch_build_row2utt(p->dev_offsets(h->values[slot_value].tlevel), batch, ctx,
                                  static_cast<int32_t*>(p->d_row2utt.p), (int)rows, s)) != hipSuccess)
      return bail(e, "build_row2utt");
  }
  // no host synchronisation: the index arrays are filled in stream order, ahead of any xv_forward enqueued on
  // `stream` afterwards (the host copies they are filled from belong to the plan)
  *out = p;
  return XV_OK;
}

int xv_plan_query(const xv_plan* p, xv_plan_info* info) {
  if (!p || !info) return XV_ERR_INVALID;
  *info = p->info;
  return XV_OK;
}

void xv_plan_destroy(xv_plan* p) {
  if (!p) return;
  {
    DeviceGuard g(p->h->device);
    pool_give(p->h, p->d_offsets);
    for (int k = 1; k < 4; ++k) pool_give(p->h, p->d_lvl[k]);
    pool_give(p->h, p->d_rowmaps);
    pool_give(p->h, p->d_row2utt);
    pool_give(p->h, p->d_slotbase);
  }
  delete p;
}

static const char* step_name(const xv_handle* h, const PlanStep& st) {
  const Op& op = h->ops[st.op];
  switch (op.kind) {
    case OP_GEMM: {
      const Layer& L = h->layers[op.layer];
      if (!L.ep[ST_AFFINE].empty()) return L.ep[ST_AFFINE].c_str();
      const size_t a = L.kernel_name.find('/'), b = L.kernel_name.rfind('/');      // "resnet_18/conv1a_conv0/kernel"
      static thread_local std::string tmp;
      tmp = (a != std::string::npos && b > a) ? L.kernel_name.substr(a + 1, b - a - 1) : L.kernel_name;
      return tmp.c_str();
    }
    case OP_STAT_POOL: return "stat_pool";
    case OP_ATT_SCORES: return "att_scores";
    case OP_ATT_SOFTMAX: return "att_softmax";
    case OP_ATT_POOL: return "att_pool";
    case OP_AFFINE_ACT: return "att_post";
    case OP_L2_SCALE: return "l2_scale";
    case OP_GRID_MAXPOOL: return "conv0_max";
    default: return "op";
  }
}

static int run_plan(xv_handle* h, const xv_plan* p, const float* feats, int feat_ld, float* out, int64_t out_cap,
                    void* workspace, int64_t ws_bytes, hipStream_t s) {
  if (!h || !p) return fail(h, XV_ERR_INVALID, "xv_forward: null handle/plan");
  if (p->h != h) return fail(h, XV_ERR_INVALID, "xv_forward: plan belongs to another handle");
  if (!feats || !out) return fail(h, XV_ERR_INVALID, "xv_forward: null feature/output pointer");
  if (feat_ld < h->desc.feat_dim) return fail(h, XV_ERR_INVALID, "xv_forward: feat_ld %d < feature_dim %d", feat_ld, h->desc.feat_dim);
  if (out_cap < p->info.out_rows * p->info.out_cols)
    return fail(h, XV_ERR_WORKSPACE, "xv_forward: output capacity %lld < %lld", (long long)out_cap,
                (long long)(p->info.out_rows * p->info.out_cols));
  if (ws_bytes < p->info.workspace_bytes || (!workspace && p->info.workspace_bytes > 0))
    return fail(h, XV_ERR_WORKSPACE, "xv_forward: workspace %lld bytes < %lld", (long long)ws_bytes, (long long)p->info.workspace_bytes);
  char* ws = reinterpret_cast<char*>(align_up((int64_t)reinterpret_cast<uintptr_t>(workspace), kAlign));
  DeviceGuard g(h->device);
  if (!g.ok) return fail(h, XV_ERR_HIP, "cannot select HIP device %d", h->device);
  const int B = p->info.batch;
  const int32_t* off = static_cast<const int32_t*>(p->d_offsets.p);
  const xv_model_desc& d = h->desc;
  const bool split = d.precision != XV_PREC_F32;
  const int f16 = d.precision == XV_PREC_F16X3 || d.precision == XV_PREC_F16F6;

  bool prof = false;
  size_t prof_base = 0;
  if (h->profiling) {                     // reserve this forward's events under the lock; record them outside it
    std::lock_guard<std::mutex> lk(h->prof_mu);
    if (h->profiling && h->prof_next + 2 * p->steps.size() <= h->prof_pool.size()) {
      prof = true;
      prof_base = h->prof_next;
      h->prof_next += 2 * p->steps.size();
      h->prof_forwards++;
    }
  }

  for (size_t si = 0; si < p->steps.size(); ++si) {
    const PlanStep& st = p->steps[si];
    const Op& op = h->ops[st.op];
    auto in_ptr = [&](int64_t o) -> const float* {
      return o == -2 ? feats : reinterpret_cast<const float*>(ws + o);
    };
    float* optr = st.out_off >= 0 ? reinterpret_cast<float*>(ws + st.out_off) : out;
    const int32_t* off_in = p->dev_offsets(st.lvl_in);      // frame offsets at the time level of the step's input / output
    const int32_t* off_out = p->dev_offsets(st.lvl_out);
    (void)split;
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    const bool prof_step = prof && (!h->opt_profile_dominant || (int)si == p->dominant_step);
    if (prof_step) {
      pe0 = h->prof_pool[prof_base + 2 * si];
      pe1 = h->prof_pool[prof_base + 2 * si + 1];
      XV_HIP(h, hipEventRecord(pe0, s));
    }
    switch (op.kind) {
      case OP_GEMM: {
        const Layer& L = h->layers[op.layer];
        GemmArgs a{};
        a.X = in_ptr(st.in0_off);
        a.ldx = op.in0 == 0 ? feat_ld : L.cin;
        a.cin = L.cin; a.M = st.M; a.K = L.w * L.cin; a.N = L.cout;
        a.Wt = static_cast<const float*>(L.wt.p); a.Kpad = L.Kpad; a.Npad = L.Npad;
        const bool bn_stage = st.stage >= ST_BN;
        a.scale = (bn_stage || !L.has_bn) ? L.d_scale() : L.d_ones();
        a.shift = (bn_stage || !L.has_bn) ? L.d_shift() : L.d_bias();
        a.act = (st.stage == ST_ACT) ? L.act : ACT_NONE;
        a.alpha = (a.act == ACT_PRELU) ? L.d_alpha() : nullptr;
        a.rowmap = st.rowmap >= 0 ? static_cast<const int32_t*>(p->d_rowmaps.p) + p->rowmap_off[st.rowmap] : nullptr;
        a.Y = optr; a.ldy = L.cout;
        a.K = L.K();
        a.f16 = f16;
        a.slab3 = h->opt_slab3;
        a.ovf = static_cast<int*>(h->ovf_flag.p);
        a.sb_mul = std::ldexp(1.f, L.out_exp);
        const Value& vo = h->values[op.out];
        if (vo.grid_F > 0) {
          // the border of a grid output must be zero.  Covered outputs: the layer's own zero-writing rows do it, except
          // for the first utterance's top border row and first left border (S + 1 positions no GEMM row maps to);
          // otherwise the whole value is zeroed first.
          const size_t head = (size_t)vo.grid_S + 1;
          // (whole-value clear: one more time row than the value has -- with the shared border column the 3x3 window of
          // the last bin of the last utterance's bottom border row reads position (L + 2, 0), just behind the grid)
          const size_t rows0 = st.grid_cover && L.mode != 4 ? head : (st.grid_cover ? 0 : (size_t)st.rows_out + vo.grid_S);
          if (st.compact || st.trows) {     // border positions only
            XV_HIP(h, launch_grid_zero_border(p->dev_offsets(st.lvl_out), B, st.frames_out, vo.grid_F, vo.grid_S, L.cout / 4,
                                              sb_ld(L.cout) / 4, st.out_off >= 0 ? reinterpret_cast<float*>(ws + st.out_off) : nullptr,
                                              st.out_sb_off >= 0 ? ws + st.out_sb_off : nullptr, s));
          } else if (rows0 > 0) {
            if (st.out_off >= 0) XV_HIP(h, hipMemsetAsync(ws + st.out_off, 0, rows0 * L.cout * 4, s));
            if (st.out_sb_off >= 0) XV_HIP(h, hipMemsetAsync(ws + st.out_sb_off, 0, rows0 * sb_ld(L.cout) * 4, s));
          }
        }
        if (L.mode == 1 || L.mode == 2 || L.mode == 3) {      // A addressing on the input grid (csrc/grid.hip)
          const int64_t Sin = h->values[op.in0].grid_S;
          a.a_pitch = (L.mode == 3 ? Sin : L.sw) * (int64_t)L.cin;
          a.a_off = L.mode == 1 ? (L.sw == 1 ? 0 : L.cin) : (L.mode == 2 ? (Sin + 1) * L.cin : L.cin);
          if (st.compact) {                 // rows address their window through arow (units of one grid position)
            a.arow = static_cast<const int32_t*>(p->d_rowmaps.p) + p->rowmap_off[st.arow];
            a.a_pitch = L.cin;
            a.a_off = L.mode == 1 ? 0 : (Sin + 1) * L.cin;
          }
          a.ntaps = L.mode == 1 ? 3 : 1;
          a.ktap = L.mode == 1 ? 3 * L.cin : (L.mode == 2 ? L.cin : L.Fin * L.cin);
          a.tap_stride = Sin * L.cin;
          a.cin = a.K;                      // one "frame" per A row for the kernel's tap logic (w = 1)
        }
        if (op.in1 > 0) { a.R = in_ptr(st.in1_off); a.ldr = L.cout; }
        if (st.fuse_att == 1) {             // partial scores instead of the key (model/pooling.py:189-194)
          a.Y = nullptr;
          a.att_part = reinterpret_cast<float*>(ws + st.out_off);
          a.att_ld = st.att_ld;
          a.att_heads = d.att_num_heads;
          a.att_q = static_cast<const float*>(h->query_eff.p);
        } else if (st.fuse_att == 2) {      // weighted moments instead of the value (:201-217)
          a.Y = nullptr;
          a.pool_part = reinterpret_cast<float*>(ws + st.out_off);
          a.pool_row2utt = static_cast<const int32_t*>(p->d_row2utt.p);
          a.pool_slotbase = static_cast<const int32_t*>(p->d_slotbase.p);
          a.pool_w = reinterpret_cast<const float*>(ws + st.att_w_off);
          a.pool_heads = d.att_num_heads;
          a.pool_split = d.att_split_value;
          a.pool_dvh = d.att_split_value ? L.cout / d.att_num_heads : L.cout;
          a.pool_odim = h->pool_dim / 2;
        }
        if (st.fuse_pool) {                 // statistics pooling partials instead of activations
          a.Y = nullptr;
          a.pool_part = reinterpret_cast<float*>(ws + st.out_off);
          a.pool_row2utt = static_cast<const int32_t*>(p->d_row2utt.p);
          a.pool_slotbase = static_cast<const int32_t*>(p->d_slotbase.p);
        }
        if (st.out_sb_off >= 0) {
          a.Ysb = ws + st.out_sb_off;
          a.ldsb = sb_ld(L.cout);
          if (st.out_off < 0 && !st.to_out) a.Y = nullptr;
        }
        if (L.mode == 4) {                  // conv0: 3x3 on the 1-channel input = im2col (9 taps, padded to 32) + dense GEMM
          if (st.scratch_off < 0) return fail(h, XV_ERR_STATE, "conv0 has no scratch");
          a.cin = 32; a.K = 32;             // taps 9..31 are zero in both operands
          // split precisions, final stage, not the requested node: nine fp32 FMAs per output on the vector units, BN +
          // activation + split store fused (one pass over the 200 MB output instead of im2col rows + a one-step GEMM)
          const bool direct = L.use_split && !st.to_out && st.stage == L.final_stage() && L.cout % 8 == 0 &&
                              (st.out_sb_off < 0 || L.cout % 32 == 0) && L.cout <= 256 && L.wdir.p;
          if (direct) {
            XV_HIP(h, launch_conv0_direct(feats, feat_ld, off, B, L.Fout, vo.grid_S, L.cout, st.M, static_cast<const float*>(L.wdir.p),
                                          L.act, L.d_alpha(), st.out_off >= 0 ? reinterpret_cast<float*>(ws + st.out_off) : nullptr,
                                          st.out_sb_off >= 0 ? ws + st.out_sb_off : nullptr, sb_ld(L.cout), f16,
                                          static_cast<int*>(h->ovf_flag.p), a.sb_mul, s));
            break;
          }
          if (L.use_split) {
            XV_HIP(h, launch_im2col2d_sb(feats, feat_ld, off, B, L.Fout, vo.grid_S, st.M, ws + st.scratch_off, f16, static_cast<int*>(h->ovf_flag.p), s));
            a.Xsb = ws + st.scratch_off; a.ldsbx = 32; a.Wsb = L.wsb.p; a.Wfr = L.wfr.p;
            XV_HIP(h, launch_gemm_bf16x3(a, s));
          } else {
            XV_HIP(h, launch_im2col2d_f32(feats, feat_ld, off, B, L.Fout, vo.grid_S, st.M, reinterpret_cast<float*>(ws + st.scratch_off), s));
            a.X = reinterpret_cast<const float*>(ws + st.scratch_off); a.ldx = 32;
            XV_HIP(h, launch_gemm_f32(a, true, s));
          }
          if (st.unpad_to_out)
            XV_HIP(h, launch_grid_unpad_n(reinterpret_cast<const float*>(ws + st.out_off), off_out, B, vo.grid_F, vo.grid_S, vo.cols,
                                          st.frames_out, out, s));
          break;
        }
        if (L.im2col) {
          // 30-dim first layer on the split kernel: materialise the w*cin-wide rows once (SB
          // format, K padded to 32), then it is a dense layer on those rows
          if (st.scratch_off < 0) return fail(h, XV_ERR_STATE, "im2col layer has no scratch");
          if (L.cin_pad) {
            // the 30-dim feature rows become SB rows of one 32-channel block (9.8 MB for 256 x 300 frames instead of the
            // 48 MB of materialised 5-frame rows); the layer is then an ordinary 5-tap convolution over them
            XV_HIP(h, launch_im2col_sb(feats, feat_ld, L.cin, 1, st.rows_in, ws + st.scratch_off, L.cin_pad, f16,
                                       static_cast<int*>(h->ovf_flag.p), s));
            a.Xsb = ws + st.scratch_off;
            a.ldsbx = L.cin_pad;
            a.cin = L.cin_pad;
            a.K = L.w * L.cin_pad;
            a.Wsb = L.wsb.p; a.Wfr = L.wfr.p;
            a.ysb_f6 = st.out_f6 ? 1 : 0;
            XV_HIP(h, launch_gemm_bf16x3(a, s));
            break;
          }
          XV_HIP(h, launch_im2col_sb(feats, feat_ld, L.cin, L.w, st.M, ws + st.scratch_off, L.Kpad, f16, static_cast<int*>(h->ovf_flag.p), s));
          a.Xsb = ws + st.scratch_off;
          a.ldsbx = L.Kpad;
          a.cin = a.K;
          a.Wsb = L.wsb.p; a.Wfr = L.wfr.p;
          XV_HIP(h, launch_gemm_bf16x3(a, s));
          break;
        }
        if (L.use_f6) {                     // two-unit split: convert the split-blocked input, then the f16 + fp6 kernel
          if (st.in0_sb_off < 0 || st.scratch_off < 0) return fail(h, XV_ERR_STATE, "f16f6 layer %s has no input / scratch", L.kernel_name.c_str());
          if (st.in_f6) {                   // the producer wrote this layer's block format
            a.Xsb = ws + st.in0_sb_off;
          } else {
            XV_HIP(h, launch_f6_from_sb(ws + st.in0_sb_off, ws + st.scratch_off, st.rows_in, sb_ld(L.cin) / 32, s));
            a.Xsb = ws + st.scratch_off;
          }
          a.ysb_f6 = st.out_f6 ? 1 : 0;
          // taps w .. 7 of a scaled MFMA have zero weights but still read rows m + w .. m + 7: behind the last input row that is
          // whatever the buffer held, and an E8M0 scale byte of 255 there is a NaN (NaN x 0 = NaN) -- eight rows are kept defined:
          // by the producer's epilogue when it wrote this format itself, by a memset behind the conversion pass otherwise
          if (!st.in_f6)
            XV_HIP(h, hipMemsetAsync(const_cast<char*>(static_cast<const char*>(a.Xsb)) + st.rows_in * (int64_t)sb_ld(L.cin) * 4, 0,
                                     (size_t)8 * sb_ld(L.cin) * 4, s));
          a.ldsbx = sb_ld(L.cin);
          a.Wfr = L.wf6m.p;
          a.Wx6 = L.wf6x.p;
          if (st.trows) {
            // 3 x 3 on the zero-bordered grid as three taps along time: a GEMM row = one padded time row of Sin positions, the window
            // of frequency bin j = the 3 cin contiguous channels from position j on; output row rowmap[m] + j of the value advanced by
            // one position (csrc/grid.hip, rowmap_trows_kernel)
            const int64_t Sin = h->values[op.in0].grid_S;
            a.a_pitch = 0; a.a_off = 0; a.arow = nullptr; a.ntaps = 0; a.ktap = 0; a.tap_stride = 0;
            a.ldsbx = Sin * L.cin;
            a.cin = 3 * L.cin;
            a.K = 9 * L.cin;
            a.nbin = L.Fout;
            a.bin_x_bytes = (int64_t)L.cin * 4;
            if (a.Ysb) a.Ysb = static_cast<char*>(a.Ysb) + (int64_t)a.ldsb * 4;
            if (a.Y) a.Y += a.ldy;
            if (a.R) a.R += a.ldr;
          }
          if (st.tail_mt > 0 && st.scratch2_off >= 0) {
            a.tail_mt = st.tail_mt;
            a.ksplit = st.ksplit;
            a.partial = reinterpret_cast<float*>(ws + st.scratch2_off);
          }
          XV_HIP(h, launch_gemm_f16f6(a, s));
          if (st.unpad_to_out)
            XV_HIP(h, launch_grid_unpad_n(reinterpret_cast<const float*>(ws + st.out_off), off_out, B, vo.grid_F, vo.grid_S, vo.cols,
                                          st.frames_out, out, s));
          break;
        }
        if (L.use_split) {
          if (st.in0_sb_off < 0) return fail(h, XV_ERR_STATE, "split layer %s has no split-blocked input", L.kernel_name.c_str());
          a.Xsb = ws + st.in0_sb_off;
          a.ldsbx = L.mode == 0 ? sb_ld(L.cin) : 0;
          a.Wsb = L.wsb.p; a.Wfr = L.wfr.p;
          a.ysb_f6 = st.out_f6 ? 1 : 0;
          if (st.tail_mt > 0 && st.scratch_off >= 0) {
            a.tail_mt = st.tail_mt;
            a.ksplit = st.ksplit;
            a.partial = reinterpret_cast<float*>(ws + st.scratch_off);
          }
          XV_HIP(h, launch_gemm_bf16x3(a, s));
          if (st.unpad_to_out)
            XV_HIP(h, launch_grid_unpad_n(reinterpret_cast<const float*>(ws + st.out_off), off_out, B, vo.grid_F, vo.grid_S, vo.cols,
                                          st.frames_out, out, s));
          break;
        }
        if (st.ksplit > 1 && st.scratch_off >= 0) {
          a.ksplit = st.ksplit;
          a.partial = reinterpret_cast<float*>(ws + st.scratch_off);
        }
        const bool aligned = op.in0 != 0 && (a.ldx % 4 == 0) && (a.K % 4 == 0);
        if (L.mode != 0 && !aligned)
          return fail(h, XV_ERR_UNSUPPORTED, "resnet convolution %s needs channel counts that are multiples of 4", L.kernel_name.c_str());
        XV_HIP(h, launch_gemm_f32(a, aligned, s));
        if (st.unpad_to_out)
          XV_HIP(h, launch_grid_unpad_n(reinterpret_cast<const float*>(ws + st.out_off), off_out, B, vo.grid_F, vo.grid_S, vo.cols,
                                        st.frames_out, out, s));
        break;
      }
      case OP_STAT_POOL: {
        const Value& vi = h->values[op.in0];
        if (st.fuse_pool) {
          XV_HIP(h, launch_pool_finalize(in_ptr(st.in0_off), vi.cols, off_in, B, vi.ctx,
                                         static_cast<const int32_t*>(p->d_slotbase.p), optr, 2 * vi.cols, s));
          break;
        }
        XV_HIP(h, launch_stat_pool(in_ptr(st.in0_off), vi.cols, vi.cols, off_in, B, vi.ctx, optr, 2 * vi.cols, s));
        break;
      }
      case OP_ATT_SCORES: {
        const Value& vi = h->values[op.in0];
        const float scale = d.att_use_scale ? 1.0f / std::sqrt((float)h->att_dk_h) : 1.0f;   // model/pooling.py:193-194
        if (st.fuse_att) {
          XV_HIP(h, launch_att_scores_reduce(in_ptr(st.in0_off), st.att_ld, h->key_npad / 32, d.att_num_heads, st.rows_in,
                                             scale, reinterpret_cast<float*>(ws + st.out_off), s));
          break;
        }
        XV_HIP(h, launch_att_scores(in_ptr(st.in0_off), vi.cols, st.rows_in, static_cast<const float*>(h->query.p),
                                    d.att_num_heads, h->att_dk_h, d.att_split_key, scale,
                                    reinterpret_cast<float*>(ws + st.out_off), s));
        break;
      }
      case OP_ATT_SOFTMAX: {
        float* sc = reinterpret_cast<float*>(ws + st.out_off);
        XV_HIP(h, launch_att_softmax(sc, d.att_num_heads, off, B, h->final_ctx, s));
        if (st.fuse_att)
          XV_HIP(h, launch_att_slot_sums(sc, d.att_num_heads, off, B, h->final_ctx, static_cast<const int32_t*>(p->d_slotbase.p),
                                         reinterpret_cast<float*>(ws + st.att_s0_off), s));
        if (st.to_out) XV_HIP(h, launch_att_weights_out(sc, d.att_num_heads, off, B, h->final_ctx, out, s));
        break;
      }
      case OP_ATT_POOL: {
        const Value& vv = h->values[op.in0];
        if (st.fuse_att) {
          XV_HIP(h, launch_att_pool_finalize(in_ptr(st.in0_off), reinterpret_cast<const float*>(ws + st.att_s0_off),
                                             h->pool_dim / 2, d.att_num_heads, d.att_split_value ? vv.cols / d.att_num_heads : vv.cols,
                                             d.att_split_value, off, B, vv.ctx, static_cast<const int32_t*>(p->d_slotbase.p),
                                             optr, h->pool_dim, s));
          break;
        }
        XV_HIP(h, launch_att_pool(in_ptr(st.in0_off), vv.cols, vv.cols, in_ptr(st.in1_off), d.att_num_heads,
                                  d.att_split_value, off, B, vv.ctx, optr, h->pool_dim, s));
        break;
      }
      case OP_AFFINE_ACT: {
        const int n = h->pool_dim;
        const float* vec = static_cast<const float*>(h->post_vec.p);
        const int a = st.stage >= 2 ? act_of(d) : ACT_NONE;
        XV_HIP(h, launch_affine_act(in_ptr(st.in0_off), n, B, n, vec, vec + n,
                                    (a == ACT_PRELU) ? vec + 2 * n : nullptr, a, optr, n, s));
        break;
      }
      case OP_GRID_MAXPOOL: {
        const Value& vo = h->values[op.out];
        float* y = st.out_off >= 0 ? reinterpret_cast<float*>(ws + st.out_off) : nullptr;
        XV_HIP(h, launch_grid_maxpool3x3(in_ptr(st.in0_off), off, B, vo.grid_F, vo.grid_S, vo.cols, st.rows_out, y,
                                         st.out_sb_off >= 0 ? ws + st.out_sb_off : nullptr, sb_ld(vo.cols), f16,
                                         static_cast<int*>(h->ovf_flag.p), std::ldexp(1.f, vo.sb_exp), s));
        if (st.unpad_to_out)
          XV_HIP(h, launch_grid_unpad_n(y, off_out, B, vo.grid_F, vo.grid_S, vo.cols, st.frames_out, out, s));
        break;
      }
      case OP_L2_SCALE: {
        XV_HIP(h, launch_l2_scale(in_ptr(st.in0_off), B, h->values[op.out].cols, d.feature_scaling_factor, optr, s));
        break;
      }
      default:
        return fail(h, XV_ERR_STATE, "unknown op kind %d", op.kind);
    }
    if (prof_step) {
      XV_HIP(h, hipEventRecord(pe1, s));
      std::lock_guard<std::mutex> lk(h->prof_mu);
      h->prof_recs.push_back({pe0, pe1, p, (int)si});
    }
  }
  return XV_OK;
}

int xv_forward(xv_handle* h, const xv_plan* p, const float* feats_dev, int feat_ld, float* out_dev, int64_t out_capacity,
               void* workspace, int64_t workspace_bytes, void* stream) {
  return run_plan(h, p, feats_dev, feat_ld, out_dev, out_capacity, workspace, workspace_bytes,
                  static_cast<hipStream_t>(stream));
}

int xv_profile_begin(xv_handle* h, int max_events) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_profile_begin: null handle");
  if (max_events < 2) return fail(h, XV_ERR_INVALID, "xv_profile_begin: max_events < 2");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->prof_mu);
  while ((int)h->prof_pool.size() < max_events) {
    hipEvent_t e;
    XV_HIP(h, hipEventCreate(&e));
    h->prof_pool.push_back(e);
  }
  h->prof_next = 0;
  h->prof_recs.clear();
  h->prof_forwards = 0;
  h->profiling = true;
  return XV_OK;
}

int xv_profile_end(xv_handle* h, xv_kernel_time* entries, int max_entries, int* n_forwards) {
  if (!h) return fail(nullptr, XV_ERR_INVALID, "xv_profile_end: null handle");
  if (!h->profiling) return fail(h, XV_ERR_STATE, "xv_profile_end without xv_profile_begin");
  if (!entries || max_entries < 1) return fail(h, XV_ERR_INVALID, "xv_profile_end: no entry buffer");
  DeviceGuard g(h->device);
  std::lock_guard<std::mutex> lk(h->prof_mu);
  h->profiling = false;
  int n = 0;
  std::vector<int> count;
  for (const auto& r : h->prof_recs) {
    XV_HIP(h, hipEventSynchronize(r.e1));
    float ms = 0.f;
    XV_HIP(h, hipEventElapsedTime(&ms, r.e0, r.e1));
    const PlanStep& st = r.plan->steps[r.step];
    const char* name = step_name(h, st);
    int i = 0;
    for (; i < n; ++i)
      if (strncmp(entries[i].name, name, sizeof(entries[i].name) - 1) == 0 && entries[i].flops == st.flops) break;
    if (i == n) {
      if (n >= max_entries) continue;
      memset(&entries[n], 0, sizeof(xv_kernel_time));
      snprintf(entries[n].name, sizeof(entries[n].name), "%s", name);
      entries[n].flops = st.flops;
      entries[n].bytes = st.bytes;
      count.push_back(0);
      ++n;
    }
    entries[i].ms += ms;
    count[i]++;
  }
  for (int i = 0; i < n; ++i) {
    entries[i].launches = count[i];
    if (count[i] > 0) entries[i].ms /= (float)count[i];
  }
  if (n_forwards) *n_forwards = h->prof_forwards;
  h->prof_recs.clear();
  return n;
}

int xv_frontend_cmn_select(int device, const float* feats_dev, int ld, int dim, const int32_t* frame_offsets_dev,
                           int batch, const int32_t* src_rows_dev, int64_t out_rows, int cmn_window, int center,
                           int min_window, double* scratch_dev, float* out_dev, void* stream) {
  if (!feats_dev || !frame_offsets_dev || !src_rows_dev || !scratch_dev || !out_dev)
    return fail(nullptr, XV_ERR_INVALID, "xv_frontend_cmn_select: null pointer");
  if (dim < 1 || dim > 1024 || ld < dim || batch < 1 || out_rows < 0 || cmn_window < 0)
    return fail(nullptr, XV_ERR_INVALID, "xv_frontend_cmn_select: bad dimensions");
  DeviceGuard g(device);
  if (!g.ok) return fail(nullptr, XV_ERR_HIP, "cannot select HIP device %d", device);
  const hipError_t e = launch_cmn_select(feats_dev, ld, dim, frame_offsets_dev, batch, scratch_dev, src_rows_dev, out_rows,
                                         cmn_window, center, min_window, out_dev, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(nullptr, XV_ERR_HIP, "cmn_select launch failed: %s", hipGetErrorString(e));
  return XV_OK;
}

int xv_length_normalize(int device, const float* x_dev, int64_t ldx, int64_t rows, int dim, int scaleup, float* out_dev,
                        int64_t ldo, void* stream) {
  if (!x_dev || !out_dev) return fail(nullptr, XV_ERR_INVALID, "xv_length_normalize: null pointer");
  if (rows < 0 || dim < 1 || ldx < dim || ldo < dim) return fail(nullptr, XV_ERR_INVALID, "xv_length_normalize: bad dimensions");
  DeviceGuard g(device);
  if (!g.ok) return fail(nullptr, XV_ERR_HIP, "cannot select HIP device %d", device);
  const hipError_t e = launch_length_norm(x_dev, ldx, rows, dim, scaleup, out_dev, ldo, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(nullptr, XV_ERR_HIP, "length_norm launch failed: %s", hipGetErrorString(e));
  return XV_OK;
}

int xv_speaker_mean(int device, const float* x_dev, int64_t ldx, int dim, const int32_t* spk_offsets_dev,
                    const int32_t* utt_index_dev, int64_t num_speakers, float* out_dev, int64_t ldo, void* stream) {
  if (!x_dev || !out_dev || !spk_offsets_dev || !utt_index_dev) return fail(nullptr, XV_ERR_INVALID, "xv_speaker_mean: null pointer");
  if (num_speakers < 0 || dim < 1 || ldx < dim || ldo < dim) return fail(nullptr, XV_ERR_INVALID, "xv_speaker_mean: bad dimensions");
  DeviceGuard g(device);
  if (!g.ok) return fail(nullptr, XV_ERR_HIP, "cannot select HIP device %d", device);
  const hipError_t e = launch_speaker_mean(x_dev, ldx, dim, spk_offsets_dev, utt_index_dev, num_speakers, out_dev, ldo,
                                           static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(nullptr, XV_ERR_HIP, "speaker_mean launch failed: %s", hipGetErrorString(e));
  return XV_OK;
}

void xv_destroy(xv_handle* h) {
  if (!h) return;
  {
    DeviceGuard g(h->device);
    for (auto& L : h->layers) { L.wt.release(); L.wsb.release(); L.wfr.release(); L.vec.release(); L.wf6m.release(); L.wf6x.release(); L.wdir.release(); }
    h->query.release();
    h->ovf_flag.release();
    h->query_eff.release();
    h->post_vec.release();
    for (auto& b : h->pool) b.release();
    for (auto e : h->prof_pool) (void)hipEventDestroy(e);
  }
  delete h;
}

}  // extern "C"
