"""Host-side mirror of the reference's predict surface: `Trainer(params, model_dir, dim,
single_cpu=True)`, `.build("predict")`, `.predict(features)`, `.close()`
(model/trainer.py:87-219, 309-338, 379-383, 277-295, 886-913, 270-275), so that
egs/voxceleb/v1/nnet/lib/extract.py:56-57,81-89,95 runs unchanged against it.

Where the reference builds a TF graph and calls sess.run, this class creates an
`xv_handle` in libxvec_hip.so (include/xvec_hip.h), uploads the checkpoint variables by
their TensorFlow names and runs the HIP kernels.  PyTorch is used only for device memory
and the stream.  There is no CPU path: without the HIP library or a GPU this raises.
"""
import collections
import ctypes as C
import os
import sys

import numpy as np

from . import _lib
from . import model_io

_PRECISIONS = {"f32": _lib.XV_PREC_F32, "bf16x3": _lib.XV_PREC_BF16X3, "f16x3": _lib.XV_PREC_F16X3, "f16f6": _lib.XV_PREC_F16F6}
_F16_RANGE = ("f16x3", "f16f6")       # precisions whose activations must stay within the fp16 range
# "bf16x3": split-precision MFMA over the full fp32 range, ~2e-6 rel-L2 on the x-vector (bar 1e-4);
# "f16x3": the same kernels on fp16 hi/lo halves, ~3e-7, inputs / activations must stay within +-65504; "f32": exact fp32 MFMA;
# "f16f6": f16x3 with the multi-tap convolutions on the two-unit split (f16 hi*hi + block-scaled fp6 cross terms), ~1e-5
DEFAULT_PRECISION = "bf16x3"


def default_precision(network_type):
    """What Trainer, the command-line drivers and bench.py run when no precision is named: f16f6 for every network -- the fastest
    precision, under the same GPU parity suite as the others (tests/test_gpu_parity.py: every endpoint of the three networks,
    both poolings, ragged batches, the rescaled-layer and feature-range tests), with both ends of the fp16 range guarded (values
    beyond +-65504 and feature batches below 2^-8 never produce wrong vectors: Trainer and the command-line driver run such a batch again in bf16x3).  Its two-unit
    kernel covers the 5 / 7 / 9-tap convolutions of the (extended) TDNN and the stride-1 3 x 3 convolutions of the ResNet stages of
    128 channels and more; "bf16x3" (full fp32 exponent range, no such refusals) stays one argument away -- and is what a defaulted
    Trainer uses for a model none of whose layers the two-unit kernel applies to (channel counts not multiples of 128)."""
    return "f16f6" if network_type in ("tdnn", "extended_tdnn", "resnet_18") else DEFAULT_PRECISION


def _relu_type(params):
    t = params.dict.get("network_relu_type")
    return {"prelu": _lib.XV_ACT_PRELU, "lrelu": _lib.XV_ACT_LRELU}.get(t, _lib.XV_ACT_RELU)


_NETWORK_TYPES = {"tdnn": 0, "extended_tdnn": 1, "resnet_18": 2}          # XV_NET_* (model/trainer.py:100-110)


def _endpoint_index(name, what):
    """'tdnn4_relu' -> 4.  Attention key/value inputs must be the relu endpoint of a frame-level
    layer at full temporal context (the library checks the index against the graph)."""
    import re
    m = re.match(r"^tdnn(\d+)_relu$", name)
    if not m:
        raise NotImplementedError("%s=%r: only tdnn<N>_relu endpoints are supported" % (what, name))
    return int(m.group(1))


class Trainer(object):
    """Predict-only counterpart of model/trainer.py `Trainer`."""

    def __init__(self, params, model_dir, dim, num_speakers=None, single_cpu=False, num_gpus=1,
                 device=None, precision=None, range_fallback=None):
        # model/trainer.py:100-110 network dispatch.  Only the TDNN is on this round's hot path.
        self.network_type = params.network_type
        if params.network_type not in _NETWORK_TYPES:
            if params.network_type == "tdnn-s":
                raise NotImplementedError("tdnn-s is dead code in the reference (quit() at model/tdnn.py:201-202)")
            raise NotImplementedError("Not implement %s network" % params.network_type)
        self.params = params
        self.model = os.path.join(model_dir, "nnet") if model_dir is not None else None
        self.dim = int(dim)
        self.num_speakers = num_speakers
        self.is_built = False
        self.is_loaded = False
        self.first_feature_split_alert = True
        self.embeddings = None            # name of the endpoint predict() returns
        self._precision = precision or os.environ.get("XVEC_PRECISION") or default_precision(params.network_type)
        self._precision_defaulted = not (precision or os.environ.get("XVEC_PRECISION"))
        if self._precision not in _PRECISIONS:
            raise ValueError("precision must be one of %s" % sorted(_PRECISIONS))
        # The reference (fp32 TensorFlow) accepts any finite features.  The fp16 precisions refuse batches outside their range; with
        # range_fallback (default on; XVEC_RANGE_FALLBACK=0 or range_fallback=False turn it off) predict / predict_list / collect
        # then run THAT batch again on a bf16x3 twin of this model (built on first need) instead of raising.
        if range_fallback is None:
            range_fallback = os.environ.get("XVEC_RANGE_FALLBACK", "1") != "0"
        self._range_fallback = bool(range_fallback) and self._precision in _F16_RANGE
        self._fallback = None
        self._weights_host = None
        self._single_cpu = single_cpu
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        self._device_index = int(device)
        self._lib = None
        self._h = None
        self._plans = collections.OrderedDict()     # (node, offsets bytes) -> (plan, info), least recently used first
        self._plan_cache_size = 128
        self._pinned_plans = set()                    # keys of plans a captured hipGraph refers to: never evicted
        self._ws = None
        self._ws_pinned = False                       # a captured graph holds the workspace pointer: no regrow
        self._options = {}
        self._step = None
        self._torch = None
        self._pinned = None
        self._slots = None                            # pipelined host interface (submit_list / collect): staging slots
        self._copy_stream = None
        self._slot_turn = 0

    # ------------------------------------------------------------------ graph build
    def build(self, mode, noupdate_var_list=None):
        """model/trainer.py:309-338.  Only `predict` exists here (training is out of scope)."""
        assert mode == "train" or mode == "valid" or mode == "predict"
        if mode != "predict":
            raise NotImplementedError("only build('predict') is implemented (extraction path)")
        p = self.params
        # defaults the reference writes into params while building (model/tdnn.py:114-116,152-154,163-164,173-174)
        if "num_nodes_pooling_layer" not in p.dict:
            p.dict["num_nodes_pooling_layer"] = 1500
        if "num_nodes_last_layer" not in p.dict:
            p.dict["num_nodes_last_layer"] = 512
        if "last_layer_no_bn" not in p.dict:
            p.dict["last_layer_no_bn"] = False
        if "last_layer_linear" not in p.dict:
            p.dict["last_layer_linear"] = False
        if p.pooling_type not in ("statistics_pooling", "self_attention"):
            raise NotImplementedError("Not implement %s pooling" % p.pooling_type)     # model/pooling.py:23
        if p.network_type == "resnet_18":
            if self.dim != 40:
                raise AssertionError("resnet_18 needs 40-dim features (model/resnet.py:190)")
            if "resnet_blocks" not in p.dict:
                p.dict["resnet_blocks"] = [2, 2, 2, 2]                                  # model/resnet.py:203-204
            if p.pooling_type != "statistics_pooling":
                raise NotImplementedError("resnet_18 registers no frame-level endpoints for attention (model/resnet.py)")
        if "feature_norm" in p.dict and p.feature_norm:
            assert "feature_scaling_factor" in p.dict, \
                "If feature normalization is applied, scaling factor is necessary."      # trainer.py:401
        self.embeddings = p.embedding_node                                              # trainer.py:380
        self.is_built = True

    def set_embedding(self, embedding_node):
        """model/trainer.py:305-307."""
        self.params.embedding_node = embedding_node
        self.embeddings = embedding_node

    def _make_desc(self, channels):
        p = self.params
        d = _lib.ModelDesc()
        d.struct_size = C.sizeof(_lib.ModelDesc)
        d.network_type = _NETWORK_TYPES[p.network_type]
        d.feat_dim = self.dim
        d.channels = int(channels)
        d.pooling_type = _lib.XV_POOL_SELF_ATTENTION if p.pooling_type == "self_attention" else _lib.XV_POOL_STATISTICS
        d.relu_type = _relu_type(p)
        d.num_nodes_pooling_layer = int(p.dict["num_nodes_pooling_layer"])
        d.num_nodes_last_layer = int(p.dict["num_nodes_last_layer"])
        d.last_layer_no_bn = int(bool(p.dict["last_layer_no_bn"]))
        d.last_layer_linear = int(bool(p.dict["last_layer_linear"]))
        d.feature_norm = int(bool(p.dict.get("feature_norm", False)))
        d.feature_scaling_factor = float(p.dict.get("feature_scaling_factor", 1.0))
        d.precision = _PRECISIONS[self._precision]
        for i, n in enumerate(p.dict.get("resnet_blocks", [2, 2, 2, 2])):
            d.resnet_blocks[i] = int(n)
        d.resnet_maxpooling = int(bool(p.dict.get("resnet_maxpooling", False)))
        d.resnet_time_stride = int(bool(p.dict.get("resnet_time_stride", False)))
        if p.pooling_type == "self_attention":
            kn = list(p.att_key_num_nodes)
            vn = list(p.att_value_num_nodes)
            if not 1 <= len(kn) <= _lib.XV_MAX_ATT_LAYERS or len(vn) > _lib.XV_MAX_ATT_LAYERS:
                raise NotImplementedError("attention key/value networks deeper than %d layers" % _lib.XV_MAX_ATT_LAYERS)
            d.att_key_input = _endpoint_index(p.att_key_input, "att_key_input")
            d.att_value_input = _endpoint_index(p.att_value_input, "att_value_input")
            d.att_num_key_layers = len(kn)
            for i, n in enumerate(kn):
                d.att_key_num_nodes[i] = int(n)
            d.att_key_network_type = int(p.att_key_network_type)
            d.att_num_value_layers = len(vn)
            for i, n in enumerate(vn):
                d.att_value_num_nodes[i] = int(n)
            d.att_value_network_type = int(p.att_value_network_type)
            d.att_apply_nonlinear = int(bool(p.att_apply_nonlinear))
            d.att_use_scale = int(bool(p.att_use_scale))
            d.att_num_heads = int(p.att_num_heads)
            d.att_split_value = int(bool(p.att_split_value))
            d.att_split_key = int(bool(p.att_split_key))
        return d

    # ------------------------------------------------------------------ weights
    def load(self):
        """model/trainer.py:277-295: restore the checkpoint named by <model>/nnet/checkpoint."""
        if not self.is_built:
            sys.exit("The graph has not been build. Cannot load the models.")
        weights, step = (None, None)
        if self.model is not None:
            weights, step = model_io.load_weights(self.model)
        if weights is None:
            sys.exit("Failed to find a checkpoint in {}".format(self.model))
        self.load_weights(weights, step)
        return step

    def load_weights(self, weights, step=0):
        """Upload a {TF variable name: array} dict (what Saver.restore does from a checkpoint)."""
        if not self.is_built:
            sys.exit("The graph has not been build. Cannot load the models.")
        import torch
        if not torch.cuda.is_available():
            raise RuntimeError("no HIP device visible: the x-vector path has no CPU fallback")
        self._torch = torch
        self._lib = _lib.load()
        self._release()
        if self.params.network_type == "resnet_18":
            k1 = np.asarray(weights["resnet_18/conv0_1/kernel"])          # [3,3,1,width]
        else:
            scope = "etdnn" if self.params.network_type == "extended_tdnn" else "tdnn"
            k1 = np.asarray(weights[scope + "/tdnn1_conv/kernel"])
        # A defaulted f16f6 only pays where its two-unit kernel runs: layers whose channel counts are multiples of 128 (the kernel
        # takes channel blocks in quads).  A narrower model would get the f16x3 kernels -- the fp16 range limits for nothing -- so
        # it runs in bf16x3 instead.  (ResNet stage widths are w, 2w, 4w, 8w.)
        if self._precision_defaulted and self._precision == "f16f6":
            widest = int(k1.shape[-1]) * (8 if self.params.network_type == "resnet_18" else 1)
            if widest % 128 != 0:
                self._precision = DEFAULT_PRECISION
                self._range_fallback = False
        desc = self._make_desc(channels=k1.shape[-1])
        h = C.c_void_p()
        _lib.check(self._lib.xv_create(C.byref(desc), self._device_index, C.byref(h)))
        self._h = h
        for name, value in self._options.items():
            _lib.check(self._lib.xv_set_option(self._h, name.encode(), value), self._h)
        try:
            unused = []
            for name, arr in weights.items():
                a = np.ascontiguousarray(np.asarray(arr), dtype=np.float32)
                shape = (C.c_int64 * a.ndim)(*a.shape)
                rc = self._lib.xv_set_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), shape, a.ndim)
                if rc < 0:
                    msg = self._lib.xv_last_error(self._h).decode()
                    if "not part of the predict graph" in msg:
                        unused.append(name)        # e.g. softmax/output layer, optimizer slots
                        continue
                    raise _lib.XvError(rc, msg)
            _lib.check(self._lib.xv_finalize(self._h), self._h)
        except Exception:
            self._release()
            raise
        self._unused_variables = unused
        self._step = step
        self._weights_host = weights if self._range_fallback else None
        self.is_loaded = True

    # ------------------------------------------------------------------ device-level predict
    def set_option(self, name, value):
        """Execution option of the library (xv_set_option: "pool_fusion", "tail_split"); affects plans created later."""
        self._options[name] = int(value)
        if self._h is not None:
            _lib.check(self._lib.xv_set_option(self._h, name.encode(), int(value)), self._h)
            if name.startswith("profile"):
                return                                   # does not change what a plan computes
            for k in [k for k in self._plans if k not in self._pinned_plans]:
                self._lib.xv_plan_destroy(self._plans.pop(k)[0])

    def _plan(self, offsets, node, pin=False):
        key = (node, offsets.tobytes())
        ent = self._plans.get(key)
        if ent is not None:
            self._plans.move_to_end(key)
            if pin:
                self._pinned_plans.add(key)
            return ent
        nid = self._lib.xv_node_id(self._h, node.encode())
        if nid < 0:
            raise KeyError(node)                       # endpoints[params.embedding_node] (trainer.py:380)
        torch = self._torch
        off = np.ascontiguousarray(offsets, dtype=np.int32)
        plan = C.c_void_p()
        stream = torch.cuda.current_stream(self._device_index).cuda_stream
        rc = self._lib.xv_plan_create(self._h, off.ctypes.data_as(C.c_void_p), len(off) - 1, nid,
                                      C.c_void_p(stream), C.byref(plan))
        if rc == _lib.XV_ERR_TOO_SHORT:
            raise ValueError(self._lib.xv_last_error(self._h).decode())
        _lib.check(rc, self._h)
        info = _lib.PlanInfo()
        _lib.check(self._lib.xv_plan_query(plan, C.byref(info)))
        # bounded LRU cache of batch geometries; the device arrays of an evicted plan go back to the library's pool
        # and are handed to the next plan in stream order (a ragged ark stream makes one plan per batch)
        while len(self._plans) >= self._plan_cache_size:
            victim = next((k for k in self._plans if k not in self._pinned_plans), None)
            if victim is None:
                break
            self._lib.xv_plan_destroy(self._plans.pop(victim)[0])
        self._plans[key] = (plan, info)
        if pin:
            self._pinned_plans.add(key)
        return plan, info

    def _workspace(self, nbytes, pin=False):
        torch = self._torch
        if self._ws is None or self._ws.numel() < nbytes:
            if self._ws_pinned:
                raise RuntimeError("the workspace is referenced by a captured hipGraph and cannot grow from %d to %d bytes; "
                                   "capture the largest geometry first or release the graph (Trainer.release_graphs)"
                                   % (self._ws.numel(), nbytes))
            self._ws = None
            # head-room so that a stream of ragged batches of one size class does not regrow it batch after batch
            self._ws = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device="cuda:%d" % self._device_index)
        if pin:
            self._ws_pinned = True
        return self._ws

    def release_graphs(self):
        """Declare every hipGraph captured from this trainer dead: their plans may be evicted and the workspace may
        grow again."""
        self._pinned_plans.clear()
        self._ws_pinned = False

    def predict_packed(self, feats_dev, offsets, node=None, out=None):
        """Device-resident entry: `feats_dev` is a float32 CUDA tensor [total_frames, ld] holding
        the utterances back to back, `offsets` the B+1 frame offsets.  Returns a CUDA tensor
        [B, E] (segment-level node) or [total_out_frames, E] (frame-level node), enqueued on
        the current stream (`out` may be a preallocated result tensor)."""
        if not self.is_loaded:
            self._lazy_load()
        torch = self._torch
        node = node or self.embeddings
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        assert feats_dev.is_cuda and feats_dev.dtype == torch.float32 and feats_dev.dim() == 2
        assert feats_dev.is_contiguous()
        assert feats_dev.shape[0] == int(offsets[-1]), "offsets[-1] must equal the number of packed frames"
        if feats_dev.shape[1] < self.dim:
            raise ValueError("features have %d columns, the network needs %d" % (feats_dev.shape[1], self.dim))
        plan, info = self._plan(offsets, node)
        ws = self._workspace(info.workspace_bytes + 256)
        if out is None:
            out = torch.empty((int(info.out_rows), int(info.out_cols)), dtype=torch.float32, device=feats_dev.device)
        # `out` may also be PINNED host memory (device-accessible): the last kernel then writes the embeddings straight into it and
        # no copy sits between two forwards (a D2H copy on the compute stream is an engine switch: tools/pipeline_probe.py)
        assert (out.is_cuda or out.is_pinned()) and out.dtype == torch.float32 and out.is_contiguous()
        stream = torch.cuda.current_stream(self._device_index).cuda_stream
        args = (self._h, plan, C.c_void_p(feats_dev.data_ptr()), int(feats_dev.shape[1]),
                C.c_void_p(out.data_ptr()), out.numel(), C.c_void_p(ws.data_ptr()), ws.numel(), C.c_void_p(stream))
        _lib.check(self._lib.xv_forward(*args), self._h)
        return out

    def capture_graph(self, feats_dev, offsets, node=None, out=None):
        """Capture one forward (xv_forward only enqueues kernels: no allocation, no host sync) into a
        hipGraph and return (graph, out).  `graph.replay()` re-runs it on the current stream with the
        same buffers; refill `feats_dev` in place between replays."""
        torch = self._torch if self._torch is not None else __import__("torch")
        if not self.is_loaded:
            self._lazy_load()
        node = node or self.embeddings
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        # plan + workspace exist before the capture starts, and stay: the graph holds raw pointers to the plan's
        # device arrays and to the workspace, so the plan is exempt from eviction and the workspace from regrowth
        _, info = self._plan(offsets, node, pin=True)
        self._workspace(info.workspace_bytes + 256, pin=True)
        if out is None:
            out = torch.empty((int(info.out_rows), int(info.out_cols)), dtype=torch.float32, device=feats_dev.device)
        self.predict_packed(feats_dev, offsets, node, out=out)         # warm-up outside the capture
        torch.cuda.synchronize(feats_dev.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            self.predict_packed(feats_dev, offsets, node, out=out)
        return graph, out

    def profile_begin(self, max_events=4096):
        """Start per-kernel hipEvent timing of the following predict calls (bench.py)."""
        if not self.is_loaded:
            self._lazy_load()
        _lib.check(self._lib.xv_profile_begin(self._h, int(max_events)), self._h)

    def profile_end(self):
        """-> (records, n_forwards); records: dicts name/ms/launches/flops/bytes per kernel."""
        ent = (_lib.KernelTime * 64)()
        nf = C.c_int(0)
        n = _lib.check(self._lib.xv_profile_end(self._h, ent, 64, C.byref(nf)), self._h)
        recs = [dict(name=ent[i].name.decode(), ms=float(ent[i].ms), launches=int(ent[i].launches),
                     flops=int(ent[i].flops), bytes=int(ent[i].bytes)) for i in range(n)]
        return recs, int(nf.value)

    def plan_info(self, offsets, node=None):
        if not self.is_loaded:
            self._lazy_load()
        _, info = self._plan(np.ascontiguousarray(offsets, dtype=np.int32), node or self.embeddings)
        return {f[0]: getattr(info, f[0]) for f in info._fields_}

    def check_overflow(self):
        """fp16 split precisions only: 1 if a feature or activation went beyond the fp16 range (+-65504) in a forward since
        the last check, 2 if every feature staged since then was below 2^-8 in magnitude (the low halves of the split are
        subnormal there), else 0 (xv_check_overflow; synchronous -- call after the results have been fetched)."""
        if self._precision not in _F16_RANGE or self._h is None:
            return 0
        return _lib.check(self._lib.xv_check_overflow(self._h, 1), self._h)

    def flags_async(self, host_flags):
        """Stream-ordered form of check_overflow (xv_flags_async): `host_flags` is a pinned int32 tensor of 2 elements that
        receives the flag words behind everything enqueued on the current stream; decode with raise_on_flags() once an
        event recorded after this call has completed."""
        if self._precision not in _F16_RANGE or self._h is None:
            host_flags.zero_()
            return
        stream = self._torch.cuda.current_stream(self._device_index).cuda_stream
        _lib.check(self._lib.xv_flags_async(self._h, C.c_void_p(host_flags.data_ptr()), C.c_void_p(stream)), self._h)

    def raise_on_flags(self, code, emb=None):
        if code == 1 or (emb is not None and self._precision in _F16_RANGE and not np.isfinite(emb).all()):
            raise FloatingPointError("the %s path converted a value beyond the fp16 range (+-65504): an input feature or an "
                                     "activation is too large; run with precision 'bf16x3' (full fp32 range) or 'f32'" % self._precision)
        if code == 2:
            raise FloatingPointError("every input feature of the batch is below 2^-8 in magnitude: the %s path would lose precision "
                                     "silently (subnormal low halves); rescale the features or run with precision 'bf16x3'" % self._precision)

    def decode_flags(self, host_flags):
        return int(self._lib.xv_flags_decode(C.c_void_p(host_flags.data_ptr()))) if self._lib is not None else 0

    def _checked(self, emb):
        """Host copies of fp16-split results are range-checked, so that an overflow fails loudly instead of writing wrong
        vectors into an ark (ReLU turns the NaNs an overflow produces into zeros: the output itself can look finite)."""
        self.raise_on_flags(self.check_overflow(), emb)
        return emb

    def _fallback_trainer(self, why):
        """The bf16x3 twin for batches the fp16 precisions refuse (full fp32 exponent range, same weights, same node)."""
        if self._fallback is None:
            import warnings
            warnings.warn("%s -- running such batches in bf16x3 (Trainer(range_fallback=False) raises instead)" % why)
            fb = Trainer(self.params, None, self.dim, single_cpu=self._single_cpu, device=self._device_index, precision="bf16x3",
                         range_fallback=False)
            fb.model = self.model
            fb.build("predict")
            for name, value in self._options.items():
                fb.set_option(name, value)
            fb.load_weights(self._weights_host, self._step)
            self._fallback = fb
        self._fallback.embeddings = self.embeddings
        return self._fallback

    def checked_or_rerun(self, emb, feats_dev, offsets, node=None, code=None):
        """`emb` = the fetched result of predict_packed(feats_dev, offsets, node): returned as it is when the range guard is clean
        (`code`: already decoded flags of that forward, else the handle's flag is read), else the same batch from the bf16x3 twin
        (or the FloatingPointError, without range_fallback)."""
        try:
            self.raise_on_flags(self.check_overflow() if code is None else code, emb)
            return emb
        except FloatingPointError as e:
            if not self._range_fallback:
                raise
            with self._torch.cuda.device(self._device_index):
                return self._fallback_trainer(str(e)).predict_packed(feats_dev, offsets, node).cpu().numpy()

    def _lazy_load(self):
        # model/trainer.py:891-895
        if self.model is not None and os.path.isfile(os.path.join(self.model, "checkpoint")):
            self.load()
        else:
            sys.exit("Cannot find model in %s" % self.model)

    # ------------------------------------------------------------------ Trainer.predict
    def predict(self, features):
        """Output the embeddings (model/trainer.py:886-913): `features` is [T,d] or [B,T,d]
        float; columns beyond `dim` are dropped; returns [E] / [B,E] for segment-level nodes
        and [T',E] / [B,T',E] for frame-level nodes."""
        if not self.is_loaded:
            self._lazy_load()
        features = np.asarray(features)
        rank = len(features.shape)
        assert (rank == 2 or rank == 3)
        if rank == 2:
            features = np.expand_dims(features, axis=0)
        if self.first_feature_split_alert and features.shape[-1] != self.dim:
            self.first_feature_split_alert = False
        if features.shape[-1] < self.dim:
            raise ValueError("features have %d columns, the network needs %d" % (features.shape[-1], self.dim))
        torch = self._torch
        b, t, d = features.shape
        host = np.ascontiguousarray(features, dtype=np.float32).reshape(b * t, d)
        with torch.cuda.device(self._device_index):
            dev = torch.from_numpy(host).to("cuda:%d" % self._device_index)
            offsets = np.arange(b + 1, dtype=np.int32) * t
            node = self.embeddings
            out = self.predict_packed(dev, offsets, node)
            _, info = self._plan(offsets, node)
            emb = self.checked_or_rerun(out.cpu().numpy(), dev, offsets, node)
        if node == "attention_weights":
            emb = emb.reshape(b, -1, emb.shape[-1])
        elif info.frame_level and emb.shape[0] > b * t:       # ResNet block output [b, l, f, c] (test-only nodes)
            emb = emb.reshape(b, t, -1, emb.shape[-1])
        elif info.frame_level:
            emb = emb.reshape(b, -1, emb.shape[-1])
        if rank == 2:
            emb = np.squeeze(emb, axis=0)
        return emb

    def predict_list(self, utterances, node=None):
        """Ragged batch: a list of [T_i, d] matrices packed back to back and run as ONE launch
        sequence (what the reference does with one sess.run per utterance, extract.py:89).
        Returns [n, E] for a segment-level node, a list of [T_i', E] for a frame-level node.
        = collect(submit_list(...)); callers that can keep two batches in flight use the pair directly."""
        return self.collect(self.submit_list(utterances, node))

    NUM_SLOTS = 3

    def submit_list(self, utterances, node=None):
        """Enqueue one ragged batch and return a ticket for collect(); nothing here waits for the device.  The batch is
        packed into a pinned staging slot, copied to the device on a COPY stream (so that the copy of batch i + 1 runs
        beside the kernels of batch i), run on the current stream, and its result and range flags are copied into the
        slot's pinned output, all in stream order.  At most NUM_SLOTS tickets may be outstanding (two keep the device busy)."""
        if not self.is_loaded:
            self._lazy_load()
        torch = self._torch
        node = node or self.embeddings
        lens = [int(u.shape[0]) for u in utterances]
        d = int(utterances[0].shape[1])
        if d < self.dim:
            raise ValueError("features have %d columns, the network needs %d" % (d, self.dim))
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        total = int(offsets[-1])
        devname = "cuda:%d" % self._device_index
        with torch.cuda.device(self._device_index):
            comp = torch.cuda.current_stream(self._device_index)
            if self._slots is None:
                self._slots = [dict(pin=None, dev=None, out=None, flags=torch.zeros(2, dtype=torch.int32).pin_memory(),
                                    done=None, busy=False) for _ in range(self.NUM_SLOTS)]
                self._copy_stream = torch.cuda.Stream(device=self._device_index)
            sl = self._slots[self._slot_turn % self.NUM_SLOTS]
            if sl["busy"]:
                raise RuntimeError("submit_list: %d batches are already in flight; collect() one first" % self.NUM_SLOTS)
            self._slot_turn += 1
            if sl["pin"] is None or sl["pin"].shape[0] < total or sl["pin"].shape[1] != d:
                rows = max(int(total * 1.25), 4096)
                sl["pin"] = torch.empty((rows, d), dtype=torch.float32, pin_memory=True)
                sl["dev"] = torch.empty((rows, d), dtype=torch.float32, device=devname)
                self._copy_stream.wait_stream(comp)          # the block may still be in use by work queued on the compute stream
            self._pack(utterances, lens, d, sl["pin"])
            with torch.cuda.stream(self._copy_stream):
                sl["dev"][:total].copy_(sl["pin"][:total], non_blocking=True)
                h2d = torch.cuda.Event()
                h2d.record(self._copy_stream)
            comp.wait_event(h2d)
            plan, info = self._plan(offsets, node)
            n_out = int(info.out_rows) * int(info.out_cols)
            if sl["out"] is None or sl["out"].numel() < n_out:
                sl["out"] = torch.empty(max(n_out, 1 << 18), dtype=torch.float32, pin_memory=True)
            host_out = sl["out"][:n_out].view(int(info.out_rows), int(info.out_cols))
            if not info.frame_level and node != "attention_weights":
                self.predict_packed(sl["dev"][:total], offsets, node, out=host_out)      # [B, E]: written into the pinned slot directly
            else:
                out = self.predict_packed(sl["dev"][:total], offsets, node)
                host_out.copy_(out, non_blocking=True)
            self.flags_async(sl["flags"])
            sl["done"] = torch.cuda.Event()
            sl["done"].record(comp)
            sl["busy"] = True
        return (sl, host_out, lens, int(offsets[-1]), node, bool(info.frame_level))

    def _pack(self, utterances, lens, d, pin):
        """The ragged batch, a list of separate [T_i, d] matrices, back to back into the pinned staging slot: natively with four
        threads when every matrix is C-contiguous float32 (xv_pack_rows; 9 MB per 256 x 300 frames are ~1 ms on one core),
        row by row otherwise."""
        n = len(utterances)
        if all(isinstance(u, np.ndarray) and u.dtype == np.float32 and u.flags.c_contiguous and u.shape[1] == d for u in utterances):
            ptrs = np.fromiter((u.__array_interface__["data"][0] for u in utterances), dtype=np.uint64, count=n)
            nbytes = np.asarray(lens, dtype=np.int64) * (4 * d)
            got = self._lib.xv_pack_rows(C.c_void_p(ptrs.ctypes.data), C.c_void_p(nbytes.ctypes.data), n,
                                         C.c_void_p(pin.data_ptr()), 4)
            if got != int(nbytes.sum()):
                raise RuntimeError("xv_pack_rows copied %d of %d bytes" % (got, int(nbytes.sum())))
            return
        stage = pin.numpy()
        pos = 0
        for u, t in zip(utterances, lens):
            stage[pos:pos + t] = u
            pos += t

    def collect(self, ticket):
        """Wait for a batch submitted with submit_list and return what predict_list returns."""
        sl, host_out, lens, total, node, frame_level = ticket
        sl["done"].synchronize()
        sl["busy"] = False
        emb = host_out.numpy().copy()                   # the slot is reused by the next submit
        offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
        # (the slot's device copy of the batch is intact until the next submit)
        emb = self.checked_or_rerun(emb, sl["dev"][:total], offsets, node, code=self.decode_flags(sl["flags"]))
        if node == "attention_weights" or not frame_level:
            return emb
        ctx = (total - emb.shape[0]) // len(lens)
        res, pos = [], 0
        for t in lens:
            res.append(emb[pos:pos + t - ctx])
            pos += t - ctx
        return res

    # ------------------------------------------------------------------ teardown
    def _release(self):
        if self._lib is not None:
            for plan, _ in self._plans.values():
                self._lib.xv_plan_destroy(plan)
            self._plans = collections.OrderedDict()
            self._pinned_plans = set()
            self._ws_pinned = False
            if self._h is not None:
                self._lib.xv_destroy(self._h)
                self._h = None
        if getattr(self, "_fallback", None) is not None:
            self._fallback.close()
            self._fallback = None
        self._ws = None
        self._pinned = None
        self._slots = None
        self._copy_stream = None
        self.is_loaded = False

    def close(self):
        """model/trainer.py:270-275."""
        self._release()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass
