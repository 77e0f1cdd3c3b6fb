"""GPU: the alternative code paths that the default run never takes.

* randomised ragged batches (seeded) through the default kernels: lengths from the network's minimum context to
  several M tiles, batch sizes that leave partial tiles, both poolings, embedding and frame-level nodes;
* the K-split tail form and the fused / unfused statistics pooling, switched with xv_set_option in-process (the two
  earlier GEMM kernels are lab-only code now, -DXV_LAB, and not part of the shipped library);
* the hardware property the default kernel's hand-counted s_waitcnt rely on (LDS-DMA and register loads retire in
  issue order under one vmcnt counter): tools/vmcnt_order_test.hip compiled and run here.
Tolerance as in test_gpu_parity.py (BASELINE.json north_star: relative L2 <= 1e-4)."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_numpy

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(b.shape)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _trainer(params, weights, dim, precision, **kw):
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd.trainer import Trainer
    tr = Trainer(Params(**dict(params)), None, dim, single_cpu=True, device=0, precision=precision, **kw)
    tr.build("predict")
    tr.load_weights(weights)
    return tr


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_ragged_batches(seed):
    from tf_kaldi_speaker_amd import synth
    rng = np.random.default_rng(seed)
    pooling = ["statistics_pooling", "self_attention"][seed % 2]
    params = dict(synth.TDNN_ATT_PARAMS if pooling == "self_attention" else synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=seed)
    nutt = int(rng.integers(1, 12))
    lens = [int(x) for x in rng.choice([15, 16, 17, 31, 64, 127, 128, 129, 200, 333], size=nutt)]
    utts = synth.synth_features(nutt, lens, 30, seed=seed + 50)
    for precision in ("bf16x3", "f16x3", "f16f6", "f32"):
        tr = _trainer(params, weights, 30, precision)
        got = tr.predict_list(utts)
        for i, u in enumerate(utts):
            ref = ref_numpy.predict(u, weights, params, 30)
            assert _rel(got[i], ref) <= TOL, (precision, pooling, lens, i)
        tr.set_embedding("tdnn5_relu")                       # frame-level node, rows compacted per utterance
        got = tr.predict_list(utts[:3])
        for i, u in enumerate(utts[:3]):
            ref = ref_numpy.predict(u, weights, params, 30, node="tdnn5_relu")
            assert np.asarray(got[i]).shape[0] == ref.shape[-2]
            assert _rel(got[i], ref) <= TOL, (precision, "tdnn5_relu", lens[i])
        tr.close()


def _baseline_batch():
    import torch
    from tf_kaldi_speaker_amd import synth
    B, T = 256, 300                                    # L3: 2336 tiles on 768 slots -> 32 tail tiles, L2: 64
    feats = torch.from_numpy(np.concatenate(synth.synth_features(B, T, 30, seed=5))).cuda()
    return feats, np.arange(B + 1, dtype=np.int32) * T


def _run_nodes(tr, feats, offs, nodes):
    res = {}
    for node in nodes:
        a = tr.predict_packed(feats, offs, node).cpu().numpy()
        b = tr.predict_packed(feats, offs, node).cpu().numpy()
        assert np.array_equal(a, b), "non-deterministic " + node
        res[node] = a.astype(np.float64)
    return res


def _rel2(a, b):
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


@pytest.mark.parametrize("net,precision", [("tdnn", "bf16x3"), ("tdnn", "f16f6"), ("etdnn", "f16f6")])
def test_tail_ksplit_matches_plain_and_exact(net, precision):
    """The K-split of the last, nearly empty round of tiles (gemm_bf16x3_tail_plan) against the same kernel without
    it (xv_set_option "tail_split" 0) and against the exact fp32 path, at the BASELINE geometry where it is active.  f16f6: the
    slices of the two-unit kernel (whole quads of channel blocks) and its reduce, which writes the split-blocked rows or -- for the
    embedding run, where tdnn2 feeds tdnn3 -- the two-unit block format (codes rounded half-up there, to nearest-even in the tile
    epilogue: a difference in the cross terms' operands only, hence the wider bound); extended TDNN: the tails of the dense layers
    that write the block format themselves (raw slices of the one-tap kernel + the same reduce)."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    if net == "etdnn":
        params.update(network_type="extended_tdnn", embedding_node="tdnn12_dense")
    weights = synth.synth_weights(params, 30, seed=0)
    feats, offs = _baseline_batch()
    # embedding; frame-level fp32 outputs of two tail layers
    nodes = ("tdnn12_dense", "tdnn3_conv", "tdnn4_relu") if net == "etdnn" else ("tdnn6_dense", "tdnn3_conv", "tdnn2_relu")
    tr = _trainer(params, weights, 30, precision)
    tail = _run_nodes(tr, feats, offs, nodes)
    tr.set_option("tail_split", 0)
    plain = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    tr = _trainer(params, weights, 30, "f32")
    exact = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    for node in nodes:
        t, pl, ex = tail[node], plain[node], exact[node]
        assert t.shape == pl.shape == ex.shape, node
        assert _rel2(t, pl) <= (2e-6 if precision == "bf16x3" else 2e-5), (node, _rel2(t, pl))   # same products, other summation order in the tail tiles
        assert _rel2(t, ex) <= TOL, node
    assert not np.array_equal(tail["tdnn3_conv"], plain["tdnn3_conv"])   # the path really was taken


def test_fused_pooling_path_is_taken_at_the_baseline_geometry():
    """Statistics pooling fused into the tdnn5 epilogue (the default whenever tdnn5's activations have no other
    reader) against the unfused path (xv_set_option "pool_fusion" 0: tdnn5 stored, stat_pool_kernel) and against the
    exact fp32 path, at 256 x 300 frames: both agree with the exact result, and their bits differ (different
    summation order), which shows that the fused epilogue is what produced the default output."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=0)
    feats, offs = _baseline_batch()
    nodes = ("pooling", "tdnn6_dense")
    tr = _trainer(params, weights, 30, "bf16x3")
    ws_fused = tr.plan_info(offs, "tdnn6_dense")["workspace_bytes"]
    fused = _run_nodes(tr, feats, offs, nodes)
    tr.set_option("pool_fusion", 0)
    ws_plain = tr.plan_info(offs, "tdnn6_dense")["workspace_bytes"]
    plain = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    tr = _trainer(params, weights, 30, "f32")
    exact = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    for node in nodes:
        assert _rel2(fused[node], exact[node]) <= TOL, node
        assert _rel2(plain[node], exact[node]) <= TOL, node
        assert _rel2(fused[node], plain[node]) <= 5e-6, node
    assert not np.array_equal(fused["pooling"], plain["pooling"])
    assert ws_plain > ws_fused                                  # the unfused plan carries the [73216, 1500] activation


@pytest.mark.parametrize("kw", [
    {},                                                                                   # the shipped config: 1 head
    {"att_num_heads": 3, "att_split_key": False, "att_split_value": False},               # every head pools every channel
    {"att_num_heads": 5, "num_nodes_pooling_layer": 1600, "att_key_num_nodes": [1500, 1600]},   # split heads of 320 channels
])
@pytest.mark.parametrize("precision", ["bf16x3", "f16f6"])
def test_fused_attention_path_matches_unfused_and_exact(kw, precision):
    """Attentive pooling with the key / value never stored (score partials in the epilogue of the last key layer,
    weighted moments in the epilogue of the value layer, model/pooling.py:189-217) against the unfused kernels
    (xv_set_option "att_fusion" 0) and against the exact fp32 path, on a ragged batch with several M tiles."""
    from tf_kaldi_speaker_amd import synth
    import torch
    params = dict(synth.TDNN_ATT_PARAMS, **kw)
    weights = synth.synth_weights(params, 30, seed=6)
    weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 30.0
    lens = [300, 64, 200, 15, 129, 333, 78, 500]
    utts = synth.synth_features(len(lens), lens, 30, seed=21)
    feats = torch.from_numpy(np.concatenate(utts)).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nodes = ("att_output_before_nonlinear", "tdnn6_dense")
    tr = _trainer(params, weights, 30, precision)
    ws_fused = tr.plan_info(offs, "tdnn6_dense")["workspace_bytes"]
    fused = _run_nodes(tr, feats, offs, nodes)
    tr.set_option("att_fusion", 0)
    ws_plain = tr.plan_info(offs, "tdnn6_dense")["workspace_bytes"]
    plain = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    tr = _trainer(params, weights, 30, "f32")
    exact = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    for node in nodes:
        assert _rel2(fused[node], exact[node]) <= TOL, (node, kw)
        assert _rel2(plain[node], exact[node]) <= TOL, (node, kw)
        assert _rel2(fused[node], plain[node]) <= 2e-5, (node, kw)
    assert not np.array_equal(fused[nodes[0]], plain[nodes[0]])      # a different summation order: the fused path ran
    assert ws_plain > ws_fused                                        # no [rows, 1500] key / value in the fused plan
    for i in (0, 3, 7):                                               # and against the float64 oracle
        ref = ref_numpy.predict(utts[i], weights, params, 30)
        assert _rel(fused["tdnn6_dense"][i], ref) <= TOL, (i, kw)


@pytest.mark.parametrize("net", ["tdnn_att", "resnet"])
def test_three_slab_kernel_is_bit_identical_to_the_two_slab_kernel(net):
    """One-tap layers (dense, attention epilogues, the ResNet grid convolutions) run on gemm_bf16x3_w1p3_kernel (three slab
    buffers, slabs issued two steps ahead).  It performs the same MFMAs in the same order as the two-buffer kernel, so
    every output must be bit-identical with xv_set_option "slab3" 0 -- a slab read before it has landed or overwritten
    before it was read would show up here."""
    from tf_kaldi_speaker_amd import synth
    import torch
    if net == "resnet":
        params = dict(synth.RESNET_PARAMS)
        weights = synth.synth_resnet_weights(params, seed=1)
        dim, lens, nodes = 40, [120, 37, 64], ("tdnn6_dense", "conv3a", "conv5_relu")
    else:
        params = dict(synth.TDNN_ATT_PARAMS)
        weights = synth.synth_weights(params, 30, seed=6)
        dim, lens, nodes = 30, [300, 64, 200, 15, 129, 333, 78, 500] * 4, ("tdnn6_dense", "tdnn4_relu", "att_key1_relu")
    utts = synth.synth_features(len(lens), lens, dim, seed=31)
    feats = torch.from_numpy(np.concatenate(utts)).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for prec in ("bf16x3", "f16x3"):
        tr = _trainer(params, weights, dim, prec)
        tr.set_option("grid_compact", 0)        # the compact grid form exists on the three-slab kernel only
        a = _run_nodes(tr, feats, offs, nodes)
        tr.set_option("slab3", 0)
        b = _run_nodes(tr, feats, offs, nodes)
        tr.close()
        for node in nodes:
            assert np.array_equal(a[node], b[node]), (net, prec, node)


@pytest.mark.parametrize("kw", [dict(), dict(resnet_time_stride=True), dict(resnet_maxpooling=True, resnet_time_stride=True)])
def test_compact_grid_rows_are_bit_identical_to_the_full_enumeration(kw):
    """ResNet grid convolutions enumerate only their output bins by default (GemmArgs::arow, csrc/grid.hip: no MFMA work on
    border positions, border zeroed by grid_zero_border_kernel).  Every bin accumulates the same products in the same order
    as in the full enumeration of the input positions (xv_set_option "grid_compact" 0), so every block output must be
    bit-identical -- on a ragged batch with even and odd lengths, with and without the time stride.  Run twice on the same
    workspace: a border left dirty by the first forward would change the second."""
    from tf_kaldi_speaker_amd import synth
    import torch
    params = dict(synth.RESNET_PARAMS, **kw)
    weights = synth.synth_resnet_weights(params, seed=2)
    lens = [120, 37, 64, 41, 16, 203]
    nodes = ("tdnn6_dense", "conv1a", "conv2a", "conv2b_0", "conv3a", "conv4a", "conv4b_0", "conv5_relu")
    utts = synth.synth_features(len(lens), lens, 40, seed=33)
    feats = torch.from_numpy(np.concatenate(utts)).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    for prec in ("bf16x3", "f16x3"):
        tr = _trainer(params, weights, 40, prec)
        a = _run_nodes(tr, feats, offs, nodes)
        a2 = _run_nodes(tr, feats, offs, nodes)
        tr.set_option("grid_compact", 0)
        b = _run_nodes(tr, feats, offs, nodes)
        tr.close()
        for node in nodes:
            assert np.array_equal(a[node], b[node]), (kw, prec, node)
            assert np.array_equal(a[node], a2[node]), (kw, prec, node, "second forward")


@pytest.mark.parametrize("net", ["tdnn", "tdnn_narrow", "tdnn_128", "etdnn"])
def test_f16f6_two_unit_split(net):
    """XV_PREC_F16F6: the 5-, 7- and 9-tap convolutions compute hi*hi in fp16 and the two cross terms on the block-scaled fp6 path
    (csrc/gemm_f6v2.hip: 1.5 MFMA units per product instead of 3).  tests/analysis/f16f8_error_model.py predicts ~1e-5 on the x-vector;
    the bar is the path's 1e-4 against the float64 oracle, on every stage endpoint of the converted layers (their conv / bn
    stages run through the same kernel with other epilogue vectors) and on the embedding; against the exact fp32 path the
    frame-level layers must stay within 5e-5.  Ragged batch with the shortest possible utterance; 64 channels: no quad of channel
    blocks, the layers fall back to the f16x3 kernels; 128 channels: exactly one quad (the peeled last body alone); extended TDNN:
    5-, 5-, 7- and 9-tap layers behind one-tap dense layers that write the block format themselves.  Deterministic."""
    from oracle import ref_numpy
    from tf_kaldi_speaker_amd import synth
    import torch
    ch = {"tdnn_narrow": 64, "tdnn_128": 128}.get(net, 512)     # 64: two channel blocks, no quad: every layer stays on the f16x3 kernels (the
                                                                  # fallback); 128: exactly one quad = the peeled last body alone
    params = dict(synth.TDNN_STAT_PARAMS)
    if net == "etdnn":                                  # conv1d k = 5, 5, 7, 9 at tdnn1 / 3 / 5 / 7 (model/tdnn.py:343-591)
        params.update(network_type="extended_tdnn", embedding_node="tdnn12_dense")
    weights = synth.synth_weights(params, 30, seed=4, channels=ch)
    lens = [300, 64, 200, 23 if net == "etdnn" else 15, 129, 333]
    utts = synth.synth_features(len(lens), lens, 30, seed=41)
    feats = torch.from_numpy(np.concatenate(utts)).cuda()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    emb = params["embedding_node"]
    nodes = ("tdnn3_conv", "tdnn3_bn", "tdnn3_relu", "tdnn5_relu", "tdnn7_conv", "tdnn7_relu", emb) if net == "etdnn" else \
            ("tdnn2_conv", "tdnn2_bn", "tdnn2_relu", "tdnn3_relu", emb)
    tr = _trainer(params, weights, 30, "f16f6")
    got = _run_nodes(tr, feats, offs, nodes)            # _run_nodes also checks bit-identical repetition
    # the zero-weight taps of a scaled MFMA read up to eleven rows behind the last input row: with the workspace full of 0xFF
    # (an E8M0 scale byte of 255 is a NaN, and NaN x 0 = NaN) the result must not change
    tr._ws.fill_(255)
    again = tr.predict_packed(feats, offs, emb).cpu().numpy().astype(np.float64)
    assert np.array_equal(again, got[emb]), net
    tr.close()
    tr = _trainer(params, weights, 30, "f32")
    exact = _run_nodes(tr, feats, offs, nodes)
    tr.close()
    for node in nodes[:-1]:
        assert _rel2(got[node], exact[node]) <= 5e-5, (net, node, _rel2(got[node], exact[node]))
        if net != "tdnn_narrow":
            assert _rel2(got[node], exact[node]) >= 1e-6, (net, node, "the two-unit kernel did not run")
    for i in (0, 3, 5):
        ref = ref_numpy.predict(utts[i], weights, params, 30)
        assert _rel(got[emb][i], ref) <= TOL, (net, i)
        assert _rel(got[emb][i], ref) <= 2e-5, (net, i)                # the error model's figure, with slack


def test_f16x3_is_tighter_than_bf16x3_and_reports_overflow():
    """The fp16 hi/lo split format (XV_PREC_F16X3): same kernels and layout as bf16x3, 22 instead of 16 significand bits.
    At the BASELINE geometry its embeddings and its (peaky) attention weights must be several times closer to the exact
    fp32 path than bf16x3's; and a feature beyond the fp16 range must surface as an error, not as a NaN embedding."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_ATT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=2)
    weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 1000.0          # peaky attention (test_gpu_parity)
    utts = synth.synth_features(8, 300, 30, seed=12)
    feats = np.stack(utts)
    res = {}
    for prec in ("f32", "bf16x3", "f16x3"):
        tr = _trainer(params, weights, 30, prec, range_fallback=False)
        res[prec] = {}
        for node in ("attention_weights", "tdnn6_dense"):
            tr.set_embedding(node)
            res[prec][node] = tr.predict(feats).astype(np.float64)
        if prec == "f16x3":
            bad = feats.copy()
            bad[3, 100, 7] = 1.0e5
            with pytest.raises(FloatingPointError):
                tr.predict(bad)
        tr.close()
    for node in ("attention_weights", "tdnn6_dense"):
        e16 = _rel2(res["f16x3"][node], res["f32"][node])
        ebf = _rel2(res["bf16x3"][node], res["f32"][node])
        assert e16 <= 2e-5 and e16 * 4 <= ebf, (node, e16, ebf)


def test_vmcnt_retires_in_issue_order(tmp_path, repo_root):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available on this box")
    exe = str(tmp_path / "vmcnt_order_test")
    subprocess.run([hipcc, "-O2", "--offload-arch=gfx950", os.path.join(repo_root, "tools", "vmcnt_order_test.hip"), "-o", exe],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("mode")]
    assert len(lines) == 2, r.stdout
    for l in lines:
        assert ": 0 out-of-order" in l, l


@pytest.mark.parametrize("precision", ["f16x3", "f16f6", "bf16x3"])
@pytest.mark.parametrize("layer,shift", [(2, 10), (4, 10), (3, 14), (1, -9)])
def test_split_formats_survive_a_rescaled_layer(precision, layer, shift):
    """The fp16 split (hi + lo, 11 + 11 significand bits) has a lower end too: below 2^-3 the low half is subnormal.  A model
    whose layer `layer` has its batch-normalisation gamma / beta scaled by 2^-shift and the next layer's kernel by 2^shift
    computes the same function (ReLU is positively homogeneous; the float64 oracle agrees to rounding) but its activations
    sit at ~1e-3 (shift 10) or ~6e-5 (shift 14: below the smallest normal fp16 number) -- or at ~500 (shift -9).  The
    library keeps every layer's split copy at a power-of-two scale derived from gamma / beta (csrc/xvec_api.hip,
    act_exponent), so all three split precisions must stay at their usual accuracy: 1e-4 is the path's bar, 2e-5 what
    they deliver on the unscaled model."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = dict(synth.synth_weights(params, 30, seed=3))
    f = 2.0 ** -shift
    for nm in ("gamma", "beta"):
        weights["tdnn/tdnn%d_bn/%s" % (layer, nm)] = (weights["tdnn/tdnn%d_bn/%s" % (layer, nm)] * f).astype(np.float32)
    nxt = "tdnn/tdnn%d_%s/kernel" % (layer + 1, "conv" if layer + 1 <= 3 else "dense")
    weights[nxt] = (weights[nxt] / f).astype(np.float32)
    lens = [300, 64, 15, 129]
    utts = synth.synth_features(len(lens), lens, 30, seed=43)
    base = dict(synth.synth_weights(params, 30, seed=3))
    tr = _trainer(params, weights, 30, precision)
    got = tr.predict_list(utts)
    mid = tr.predict_list(utts[:2], node="tdnn%d_relu" % layer)          # the scaled activations themselves (fp32 endpoint)
    tr.close()
    for i, u in enumerate(utts):
        ref = ref_numpy.predict(u, weights, params, 30)
        ref0 = ref_numpy.predict(u, base, params, 30)
        assert _rel(ref, ref0.astype(np.float64)) <= 1e-6                  # the rescaled model is the same function
        err = _rel(got[i], ref)
        assert err <= TOL, (precision, layer, shift, i, err)
        assert err <= 2e-5, (precision, layer, shift, i, err)
    for i in range(2):
        ref = ref_numpy.predict(utts[i], weights, params, 30, node="tdnn%d_relu" % layer)
        assert _rel(mid[i], ref) <= TOL, (precision, layer, shift, i)


def test_fp16_split_refuses_features_it_cannot_represent():
    """Input features are whatever the caller sends (no normalisation in front of them): features so small that every low
    half of the fp16 split is subnormal (all below 2^-8) are refused with an error in the fp16 precisions -- as features
    beyond 65504 are -- and are no problem for bf16x3, which has the full fp32 exponent range."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=0)
    feats = np.stack(synth.synth_features(2, 40, 30, seed=9))
    tiny = (feats * 2.0 ** -14).astype(np.float32)
    for prec in ("f16x3", "f16f6"):
        tr = _trainer(params, weights, 30, prec, range_fallback=False)
        with pytest.raises(FloatingPointError, match="below 2\\^-8"):
            tr.predict(tiny)
        ok = tr.predict(feats)                                              # the flag was reset: the next batch is fine
        assert _rel(ok[0], ref_numpy.predict(feats[0], weights, params, 30)) <= TOL
        tr.close()
        # the default: the refused batch runs on a bf16x3 twin of the model instead (the reference accepts any finite features);
        # batches inside the range stay on the fast path, bit-identical to a trainer without the twin
        tr = _trainer(params, weights, 30, prec)
        with pytest.warns(UserWarning, match="bf16x3"):
            got = tr.predict(tiny)
        assert _rel(got[1], ref_numpy.predict(tiny[1], weights, params, 30)) <= TOL
        assert np.array_equal(tr.predict(feats), ok)
        huge = feats.copy()
        huge[1, 7, 3] = 1.0e5
        got = tr.predict_list([huge[0], huge[1]])                          # the pipelined interface: collect() re-runs the batch
        assert _rel(got[1], ref_numpy.predict(huge[1], weights, params, 30)) <= TOL
        assert np.array_equal(tr.predict_list([feats[0], feats[1]]), ok)
        tr.close()
    tr = _trainer(params, weights, 30, "bf16x3")
    got = tr.predict(tiny)
    assert _rel(got[1], ref_numpy.predict(tiny[1], weights, params, 30)) <= TOL
    tr.close()


def test_handles_release_their_device_memory():
    """xv_destroy frees everything xv_finalize uploaded (incl. the two-unit layers' second weight copy and the ResNet's direct
    conv0 kernel): creating and closing handles in a loop must not eat device memory."""
    import torch
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=0)
    rparams = dict(synth.RESNET_PARAMS)
    rweights = synth.synth_resnet_weights(rparams, seed=0)
    feats = np.stack(synth.synth_features(1, 40, 30, seed=1))
    def cycle():
        tr = _trainer(params, weights, 30, "f16f6")
        tr.predict(feats)
        tr.close()
        tr = _trainer(rparams, rweights, 40, "f16x3")
        tr.close()
    cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        cycle()
    torch.cuda.synchronize()
    torch.cuda.empty_cache()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 8 << 20, (free0, free1)        # one cycle uploads > 60 MB of weights


def test_pipelined_host_interface_equals_blocking_calls():
    """Trainer.submit_list / collect (two batches in flight: H2D of batch i + 1 on a copy stream beside the kernels of batch i,
    range flags read out in stream order) must return exactly what one blocking predict_list per batch returns, in any
    interleaving the slots allow, for ragged batches of different sizes; a fourth outstanding ticket is refused; an overflow in
    one batch is reported for THAT batch only."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=0)
    rs = np.random.RandomState(4)
    batches = [synth.synth_features(n, [int(t) for t in rs.randint(15, 400, size=n)], 30, seed=60 + n) for n in (5, 64, 1, 17, 33, 9)]
    tr = _trainer(params, weights, 30, "f16f6", range_fallback=False)
    want = [tr.predict_list(b) for b in batches]
    got, tickets = [], []
    for b in batches:
        tickets.append(tr.submit_list(b))
        if len(tickets) == 2:
            got.append(tr.collect(tickets.pop(0)))
    t3 = tr.submit_list(batches[0])
    t4 = tr.submit_list(batches[1])
    with pytest.raises(RuntimeError):
        tr.submit_list(batches[2])                        # NUM_SLOTS = 3 outstanding
    got.append(tr.collect(tickets.pop(0)))
    assert np.array_equal(tr.collect(t3), want[0])
    assert np.array_equal(tr.collect(t4), want[1])
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
    bad = [u.copy() for u in batches[3]]
    bad[2][5, 7] = 1.0e5                                   # beyond the fp16 range
    t_ok, t_bad, = tr.submit_list(batches[2]), tr.submit_list(bad)
    assert np.array_equal(tr.collect(t_ok), want[2])
    with pytest.raises(FloatingPointError):
        tr.collect(t_bad)
    assert np.array_equal(tr.predict_list(batches[4]), want[4])         # the flag was cleared in stream order
    frames = tr.collect(tr.submit_list(batches[0], node="tdnn3_relu"))     # frame-level node: list of [T_i - 14, 512]
    assert [f.shape[0] for f in frames] == [u.shape[0] - 14 for u in batches[0]]
    tr.close()
