"""GPU parity tests proper: the HIP path (through the C ABI, via the Trainer mirror) against
oracle/ref_numpy.py (float64) on the same seeded inputs.

Tolerance: BASELINE.json north_star -> relative L2 <= 1e-4 per utterance embedding.  The
fp32-MFMA path is expected at ~1e-6; per-endpoint checks use the same bar.
"""
import numpy as np
import pytest

from oracle import ref_numpy

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(b.shape)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def _trainer(params_dict, weights, dim, precision="f32"):
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd.trainer import Trainer
    p = Params(**dict(params_dict))
    tr = Trainer(p, None, dim, single_cpu=True, device=0, precision=precision)
    tr.build("predict")
    tr.load_weights(weights)
    return tr, p


@pytest.fixture(scope="module")
def stat_model():
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_STAT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=0)
    return params, weights


PRECISIONS = ["f32", "bf16x3", "f16x3", "f16f6"]     # f16f6 = what bench.py measures; bf16x3 = the library default
_report = []


def _note(test, precision, name, err):
    _report.append("%-28s %-7s %-28s %.3e" % (test, precision, name, err))


def teardown_module(module):
    import os
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/parity_report.txt", "w") as f:
        f.write("\n".join(_report) + "\n")


@pytest.mark.parametrize("precision", PRECISIONS)
def test_every_endpoint_statistics_pooling(stat_model, precision):
    """All endpoints of model/tdnn.py:36-181 on a [3,41,30] batch."""
    from tf_kaldi_speaker_amd import synth
    params, weights = stat_model
    feats = np.stack(synth.synth_features(3, 41, 30, seed=11))
    _, ep = ref_numpy.entire_network(feats, weights, params)
    tr, _ = _trainer(params, weights, 30, precision)
    for name, ref in ep.items():
        tr.set_embedding(name)
        got = tr.predict(feats)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = _rel(got, ref)
        _note("endpoints_stat", precision, name, err)
        assert err <= TOL, (name, err)
    tr.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_every_endpoint_self_attention(precision):
    """model/pooling.py:55-240 with the shipped attention config
    (egs/voxceleb/v1/nnet_conf/tdnn_softmax_1e-2_tdnn4_att_pretrain.json:15-28)."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_ATT_PARAMS)
    weights = synth.synth_weights(params, 30, seed=2)
    weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 1000.0   # peaky attention
    feats = np.stack(synth.synth_features(2, 52, 30, seed=12))
    _, ep = ref_numpy.entire_network(feats, weights, params)
    assert ep["attention_weights"].max() > 5.0 / 38                # really non-uniform (uniform = 1/38)
    tr, _ = _trainer(params, weights, 30, precision)
    for name, ref in ep.items():
        if name.startswith("tdnn") and name[4] in "123" or name in ("tdnn4_dense", "tdnn4_bn", "tdnn5_dense", "tdnn5_bn"):
            continue                                               # covered by the statistics test
        tr.set_embedding(name)
        got = tr.predict(feats)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = _rel(got, ref)
        _note("endpoints_att", precision, name, err)
        assert err <= TOL, (name, err)
    tr.close()


@pytest.mark.parametrize("kw", [
    {"att_num_heads": 4, "att_apply_nonlinear": True, "network_relu_type": "prelu"},
    {"att_num_heads": 3, "att_split_key": False, "att_split_value": False, "att_key_network_type": 3,
     "att_key_num_nodes": [96], "att_value_num_nodes": [64, 48], "att_value_network_type": 2, "att_use_scale": False,
     "att_key_input": "tdnn5_relu", "att_value_input": "tdnn4_relu", "network_relu_type": "lrelu",
     "last_layer_no_bn": True, "feature_norm": True, "feature_scaling_factor": 12.0},
    {"att_num_heads": 2, "att_key_num_nodes": [64, 32, 16], "att_key_network_type": 0, "last_layer_linear": True},
])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_attention_variants_small(kw, precision):
    """Multi-head / split / non-split / tanh / post-BN variants on a shrunk graph (channels 64)."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_ATT_PARAMS, num_nodes_pooling_layer=96, num_nodes_last_layer=40)
    params["att_key_num_nodes"] = [64, 48]
    params.update(kw)
    weights = synth.synth_weights(params, 23, seed=4, channels=64)
    weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 20.0
    feats = np.stack(synth.synth_features(3, 37, 23, seed=13))
    _, ep = ref_numpy.entire_network(feats, weights, params)
    tr, _ = _trainer(params, weights, 23, precision)
    for name in ("attention_weights", "att_output_before_nonlinear", "pooling", "tdnn6_dense", "output"):
        tr.set_embedding(name)
        got = tr.predict(feats)
        assert got.shape == ep[name].shape, (name, got.shape)
        err = _rel(got, ep[name])
        _note("att_variants", precision, name, err)
        assert err <= TOL, (name, kw, err)
    tr.close()


@pytest.mark.parametrize("pooling", ["statistics_pooling", "self_attention"])
@pytest.mark.parametrize("precision", PRECISIONS)
def test_extended_tdnn_every_endpoint(precision, pooling):
    """network_type "extended_tdnn" (model/tdnn.py:343-591): conv1d k=5,5,7,9 interleaved with dense,
    context 22, segment layers tdnn12/tdnn13.  Full-width graph on a [2,45,30] batch."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.TDNN_ATT_PARAMS if pooling == "self_attention" else synth.TDNN_STAT_PARAMS,
                  network_type="extended_tdnn", embedding_node="tdnn12_dense")
    if pooling == "self_attention":
        params.update(att_key_input="tdnn9_relu", att_value_input="tdnn10_relu", att_key_num_nodes=[256, 128], att_num_heads=2)
    weights = synth.synth_weights(params, 30, seed=6)
    if "etdnn/attention/query" in weights:
        weights["etdnn/attention/query"] = weights["etdnn/attention/query"] * 100.0
    feats = np.stack(synth.synth_features(2, 45, 30, seed=14))
    _, ep = ref_numpy.entire_network(feats, weights, params)
    assert ep["tdnn7_relu"].shape == (2, 23, 512) and ep["tdnn12_dense"].shape == (2, 512)
    tr, _ = _trainer(params, weights, 30, precision)
    for name, ref in ep.items():
        tr.set_embedding(name)
        got = tr.predict(feats)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = _rel(got, ref)
        _note("etdnn_" + pooling[:4], precision, name, err)
        assert err <= TOL, (name, err)
    with pytest.raises(ValueError):
        tr.predict(np.zeros((22, 30), np.float32))                  # context is 22 frames
    tr.close()


@pytest.mark.parametrize("maxpool", [False, True])
@pytest.mark.parametrize("width,precision", [(8, "f32"), (32, "f32"), (32, "bf16x3"), (32, "f16x3"), (32, "f16f6")])
def test_resnet18_every_block(width, precision, maxpool):
    """network_type "resnet_18" (model/resnet.py:152-351) block by block on a ragged batch; width 8 runs the
    fp32 kernels with 24-wide taps, width 32 the split kernel on whole SB blocks.  With resnet_maxpooling (:230-231)
    the 3x3 'same' max-pool behind conv0 is in the graph (endpoint conv0_max); the parametric ReLU makes activations
    negative, so a max-pool that let the zero border take part would be caught."""
    import torch
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.RESNET_PARAMS, num_nodes_pooling_layer=96, network_relu_type="prelu", resnet_blocks=[2, 1, 2, 2],
                  resnet_maxpooling=maxpool)
    weights = synth.synth_resnet_weights(params, seed=5, width=width)
    lens = [9, 14, 3]
    utts = synth.synth_features(len(lens), lens, 40, seed=15)
    tr, _ = _trainer(params, weights, 40, precision)
    packed = torch.from_numpy(np.concatenate(utts, axis=0)).cuda()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    eps = [ref_numpy.entire_network(u[None], weights, params)[1] for u in utts]
    assert ("conv0_max" in eps[0]) == maxpool
    for name in eps[0]:
        got = tr.predict_packed(packed, offsets, node=name).cpu().numpy()
        ref = np.concatenate([e[name].reshape(-1, e[name].shape[-1]) for e in eps], axis=0)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = _rel(got, ref)
        _note("resnet_w%d%s" % (width, "_max" if maxpool else ""), precision, name, err)
        assert err <= TOL, (name, err)
    tr.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_resnet18_full_width_xvector(precision):
    """config 5 graph at full width (64..512 channels, 13.5 M parameters), short utterances."""
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.RESNET_PARAMS)
    weights = synth.synth_resnet_weights(params, seed=0)
    feats = np.stack(synth.synth_features(2, 24, 40, seed=16))
    ref = ref_numpy.predict(feats, weights, params, 40)
    tr, _ = _trainer(params, weights, 40, precision)
    got = tr.predict(feats)
    err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    _note("resnet_full", precision, "tdnn6_dense", err.max())
    assert err.max() <= TOL, err
    tr.close()


@pytest.mark.parametrize("precision", ["bf16x3", "f16f6", "f32"])
def test_resnet18_full_width_baseline_geometry(precision):
    """config 5 at its BASELINE geometry: full width (13.5 M parameters), 40-dim x 300-frame utterances (B = 2; the float64 oracle
    on one of them takes a few seconds), plus a 301-frame neighbour in the pack so that the utterance checked does not start the
    grid.  Everything else about the ResNet is tested on short utterances; this is the shape the bench measures."""
    import torch
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.RESNET_PARAMS)
    weights = synth.synth_resnet_weights(params, seed=0)
    lens = [301, 300]
    utts = synth.synth_features(2, lens, 40, seed=19)
    tr, _ = _trainer(params, weights, 40, precision)
    packed = torch.from_numpy(np.concatenate(utts, axis=0)).cuda()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    got = tr.predict_packed(packed, offsets).cpu().numpy()
    ref = ref_numpy.predict(utts[1], weights, params, 40)
    err = _rel(got[1], ref)
    _note("resnet_full_T300", precision, "tdnn6_dense", err)
    assert err <= TOL, err
    tr.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_xvector_300_frames(stat_model, precision):
    """BASELINE config 2 shape (30-dim x 300 frames), small batch, tdnn6_dense."""
    from tf_kaldi_speaker_amd import synth
    params, weights = stat_model
    feats = np.stack(synth.synth_features(4, 300, 30, seed=1234))
    ref = ref_numpy.predict(feats, weights, params, 30)
    tr, _ = _trainer(params, weights, 30, precision)
    got = tr.predict(feats)
    err = np.linalg.norm(got - ref, axis=1) / np.linalg.norm(ref, axis=1)
    _note("xvector_300", precision, "tdnn6_dense", err.max())
    assert err.max() <= TOL, err
    # rank-2 input squeezes (model/trainer.py:911-912); extra feature columns are dropped (:906-907)
    one = tr.predict(np.concatenate([feats[1], np.ones((300, 3), np.float32)], axis=1))
    assert one.shape == (512,)
    assert _rel(one, ref[1]) <= TOL
    tr.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_ragged_batch_matches_per_utterance(stat_model, precision):
    """Packed ragged batch (config 4 shape) == each utterance alone (oracle)."""
    import torch
    from tf_kaldi_speaker_amd import synth
    params, weights = stat_model
    lens = [15, 200, 16, 333, 25, 64, 1000, 129]
    utts = synth.synth_features(len(lens), lens, 30, seed=5)
    tr, _ = _trainer(params, weights, 30, precision)
    packed = torch.from_numpy(np.concatenate(utts, axis=0)).cuda()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    got = tr.predict_packed(packed, offsets).cpu().numpy()
    for i, u in enumerate(utts):
        ref = ref_numpy.predict(u, weights, params, 30)
        _note("ragged", precision, "T=%d" % lens[i], _rel(got[i], ref))
        assert _rel(got[i], ref) <= TOL, (i, lens[i])
    # frame-level node on the ragged batch: packed rows in utterance order
    frames = tr.predict_packed(packed, offsets, node="tdnn3_relu").cpu().numpy()
    pos = 0
    for i, u in enumerate(utts):
        ref = ref_numpy.predict(u, weights, params, 30, node="tdnn3_relu")
        n = ref.shape[0]
        assert _rel(frames[pos:pos + n], ref) <= TOL, i
        pos += n
    assert pos == frames.shape[0]
    tr.close()


@pytest.mark.parametrize("width,precision", [(8, "f32"), (32, "f32"), (32, "bf16x3"), (32, "f16x3"), (32, "f16f6")])
def test_resnet18_time_stride_every_block(width, precision):
    """resnet_time_stride (model/resnet.py:187,239,244,249): stride 2 along time in the first block of stages 2-4 under
    TensorFlow's 'same' padding, whose placement depends on the parity of each utterance's length (even: 0 before / 1
    after, odd: 1 / 1).  Ragged batch with even and odd lengths at every level (16 -> 8 -> 4 -> 2, 9 -> 5 -> 3 -> 2,
    14 -> 7 -> 4 -> 2, 3 -> 2 -> 1 -> 1, 1 -> 1 -> 1 -> 1), every block output against the oracle."""
    import torch
    from tf_kaldi_speaker_amd import synth
    params = dict(synth.RESNET_PARAMS, num_nodes_pooling_layer=96, network_relu_type="lrelu", resnet_time_stride=True,
                  resnet_maxpooling=True)
    weights = synth.synth_resnet_weights(params, seed=7, width=width)
    lens = [16, 9, 14, 3, 1]
    utts = synth.synth_features(len(lens), lens, 40, seed=17)
    tr, _ = _trainer(params, weights, 40, precision)
    packed = torch.from_numpy(np.concatenate(utts, axis=0)).cuda()
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    eps = [ref_numpy.entire_network(u[None], weights, params)[1] for u in utts]
    assert eps[1]["conv2a"].shape[1] == 5 and eps[1]["conv4a"].shape[1] == 2 and eps[3]["conv5_relu"].shape[1] == 1
    for name in eps[0]:
        got = tr.predict_packed(packed, offsets, node=name).cpu().numpy()
        ref = np.concatenate([e[name].reshape(-1, e[name].shape[-1]) for e in eps], axis=0)
        assert got.shape == ref.shape, (name, got.shape, ref.shape)
        err = _rel(got, ref)
        _note("resnet_ts_w%d" % width, precision, name, err)
        assert err <= TOL, (name, err)
    tr.close()


def test_too_short_utterance_raises(stat_model):
    params, weights = stat_model
    tr, _ = _trainer(params, weights, 30)
    with pytest.raises(ValueError):
        tr.predict(np.zeros((14, 30), np.float32))
    tr.close()


@pytest.mark.parametrize("precision", PRECISIONS)
def test_against_committed_oracle_embeddings(precision):
    """The HIP path against vectors committed under tests/golden/ (generated by make_oracle_golden.py from the float64
    oracle): a check that does not depend on running the oracle on this box."""
    import importlib.util
    import os
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(golden, "make_oracle_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from tf_kaldi_speaker_amd import synth
    with np.load(os.path.join(golden, "oracle_embeddings.npz"), allow_pickle=False) as z:
        for name, (params, dim, wseed, fseed, frames) in mod.CASES.items():
            if params.get("network_type") == "resnet_18":
                weights = synth.synth_resnet_weights(params, seed=wseed)
            else:
                weights = synth.synth_weights(params, dim, seed=wseed)
            feats = synth.synth_features(1, frames, dim, seed=fseed)[0]
            tr, _ = _trainer(params, weights, dim, precision)
            err = _rel(tr.predict(feats), z[name])
            tr.close()
            _note("committed_golden", precision, name, err)
            assert err <= TOL, (name, err)
