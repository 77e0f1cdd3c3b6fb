"""CPU: the oracle against the reference's own pins (tests/golden/, made by make_golden.py from
/root/reference/model/test_utils.py) and the two restatements against each other.

The TDNN layers themselves are **parity unpinned** (the reference holds no golden vector for
them and TensorFlow is not installed): for those the check is numpy-f64 vs torch-f32
(independent formulations: shifted matmuls vs F.conv1d)."""
import glob
import os

import numpy as np
import pytest

from oracle import ref_numpy, ref_torch
from tf_kaldi_speaker_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLD, "attention_*.npz"))))
def test_attention_core_matches_reference_numpy_twin(path):
    """model/test_utils.py:321-392 compute_self_attention (fixtures attention_*.npz)."""
    z = np.load(path)
    split = bool(z["split"])
    att, w = ref_numpy.attention_core(z["value"], z["key"], z["query"], int(z["heads"]), split, split, bool(z["use_scale"]))
    np.testing.assert_allclose(att, z["att"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(w.sum(-1), 1.0, rtol=1e-12)


def test_statistics_pooling_matches_reference_mean_std():
    """model/test_utils.py:1114 (np.mean || np.std) + the 1e-12 variance floor of pooling.py:46-48."""
    z = np.load(os.path.join(GOLD, "stat_pool.npz"))
    got = ref_numpy.statistics_pooling(z["x"])
    ref = z["mean_std"].copy()
    c = z["x"].shape[2]
    ref[:, c:] = np.maximum(ref[:, c:], 1e-6)          # sqrt(1e-12): reference floors the variance
    np.testing.assert_allclose(got, ref, rtol=1e-10, atol=1e-13)
    assert np.all(got[:, c + 5] == 1e-6)                # the constant channel hits the floor


def _small_params(pooling, **kw):
    p = dict(synth.TDNN_ATT_PARAMS if pooling == "self_attention" else synth.TDNN_STAT_PARAMS)
    p["num_nodes_pooling_layer"] = 40
    p["num_nodes_last_layer"] = 24
    if pooling == "self_attention":
        p["att_key_num_nodes"] = [20, 16]
    p.update(kw)
    return p


@pytest.mark.parametrize("pooling,kw", [
    ("statistics_pooling", {}),
    ("statistics_pooling", {"network_relu_type": "prelu", "last_layer_linear": True}),
    ("statistics_pooling", {"network_relu_type": "lrelu", "last_layer_no_bn": True, "feature_norm": True,
                            "feature_scaling_factor": 30.0}),
    ("self_attention", {}),
    ("self_attention", {"att_num_heads": 4, "att_apply_nonlinear": True, "network_relu_type": "prelu"}),
    ("self_attention", {"att_num_heads": 2, "att_split_key": False, "att_split_value": False, "att_key_network_type": 3,
                        "att_value_num_nodes": [12], "att_value_network_type": 2, "att_use_scale": False}),
])
def test_numpy_and_torch_restatements_agree(pooling, kw):
    p = _small_params(pooling, **kw)
    w = synth.synth_weights(p, 7, seed=3, channels=32)
    if "tdnn/attention/query" in w:
        w["tdnn/attention/query"] = w["tdnn/attention/query"] * 20.0    # non-uniform attention
    feats = np.stack(synth.synth_features(3, 33, 7, seed=5))
    _, ep = ref_numpy.entire_network(feats, w, p)
    ep_t = ref_torch.TorchTdnn(w, p).forward(feats)
    assert list(ep.keys()) == list(ep_t.keys())
    for k in ep:
        a, b = ep[k], ep_t[k].numpy().astype(np.float64)
        assert a.shape == b.shape, k
        err = np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30)
        assert err < 2e-6, (k, err)


@pytest.mark.parametrize("pooling", ["statistics_pooling", "self_attention"])
def test_extended_tdnn_restatements_agree(pooling):
    """model/tdnn.py:343-591: 10 frame layers (conv1d k=5,5,7,9 + dense), tdnn12/tdnn13 segment layers."""
    p = _small_params(pooling, network_type="extended_tdnn", embedding_node="tdnn12_dense")
    if pooling == "self_attention":
        p.update(att_key_input="tdnn8_relu", att_value_input="tdnn10_relu")
    w = synth.synth_weights(p, 7, seed=3, channels=32)
    assert w["etdnn/tdnn7_conv/kernel"].shape == (9, 32, 32)            # conv1d kernels are rank 3
    feats = np.stack(synth.synth_features(2, 40, 7, seed=5))
    _, ep = ref_numpy.entire_network(feats, w, p)
    ep_t = ref_torch.TorchTdnn(w, p).forward(feats)
    assert list(ep.keys()) == list(ep_t.keys())
    assert ep["tdnn7_relu"].shape == (2, 40 - 22, 32) and "tdnn13_relu" in ep and "tdnn11_dense" not in ep
    for k in ep:
        a, b = ep[k], ep_t[k].numpy().astype(np.float64)
        assert np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30) < 2e-6, k


@pytest.mark.parametrize("kw", [{}, {"network_relu_type": "prelu", "resnet_blocks": [1, 2, 1, 3]},
                                {"network_relu_type": "lrelu", "resnet_maxpooling": True},
                                {"resnet_time_stride": True, "resnet_maxpooling": True},
                                {"network_relu_type": "lrelu", "last_layer_linear": True}])
def test_resnet18_restatements_agree(kw):
    """model/resnet.py:152-351: numpy tap-loop conv vs torch F.conv2d with explicit TF 'same' padding
    (stride-2 frequency axis pads 0 before / 1 after)."""
    p = dict(synth.RESNET_PARAMS, num_nodes_pooling_layer=48, **kw)
    w = synth.synth_resnet_weights(p, seed=1, width=8)
    feats = np.stack(synth.synth_features(2, 13, 40, seed=3))
    _, ep = ref_numpy.entire_network(feats, w, p)
    ep_t = ref_torch.TorchResnet18(w, p).forward(feats)
    t2, t4 = (7, 2) if kw.get("resnet_time_stride") else (13, 13)                   # 'same', stride 2: 13 -> 7 -> 4 -> 2
    assert ep["conv2a"].shape == (2, t2, 20, 16) and ep["conv4a"].shape == (2, t4, 5, 64)      # 40 -> 20 -> 10 -> 5
    assert ep["conv5_relu"].shape == (2, t4, 64) and ep["tdnn6_dense"].shape == (2, 64)
    for k in ep:
        a, b = ep[k], ep_t[k].numpy().astype(np.float64)
        assert a.shape == b.shape, k
        assert np.linalg.norm(a - b) / max(np.linalg.norm(a), 1e-30) < 2e-6, k


def test_resnet18_parameter_count_matches_reference_formula():
    """model/resnet.py:292-297 `num_params` (frame-level network, BN excluded) for blocks [2,2,2,2]: 13.5 M (:170)."""
    w = synth.synth_resnet_weights(dict(synth.RESNET_PARAMS), seed=0)
    got = sum(v.size for k, v in w.items() if k.endswith("/kernel") and "tdnn" not in k)
    blocks = [2, 2, 2, 2]
    ref = 3*3*64 + (2*3*3*64*64*blocks[0] + 64*64) + \
        (3*3*64*128 + 3*3*128*128 + 64*128 + 2*3*3*128*128*(blocks[1]-1)) + \
        (3*3*128*256 + 3*3*256*256 + 128*256 + 2*3*3*256*256*(blocks[2]-1)) + \
        (3*3*256*512 + 3*3*512*512 + 256*512 + 2*3*3*512*512*(blocks[3]-1)) + \
        (1*5*512*512 + 512*512 + 512*1500)
    assert got == ref == 13503040


def test_tf_same_padding_stride2_is_asymmetric():
    """k=3, s=2, n even: pad 0 before / 1 after (TF 'same'), unlike a symmetric pad of 1."""
    x = np.arange(1 * 1 * 6 * 1, dtype=np.float64).reshape(1, 1, 6, 1)
    k = np.ones((1, 3, 1, 1))
    y = ref_numpy.conv2d(x, k, strides=(1, 2))
    assert y[0, 0, :, 0].tolist() == [0 + 1 + 2, 2 + 3 + 4, 4 + 5]


def test_full_size_tdnn_restatements_agree():
    p = dict(synth.TDNN_STAT_PARAMS)
    w = synth.synth_weights(p, 30, seed=0)
    feats = np.stack(synth.synth_features(1, 60, 30, seed=1))
    a = ref_numpy.predict(feats, w, p, 30)
    b = ref_torch.TorchTdnn(w, p).predict(feats, 30)
    assert a.shape == (1, 512)
    assert np.linalg.norm(a - b) / np.linalg.norm(a) < 1e-6


def test_conv_is_valid_cross_correlation_no_flip():
    """y[t,o] = sum_k sum_c x[t+k,c] W[0,k,c,o] + b[o] (model/tdnn.py:42-47, TF 'valid', stride 1)."""
    rs = np.random.RandomState(0)
    x = rs.standard_normal((1, 9, 3))
    k = rs.standard_normal((1, 5, 3, 4))
    b = rs.standard_normal(4)
    y = ref_numpy.conv_valid(x, k, b)
    assert y.shape == (1, 5, 4)
    for t in range(5):
        for o in range(4):
            ref = sum(x[0, t + j, c] * k[0, j, c, o] for j in range(5) for c in range(3)) + b[o]
            assert abs(y[0, t, o] - ref) < 1e-12


def test_predict_rank_handling_and_dim_truncation():
    """model/trainer.py:897-912."""
    p = _small_params("statistics_pooling")
    w = synth.synth_weights(p, 7, seed=3, channels=32)
    f = synth.synth_features(2, 20, 9, seed=2)
    e3 = ref_numpy.predict(np.stack(f), w, p, 7)
    e2 = ref_numpy.predict(f[1], w, p, 7)
    assert e3.shape == (2, 32) and e2.shape == (32,)
    np.testing.assert_allclose(e2, e3[1], rtol=1e-12)
    np.testing.assert_allclose(ref_numpy.predict(f[1][:, :7], w, p, 7), e2, rtol=1e-12)
    assert ref_numpy.predict(f[0], w, p, 7, node="tdnn3_relu").shape == (6, 32)


def test_oracle_reproduces_its_committed_embeddings():
    """Regression goldens of the oracle itself (tests/golden/make_oracle_golden.py): a change of ref_numpy or of the
    synthetic generators shows up here.  Not a pin from the reference."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_oracle_golden", os.path.join(GOLD, "make_oracle_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with np.load(os.path.join(GOLD, "oracle_embeddings.npz"), allow_pickle=False) as z:
        assert set(z.files) == set(mod.CASES)
        for name in ("tdnn_stat", "tdnn_att"):          # the other two are checked on the GPU side (slower oracle)
            got = mod.compute(name)
            assert np.linalg.norm(got - z[name]) / np.linalg.norm(z[name]) < 1e-12, name
