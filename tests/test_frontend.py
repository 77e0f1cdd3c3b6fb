"""Feature front-end (sliding CMN + voiced-frame selection): oracle properties on the CPU, HIP-vs-oracle parity
on the GPU.  Parity with Kaldi itself is unpinned (no Kaldi binary or fixture here, see oracle/ref_frontend.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_frontend  # noqa: E402


def test_oracle_cmn_short_utterance_is_global_mean():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((120, 7)).astype(np.float32) + 3.0
    y = ref_frontend.sliding_cmn(x, cmn_window=300, center=True)
    np.testing.assert_allclose(y, x - x.astype(np.float64).mean(0), atol=1e-6)


def test_oracle_cmn_window_positions():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((1000, 3)).astype(np.float32)
    y = ref_frontend.sliding_cmn(x, cmn_window=300, center=True)
    x64 = x.astype(np.float64)
    np.testing.assert_allclose(y[0], x[0] - x64[0:300].mean(0), atol=1e-6)          # left edge: window [0, 300)
    np.testing.assert_allclose(y[500], x[500] - x64[350:650].mean(0), atol=1e-6)    # interior: [t-150, t+150)
    np.testing.assert_allclose(y[999], x[999] - x64[700:1000].mean(0), atol=1e-6)   # right edge: [T-300, T)
    z = ref_frontend.sliding_cmn(x, cmn_window=300, center=False, min_window=100)
    np.testing.assert_allclose(z[10], x[10] - x64[0:100].mean(0), atol=1e-6)        # causal: at least min_window
    np.testing.assert_allclose(z[500], x[500] - x64[200:501].mean(0), atol=1e-6)    # causal: [t-W, t]


def test_oracle_select_voiced():
    x = np.arange(20, dtype=np.float32).reshape(10, 2)
    vad = np.array([0, 1, 1, 0, 0, 1, 0, 0, 0, 1], dtype=np.float32)
    np.testing.assert_array_equal(ref_frontend.select_voiced(x, vad), x[[1, 2, 5, 9]])


@pytest.mark.gpu
@pytest.mark.parametrize("center,window", [(True, 300), (True, 51), (False, 300), (True, 0)])
def test_gpu_frontend_matches_oracle(center, window):
    import torch
    import tf_kaldi_speaker_amd as xv
    from tf_kaldi_speaker_amd.frontend import cmn_select_packed
    rng = np.random.default_rng(5)
    lens = [1, 24, 299, 300, 301, 777, 1500, 40]
    feats = [(rng.standard_normal((t, 30)) * 4 + rng.standard_normal(30) * 10).astype(np.float32) for t in lens]
    vads = [(rng.random(t) < 0.7).astype(np.float32) for t in lens]
    vads[0][:] = 0                                     # an utterance with no voiced frame is dropped
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    dev = torch.from_numpy(np.concatenate(feats)).cuda()
    out, new_off, kept = cmn_select_packed(dev, offsets, vads, cmn_window=window, center=center, min_window=100, min_frames=10)
    out = out.cpu().numpy()
    want, want_kept = [], []
    for i, (f, v) in enumerate(zip(feats, vads)):
        g = ref_frontend.sliding_cmn(f, window, center, 100) if window > 0 else f
        g = ref_frontend.select_voiced(g, v)
        if g.shape[0] >= 10:
            want.append(g)
            want_kept.append(i)
    assert list(kept) == want_kept
    np.testing.assert_array_equal(new_off, np.concatenate([[0], np.cumsum([w.shape[0] for w in want])]))
    want = np.concatenate(want)
    assert out.shape == want.shape
    if window > 0:
        np.testing.assert_allclose(out, want, atol=2e-6 * np.abs(want).max())   # double sums in a different order
    else:
        np.testing.assert_array_equal(out, want)
    # no VAD: every frame kept
    out2, off2, kept2 = cmn_select_packed(dev, offsets, None, cmn_window=300)
    np.testing.assert_array_equal(off2, offsets)
    w2 = np.concatenate([ref_frontend.sliding_cmn(f, 300, True) for f in feats])
    np.testing.assert_allclose(out2.cpu().numpy(), w2, atol=2e-6 * np.abs(w2).max())


@pytest.mark.gpu
def test_gpu_cli_with_frontend(tmp_path):
    """extract.py --cmn-window/--vad-rspecifier == oracle front-end + oracle network on the same raw features."""
    from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
    from oracle import ref_numpy
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=160, num_nodes_last_layer=48)
    weights = synth.synth_weights(params, 30, seed=3, channels=64)
    model_dir = str(tmp_path / "model")
    model_io.save_model(model_dir, params, 30, weights, step=7)
    rng = np.random.default_rng(9)
    lens = [200, 60, 450, 90, 35]
    raw = [(rng.standard_normal((t, 30)) + rng.standard_normal(30) * 5).astype(np.float32) for t in lens]
    vads = [(rng.random(t) < 0.6).astype(np.float32) for t in lens]
    vads[4][:] = 0
    vads[4][:20] = 1                                   # 20 voiced frames < min chunk 25: skipped
    keys = ["utt%d" % i for i in range(len(lens))]
    with open(tmp_path / "raw.ark", "wb") as f:
        for k, m in zip(keys, raw):
            kaldi_io.write_mat(f, m, key=k)
    with open(tmp_path / "vad.ark", "wb") as f:
        for k, v in zip(keys, vads):
            kaldi_io.write_vec_flt(f, v, key=k)
    rc = extract.main(["--cmn-window", "300", "--vad-rspecifier", "ark:%s" % (tmp_path / "vad.ark"), "-s", "250",
                       model_dir, "ark:%s" % (tmp_path / "raw.ark"), "ark:%s" % (tmp_path / "out.ark")])
    assert rc == 0
    got = dict(kaldi_io.read_vec_flt_ark(str(tmp_path / "out.ark")))
    assert list(got) == keys[:4]
    for k, m, v in zip(keys[:4], raw, vads):
        f = ref_frontend.select_voiced(ref_frontend.sliding_cmn(m, 300, True), v)
        want = ref_numpy.extract_utterance(f, lambda a: ref_numpy.predict(a, weights, params, 30), min_chunk_size=25,
                                           chunk_size=250, normalize=False)
        rel = np.linalg.norm(got[k] - want) / np.linalg.norm(want)
        assert rel < 1e-4, (k, rel)
