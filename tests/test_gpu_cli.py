"""GPU: the ark-in -> ark-out command line (egs/voxceleb/v1/nnet/lib/extract.py surface) end to
end through the HIP path, against the oracle's restatement of the driver semantics."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import ref_numpy

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("normalize", [False, True])
def test_extract_cli_matches_oracle(tmp_path, repo_root, normalize):
    from tf_kaldi_speaker_amd import kaldi_io, model_io, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=160, num_nodes_last_layer=48)
    weights = synth.synth_weights(params, 30, seed=3, channels=64)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, weights, step=1234)
    lens = [24, 25, 60, 99, 100, 101, 151, 260, 30, 400]
    utts = synth.synth_features(len(lens), lens, 33, seed=8)          # 3 extra columns: dropped
    ark = str(tmp_path / "feats.ark")
    with open(ark, "wb") as f:
        for i, u in enumerate(utts):
            kaldi_io.write_mat(f, u, key="spk%d-utt%d" % (i % 3, i))
    out = str(tmp_path / "xvector.ark")
    cmd = [sys.executable, "-m", "tf_kaldi_speaker_amd.extract", "--gpu", "0", "--node", "tdnn6_dense",
           "--min-chunk-size", "25", "--chunk-size", "100", "--batch-frames", "500", "--precision", "f32"]
    if normalize:
        cmd.append("--normalize")
    cmd += [model_dir, "ark:cat %s |" % ark, "ark:| cat > %s" % out]
    env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = list(kaldi_io.read_vec_flt_ark(out))

    def predict(x):
        return ref_numpy.predict(x, weights, params, 30)

    expect = []
    for i, u in enumerate(utts):
        e = ref_numpy.extract_utterance(u, predict, 25, 100, normalize)
        if e is not None:
            expect.append(("spk%d-utt%d" % (i % 3, i), e))
    assert [k for k, _ in got] == [k for k, _ in expect]              # T=24 skipped, order kept
    for (k, a), (_, b) in zip(got, expect):
        assert a.dtype == np.float32 and a.shape == (64,)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 1e-4, k


def test_extract_cli_reruns_a_batch_outside_the_fp16_range(tmp_path, repo_root):
    """Default precision (f16f6), several batches in flight, one utterance with a feature of 1e5 (beyond the fp16 range) in the
    middle and one whose features are all below 2^-8: the reference accepts any finite features, so the driver runs the batches
    that hold them again on the bf16x3 twin and every vector still meets the parity bar, in order."""
    from tf_kaldi_speaker_amd import kaldi_io, model_io, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=160, num_nodes_last_layer=48)
    weights = synth.synth_weights(params, 30, seed=3, channels=128)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, weights, step=1)
    lens = [40, 55, 70, 33, 90, 64, 48, 77, 52, 61, 45, 80]
    utts = synth.synth_features(len(lens), lens, 30, seed=21)
    utts[5][10, 3] = 1.0e5
    utts[9] = (utts[9] * 2.0 ** -14).astype(np.float32)
    ark = str(tmp_path / "feats.ark")
    with open(ark, "wb") as f:
        for i, u in enumerate(utts):
            kaldi_io.write_mat(f, u, key="utt%02d" % i)
    out = str(tmp_path / "xvector.ark")
    cmd = [sys.executable, "-m", "tf_kaldi_speaker_amd.extract", "--gpu", "0", "--node", "tdnn6_dense", "--batch-frames", "130",
           model_dir, "ark:%s" % ark, "ark:%s" % out]
    env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    env.pop("XVEC_PRECISION", None)
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "bf16x3" in r.stderr                                        # the warning that names the reason
    got = list(kaldi_io.read_vec_flt_ark(out))
    assert [k for k, _ in got] == ["utt%02d" % i for i in range(len(lens))]
    for (k, a), u in zip(got, utts):
        b = ref_numpy.predict(u, weights, params, 30)
        assert np.linalg.norm(a - b) / np.linalg.norm(b) <= 1e-4, k
    # and without the fallback the driver stops with the message
    r = subprocess.run(cmd, env=dict(env, XVEC_RANGE_FALLBACK="0"), cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "fp16 range" in r.stderr


def test_scp_rspecifier_is_refused(tmp_path, repo_root):
    """extract.py:59-61."""
    from tf_kaldi_speaker_amd import model_io, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=32)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, synth.synth_weights(params, 30, channels=32), step=1)
    env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "tf_kaldi_speaker_amd.extract", model_dir, "feats.scp", "ark:out.ark"],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "rspecifier must be ark" in r.stderr


def test_extract_frame_and_attention_cli(tmp_path, repo_root):
    """extract_frame.py / extract_attention.py surfaces end to end (matrix ark out)."""
    from tf_kaldi_speaker_amd import extract_frame, kaldi_io, model_io, synth
    params = dict(synth.TDNN_ATT_PARAMS, num_nodes_pooling_layer=96, att_key_num_nodes=[64, 48], att_num_heads=2,
                  embedding_node="tdnn6_dense")
    weights = synth.synth_weights(params, 30, seed=3, channels=64)
    weights["tdnn/attention/query"] = weights["tdnn/attention/query"] * 50.0
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, weights, step=7)
    lens = [20, 40, 130, 77]
    utts = synth.synth_features(len(lens), lens, 30, seed=9)
    ark = str(tmp_path / "feats.ark")
    with open(ark, "wb") as f:
        for i, u in enumerate(utts):
            kaldi_io.write_mat(f, u, key="u%d" % i)
    env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    out_f = str(tmp_path / "frames.ark")
    r = subprocess.run([sys.executable, "-m", "tf_kaldi_speaker_amd.extract_frame", "--gpu", "0", "--node", "tdnn4_relu",
                        "--chunk-size", "50", "--precision", "f32", model_dir, "ark:" + ark, "ark:" + out_f],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    got = dict(kaldi_io.read_mat_ark(out_f))
    assert list(got) == ["u1", "u2", "u3"]
    for i in (1, 2, 3):
        u = utts[i]
        parts = []
        for s0, n in extract_frame.split_plain(u.shape[0], 50) if u.shape[0] > 50 else [(0, u.shape[0])]:
            e = ref_numpy.predict(u[s0:s0 + n], weights, params, 30, node="tdnn4_relu")
            parts.append(extract_frame.pad_edges(e, n))
        ref = np.concatenate(parts, axis=0)
        assert got["u%d" % i].shape == (u.shape[0], 64)
        assert np.linalg.norm(got["u%d" % i] - ref) / np.linalg.norm(ref) <= 1e-4
    out_a = str(tmp_path / "att.ark")
    r = subprocess.run([sys.executable, "-m", "tf_kaldi_speaker_amd.extract_attention", "--gpu", "0", "--chunk-size", "100",
                        "--precision", "f32", model_dir, "ark:" + ark, "ark:" + out_a],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    att = dict(kaldi_io.read_mat_ark(out_a))
    assert list(att) == ["u1", "u2", "u3"]
    for i in (1, 2, 3):
        u = utts[i][:100]
        ref = ref_numpy.predict(u, weights, params, 30, node="attention_weights")
        assert att["u%d" % i].shape == ref.shape == (2, u.shape[0] - 14)
        assert np.linalg.norm(att["u%d" % i] - ref) / np.linalg.norm(ref) <= 1e-4


def test_extract_angle_matches_oracle(tmp_path, repo_root):
    """extract_angle.py surface: scp in, `key angle` lines out, angle to the speaker's softmax weight vector."""
    from tf_kaldi_speaker_amd import extract_angle, kaldi_io, model_io, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=160, num_nodes_last_layer=48)
    weights = synth.synth_weights(params, 30, seed=5, channels=64)
    rng = np.random.default_rng(1)
    weights[model_io.SOFTMAX_KERNEL] = rng.standard_normal((48, 4)).astype(np.float32)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, weights, step=10)
    lens = [30, 24, 80, 120]
    utts = synth.synth_features(len(lens), lens, 30, seed=2)
    ark = str(tmp_path / "feats.ark")
    with open(ark, "wb") as f, open(tmp_path / "feats.scp", "w") as scp, open(tmp_path / "utt2spk", "w") as u2s:
        for i, u in enumerate(utts):
            key = "utt%d" % i
            f.write((key + " ").encode())
            scp.write("%s %s:%d\n" % (key, ark, f.tell()))
            kaldi_io.write_mat(f, u)
            u2s.write("%s spk%d\n" % (key, i % 4))
    with open(tmp_path / "spklist", "w") as f:
        for i in range(4):
            f.write("spk%d %d\n" % (i, i))
    rc = extract_angle.main(["--chunk-size", "100", "--precision", "f32", model_dir, str(tmp_path / "feats.scp"),
                             str(tmp_path / "utt2spk"), str(tmp_path / "spklist"), str(tmp_path / "angles")])
    assert rc == 0
    got = [l.split() for l in open(tmp_path / "angles")]
    assert [g[0] for g in got] == ["utt0", "utt2", "utt3"]                 # T=24 < 25 skipped
    cw = np.transpose(weights[model_io.SOFTMAX_KERNEL])
    net = {k: v for k, v in weights.items() if k != model_io.SOFTMAX_KERNEL}
    for (key, val), i in zip(got, [0, 2, 3]):
        out = ref_numpy.predict(utts[i][:100], net, params, 30, node="output")
        want = extract_angle.angle(cw[i % 4].astype(np.float64), out)
        assert abs(float(val) - want) < 5e-5, (key, val, want)
