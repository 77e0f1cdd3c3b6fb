"""GPU: the launcher end to end on one MI355X (two jobs sharing GPU 0, run one after the other): raw features +
VAD tables in, GPU front-end (sliding CMN + voiced-frame selection), x-vectors, ordered xvector.scp, speaker
means and length normalisation -- against the oracle (ref_frontend -> ref_numpy -> ref_post); and the post-step
kernels on their own (csrc/post.hip) through the C ABI."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_frontend, ref_numpy, ref_post

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_launcher import make_data_dir  # noqa: E402


def test_post_step_kernels_match_oracle():
    from tf_kaldi_speaker_amd import postprocess
    rs = np.random.RandomState(3)
    x = rs.standard_normal((257, 512)).astype(np.float32) * 7
    x[5] = 0.0                                                        # "Zero iVector": written unchanged
    for scaleup in (False, True):
        got = postprocess.length_normalize(x, scaleup)
        want = ref_post.normalize_length(x, scaleup)
        assert np.array_equal(got[5], x[5])
        np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    keys = ["u%03d" % i for i in range(257)]
    spk2utt = [("s%d" % s, ["u%03d" % i for i in range(s, 300, 9)]) for s in range(9)] + [("empty", ["nope"])]
    spks, means, counts = postprocess.speaker_mean(keys, x, spk2utt)
    want_m, want_c = ref_post.speaker_mean(dict(zip(keys, x)), spk2utt)
    assert spks == [s for s, _ in want_m] and list(counts) == [n for _, n in want_c]
    assert np.array_equal(means, np.stack([m for _, m in want_m]))                 # same float32 adds, same order: bit-exact


@pytest.mark.parametrize("normalize", [False, True])
def test_launcher_end_to_end_on_gpu(tmp_path, normalize):
    from tf_kaldi_speaker_amd import kaldi_io, model_io, native_ark, run_extract, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=160, num_nodes_last_layer=48, embedding_node="tdnn6_dense")
    weights = synth.synth_weights(params, 30, seed=3, channels=64)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, weights, step=77)
    lens = list(np.random.RandomState(11).randint(40, 700, size=23)) + [30]
    data, feats, vads = make_data_dir(tmp_path, lens, dim=30, seed=5, n_arks=3)
    out = str(tmp_path / "xv")
    argv = ["--nj", "2", "--gpus", "0", "--min-chunk-size", "25", "--chunk-size", "300", "--batch-frames", "1500",
            "--normalize", "true" if normalize else "false", "--node", "tdnn6_dense", "--precision", "f32", "--checkpoint", "last",
            model_dir, data, out]
    rc = run_extract.main(argv)
    logs = "".join(open(os.path.join(out, "log", "extract.%d.log" % j)).read()[-1500:] for j in (1, 2))
    assert rc == 0, logs

    def predict(x):
        return ref_numpy.predict(x, weights, params, 30)

    expect = {}
    for k, f in feats.items():
        x = ref_frontend.select_voiced(ref_frontend.sliding_cmn(f, 300), vads[k])
        e = ref_numpy.extract_utterance(x, predict, 25, 300, normalize)
        if e is not None:
            expect[k] = e
    scp = os.path.join(out, "xvector_before_norm.scp" if normalize else "xvector.scp")
    table = native_ark.read_scp_table(scp)
    assert [k for k, _ in table] == [k for k in feats if k in expect]                # input order, short ones dropped
    got = {k: kaldi_io.read_vec_flt(rx) for k, rx in table}
    for k, e in expect.items():
        assert np.linalg.norm(got[k] - e) / np.linalg.norm(e) <= 1e-4, k
    spk2utt = [(l.split()[0], l.split()[1:]) for l in open(os.path.join(data, "spk2utt"))]
    xin = {k: (ref_post.normalize_length(v[None])[0] if normalize else v) for k, v in got.items()}
    means, counts = ref_post.speaker_mean(xin, spk2utt)
    spk = native_ark.read_scp_table(os.path.join(out, "spk_xvector.scp"))
    assert [s for s, _ in spk] == [s for s, _ in means]
    for (s, rx), (_, m) in zip(spk, means):
        want = ref_post.normalize_length(m[None])[0] if normalize else m
        np.testing.assert_allclose(kaldi_io.read_vec_flt(rx), want, rtol=2e-6, atol=1e-8)
    assert open(os.path.join(out, "num_utts.ark")).read() == "".join("%s %d\n" % sc for sc in counts)
    if normalize:
        for k, rx in native_ark.read_scp_table(os.path.join(out, "xvector.scp")):
            np.testing.assert_allclose(kaldi_io.read_vec_flt(rx), ref_post.normalize_length(got[k][None])[0], rtol=2e-6, atol=1e-8)


def test_gpu_minus_one_spreads_jobs(tmp_path, repo_root):
    """`--gpu -1` with a reference-style wspecifier (xvector.JOB.ark): the job picks (JOB-1) mod #GPUs, so on this
    one-GPU box job 3 runs on device 0 and says so."""
    import subprocess
    from tf_kaldi_speaker_amd import kaldi_io, model_io, synth
    params = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=32)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, params, 30, synth.synth_weights(params, 30, channels=32), step=1)
    ark = str(tmp_path / "feats.ark")
    with open(ark, "wb") as f:
        kaldi_io.write_mat(f, synth.synth_features(1, [60], 30, seed=1)[0], key="u0")
    env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-m", "tf_kaldi_speaker_amd.extract", "--gpu", "-1", "--node", "tdnn6_dense", model_dir,
                        "ark:" + ark, "ark,scp:%s,%s" % (tmp_path / "xvector.3.ark", tmp_path / "xvector.3.scp")],
                       env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "Using HIP device 0" in r.stderr
    assert len(open(tmp_path / "xvector.3.scp").read().split()) == 2
