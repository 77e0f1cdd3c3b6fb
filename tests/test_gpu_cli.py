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
