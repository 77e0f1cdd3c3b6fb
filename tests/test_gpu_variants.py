"""GPU: the alternative code paths that the default run never takes.

* randomised ragged batches (seeded) through the default kernels: lengths from the network's minimum context to
  several M tiles, batch sizes that leave partial tiles, both poolings, embedding and frame-level nodes;
* the two other bf16x3 GEMM kernels (XVEC_GEMM_TILE=1: LDS-DMA weights, =128: register-staged fallback that shapes
  with more than 9 taps fall back to) in a child process each, against the same oracle;
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


def _trainer(params, weights, dim, precision):
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd.trainer import Trainer
    tr = Trainer(Params(**dict(params)), None, dim, single_cpu=True, device=0, precision=precision)
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
    for precision in ("bf16x3", "f32"):
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


_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
from oracle import ref_numpy
worst = 0.0
for net, P, dim in (("tdnn", synth.TDNN_STAT_PARAMS, 30), ("etdnn", dict(synth.TDNN_STAT_PARAMS, network_type="extended_tdnn"), 30)):
    params = dict(P)
    weights = synth.synth_weights(params, dim, seed=4)
    utts = synth.synth_features(5, [40, 150, 129, 64, 300], dim, seed=9)
    tr = Trainer(Params(**params), None, dim, single_cpu=True, device=0, precision="bf16x3")
    tr.build("predict"); tr.load_weights(weights)
    got = tr.predict_list(utts)
    for g, u in zip(got, utts):
        ref = ref_numpy.predict(u, weights, params, dim)
        worst = max(worst, float(np.linalg.norm(np.asarray(g, np.float64).reshape(ref.shape) - ref) / np.linalg.norm(ref)))
    tr.close()
print("WORST %%.3e" %% worst)
"""


@pytest.mark.parametrize("tile", ["1", "128"])
def test_alternative_gemm_kernels(tile, repo_root):
    env = dict(os.environ, XVEC_GEMM_TILE=tile, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", _CHILD % {"root": repo_root}], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    worst = float(r.stdout.strip().split("WORST")[-1])
    assert worst <= TOL, (tile, worst)


_TAIL_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r)
import torch
from tf_kaldi_speaker_amd import synth
from tf_kaldi_speaker_amd.params import Params
from tf_kaldi_speaker_amd.trainer import Trainer
params = dict(synth.TDNN_STAT_PARAMS)
weights = synth.synth_weights(params, 30, seed=0)
tr = Trainer(Params(**params), None, 30, single_cpu=True, device=0, precision=%(prec)r)
tr.build("predict"); tr.load_weights(weights)
B, T = 256, 300                                    # L3: 2336 tiles on 768 slots -> 32 tail tiles, L2: 64
feats = torch.from_numpy(np.concatenate(synth.synth_features(B, T, 30, seed=5))).cuda()
offs = np.arange(B + 1, dtype=np.int32) * T
res = {}
for node in ("tdnn6_dense", "tdnn3_conv", "tdnn2_relu"):      # embedding; frame-level fp32 outputs of the two tail layers
    a = tr.predict_packed(feats, offs, node).cpu().numpy()
    b = tr.predict_packed(feats, offs, node).cpu().numpy()
    assert np.array_equal(a, b), "non-deterministic " + node
    res[node] = a
np.savez(%(out)r, **res)
"""


def test_tail_ksplit_matches_plain_and_exact(tmp_path, repo_root):
    """The K-split of the last, nearly empty round of tiles (gemm_bf16x3_tail_plan) against the same kernel without
    it (XVEC_GEMM_TAIL=0) and against the exact fp32 path, at the BASELINE geometry where it is active."""
    outs = {}
    for tag, env_extra, prec in (("tail", {"XVEC_GEMM_TAIL": "1"}, "bf16x3"), ("plain", {"XVEC_GEMM_TAIL": "0"}, "bf16x3"),
                                 ("exact", {}, "f32")):
        out = str(tmp_path / (tag + ".npz"))
        env = dict(os.environ, PYTHONPATH=repo_root + os.pathsep + os.environ.get("PYTHONPATH", ""), **env_extra)
        r = subprocess.run([sys.executable, "-c", _TAIL_CHILD % {"root": repo_root, "prec": prec, "out": out}], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        with np.load(out) as z:
            outs[tag] = {k: z[k].astype(np.float64) for k in z.files}

    def rel(a, b):
        return float(np.linalg.norm(a - b) / np.linalg.norm(b))
    for node in ("tdnn6_dense", "tdnn3_conv", "tdnn2_relu"):
        t, pl, ex = outs["tail"][node], outs["plain"][node], outs["exact"][node]
        assert t.shape == pl.shape == ex.shape, node
        assert rel(t, pl) <= 2e-6, node                      # same products, different summation order in the tail tiles
        assert rel(t, ex) <= TOL, node
    assert not np.array_equal(outs["tail"]["tdnn3_conv"], outs["plain"]["tdnn3_conv"])   # the path really was taken


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
