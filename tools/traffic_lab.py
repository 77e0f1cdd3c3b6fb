"""Forward of the BASELINE geometry (256 x [300, 30], bf16x3) with the library named by XVEC_TLAB_LIB; prints the per-layer
hipEvent times.  Lab variants with bits 0/1 compute wrong results on purpose (tools/traffic_lab.sh)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from tf_kaldi_speaker_amd import _lib  # noqa: E402

if os.environ.get("XVEC_TLAB_LIB"):
    _lib.LIB_PATH = os.environ["XVEC_TLAB_LIB"]
import torch  # noqa: E402
from tf_kaldi_speaker_amd import synth  # noqa: E402
from tf_kaldi_speaker_amd.params import Params  # noqa: E402
from tf_kaldi_speaker_amd.trainer import Trainer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
params = Params(**dict(synth.TDNN_STAT_PARAMS))
tr = Trainer(params, None, 30, single_cpu=True, device=0, precision=os.environ.get("XVEC_TLAB_PRECISION", "bf16x3"))
tr.build("predict")
tr.load_weights(synth.synth_weights(params, 30, seed=0))
feats = torch.from_numpy(np.concatenate(synth.synth_features(256, 300, 30, seed=1234))).cuda()
offs = np.arange(257, dtype=np.int32) * 300
for _ in range(3):
    tr.predict_packed(feats, offs)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = []
for rep in range(5 if steps >= 20 else 1):
    e0.record()
    for _ in range(steps):
        tr.predict_packed(feats, offs)
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / steps)
print("lib %s  ms/forward median %.4f min %.4f" % (os.path.basename(os.path.dirname(_lib.LIB_PATH)), float(np.median(best)), min(best)))
if steps >= 20:
    tr.profile_begin(8192)
    for _ in range(20):
        tr.predict_packed(feats, offs)
    recs, _ = tr.profile_end()
    print(" ".join("%s=%.4f" % (r["name"], r["ms"]) for r in recs))
