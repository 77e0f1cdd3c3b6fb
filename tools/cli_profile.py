#!/usr/bin/env python
"""cProfile of the extraction driver's loop on a page-cached ark (where does the host time of the CLI go?).
usage: python tools/cli_profile.py [n_utts]"""
import cProfile
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
    tmp = tempfile.mkdtemp(prefix="xvcli_", dir="/tmp")
    params = dict(synth.TDNN_STAT_PARAMS)
    model_io.save_model(os.path.join(tmp, "exp"), params, 30, synth.synth_weights(params, 30, seed=0), step=1)
    ark = os.path.join(tmp, "feats.ark")
    base = np.random.RandomState(0).standard_normal((364, 30)).astype(np.float32)
    with open(ark, "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + 300], key="utt%07d" % i)
    out = os.path.join(tmp, "xvector.ark")
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])        # warm: GPU, page cache
    pr = cProfile.Profile()
    pr.enable()
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
