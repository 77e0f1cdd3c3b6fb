#!/usr/bin/env python
"""cProfile of the extraction driver's loop on a page-cached ark (where does the host time of the CLI go?).
usage: python tools/cli_profile.py [n_utts] [--varlen]"""
import cProfile
import os
import pstats
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    varlen = "--varlen" in sys.argv
    n = int(argv[0]) if argv else 100000
    from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
    tmp = tempfile.mkdtemp(prefix="xvcli_", dir="/tmp")
    params = dict(synth.TDNN_STAT_PARAMS)
    model_io.save_model(os.path.join(tmp, "exp"), params, 30, synth.synth_weights(params, 30, seed=0), step=1)
    ark = os.path.join(tmp, "feats.ark")
    rs = np.random.RandomState(0)
    lens = rs.randint(200, 1001, size=n) if varlen else np.full(n, 300)
    base = rs.standard_normal((1064, 30)).astype(np.float32)
    with open(ark, "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + int(lens[i])], key="utt%07d" % i)
    out = os.path.join(tmp, "xvector.ark")
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])        # warm: GPU, page cache
    pr = cProfile.Profile()
    pr.enable()
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])
    pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(18)


if __name__ == "__main__":
    main()
