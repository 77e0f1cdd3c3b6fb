#!/usr/bin/env python
"""End-to-end ark -> ark throughput of the extraction driver (host parsing + H2D + kernels + D2H + ark write).
usage: python tools/cli_throughput.py [n_utts] [frames]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 300
    from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
    tmp = tempfile.mkdtemp(prefix="xvcli_", dir="/tmp")
    params = dict(synth.TDNN_STAT_PARAMS)
    model_io.save_model(os.path.join(tmp, "exp"), params, 30, synth.synth_weights(params, 30, seed=0), step=1)
    ark = os.path.join(tmp, "feats.ark")
    rs = np.random.RandomState(0)
    base = rs.standard_normal((frames + 64, 30)).astype(np.float32)
    t0 = time.perf_counter()
    with open(ark, "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + frames], key="utt%07d" % i)
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    k = sum(1 for _ in kaldi_io.read_mat_ark(ark))
    t_parse = time.perf_counter() - t0
    out = os.path.join(tmp, "xvector.ark")
    t0 = time.perf_counter()
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])
    t_cli = time.perf_counter() - t0
    m = sum(1 for _ in kaldi_io.read_vec_flt_ark(out))
    print("utts %d (%d read back) | ark write %.2fs | python parse only %.2fs = %.0f utt/s | CLI total %.2fs = %.0f utt/s"
          % (n, m, t_write, t_parse, k / t_parse, t_cli, n / t_cli))


if __name__ == "__main__":
    import logging
    logging.disable(logging.INFO)
    main()
