#!/usr/bin/env python
"""End-to-end ark -> ark throughput of the extraction driver (host parsing + H2D + kernels + D2H + ark write).
usage: python tools/cli_throughput.py [n_utts] [frames] [--varlen]      (--varlen: T ~ U[200,1000], one plan per batch)"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402


def main():
    argv = [a for a in sys.argv[1:] if not a.startswith("--")]
    varlen = "--varlen" in sys.argv
    n = int(argv[0]) if len(argv) > 0 else 20000
    frames = int(argv[1]) if len(argv) > 1 else 300
    from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
    tmp = tempfile.mkdtemp(prefix="xvcli_", dir="/tmp")
    params = dict(synth.TDNN_STAT_PARAMS)
    model_io.save_model(os.path.join(tmp, "exp"), params, 30, synth.synth_weights(params, 30, seed=0), step=1)
    ark = os.path.join(tmp, "feats.ark")
    rs = np.random.RandomState(0)
    lens = rs.randint(200, 1001, size=n) if varlen else np.full(n, frames)
    base = rs.standard_normal((int(lens.max()) + 64, 30)).astype(np.float32)
    t0 = time.perf_counter()
    with open(ark, "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + int(lens[i])], key="utt%07d" % i)
    t_write = time.perf_counter() - t0
    t0 = time.perf_counter()
    k = sum(1 for _ in kaldi_io.read_mat_ark(ark))
    t_parse = time.perf_counter() - t0
    out = os.path.join(tmp, "xvector.ark")
    import io
    import logging
    buf = io.StringIO()
    hdl = logging.StreamHandler(buf)
    logging.disable(logging.NOTSET)
    logging.getLogger("xvec.extract").addHandler(hdl)
    logging.getLogger("xvec.extract").setLevel(logging.WARNING)
    t0 = time.perf_counter()
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])
    t_cli = time.perf_counter() - t0
    # the driver logs its loop time at INFO: run once more in-process (GPU warm, model load repeated) with that line captured
    logging.getLogger("xvec.extract").setLevel(logging.INFO)
    t0 = time.perf_counter()
    extract.main(["--gpu", "0", os.path.join(tmp, "exp"), "ark:" + ark, "ark:" + out])
    t_cli2 = time.perf_counter() - t0
    loop = [l for l in buf.getvalue().splitlines() if l.startswith("Extracted ") or "driver loop" in l]
    loop_s = float(loop[-1].rsplit(" in ", 1)[1].split()[0]) if loop and loop[-1].startswith("Extracted") else None
    m = sum(1 for _ in kaldi_io.read_vec_flt_ark(out))
    print("utts %d (%d read back, %s, %.1f M frames) | ark write %.2fs | python parse only %.2fs = %.0f utt/s | CLI total %.2fs = "
          "%.0f utt/s = %.1f M frames/s (incl. process-level start-up: model upload, first touch of the GPU)"
          % (n, m, "T~U[200,1000]" if varlen else "T=%d" % frames, lens.sum() / 1e6, t_write, t_parse, k / t_parse, t_cli,
             n / t_cli, lens.sum() / t_cli / 1e6))
    if loop_s:
        print("   second run in the same process: total %.2fs; driver loop %.3fs = %.0f utt/s = %.1f M frames/s | %s"
              % (t_cli2, loop_s, n / loop_s, lens.sum() / loop_s / 1e6, (loop[-2].split("driver loop:", 1)[1].strip() if len(loop) > 1 else "")))


if __name__ == "__main__":
    main()
