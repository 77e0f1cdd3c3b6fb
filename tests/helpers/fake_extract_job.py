"""Test stand-in for `python -m tf_kaldi_speaker_amd.extract` (same command line) that needs no GPU: the
"embedding" of an utterance is [mean | std + 0.01 T | #frames | device] of its (VAD-selected) frames.  Used by
tests/test_launcher.py to exercise the launcher's fan-out / logs / ordered concatenation on CPU."""
import sys

import numpy as np

from tf_kaldi_speaker_amd import extract, kaldi_io, native_ark


def embed(feat, device):
    return np.concatenate([feat.mean(0), feat.std(0) + 0.01 * feat.shape[0], [feat.shape[0], device]]).astype(np.float32)


def main(argv=None):
    args = extract.build_parser().parse_args(argv)
    assert args.scp_input and args.rspecifier.startswith("scp:")
    device = args.gpu if args.gpu >= 0 else extract.auto_device(args.rspecifier, args.wspecifier, device_count=2)
    print("fake job on device %d: %s -> %s" % (device, args.rspecifier, args.wspecifier))
    vads = None
    if args.vad_rspecifier:
        vads = dict(extract._vad_records(args.vad_rspecifier))
    w = native_ark.VectorWriter(args.wspecifier)
    for key, feat in kaldi_io.read_mat_scp(args.rspecifier.split(":", 1)[1]):
        if key.startswith("FAIL"):
            print("injected failure at %s" % key)
            return 3
        if vads is not None:
            feat = feat[vads[key] != 0]
        if feat.shape[0] < args.min_chunk_size:
            continue
        w.write([key], embed(feat, device)[None, :])
    return w.close()


if __name__ == "__main__":
    sys.exit(main())
