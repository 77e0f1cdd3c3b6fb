"""ORACLE (test infrastructure only): numpy restatement of the Kaldi feature pipe the reference puts in
front of extract.py (egs/voxceleb/v1/nnet/run_extract_embeddings.sh:47):

    apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 ... | select-voiced-frames ...

Kaldi is an external dependency of the reference ($KALDI_ROOT, not vendored, no pinned version) and is absent
here, so this follows the published algorithm of `SlidingWindowCmnInternal` (kaldi/src/feat/feature-functions.cc)
and of select-voiced-frames.cc.  **parity unpinned**: no Kaldi binary, test or fixture is available to pin it."""
import numpy as np


def sliding_cmn(x, cmn_window=300, center=True, min_window=100):
    """Mean normalisation over a sliding window, arithmetic in double (Kaldi converts to Matrix<double>)."""
    x = np.asarray(x, dtype=np.float64)
    T = x.shape[0]
    out = np.empty_like(x)
    for t in range(T):
        if center:
            ws = t - cmn_window // 2
            we = ws + cmn_window
        else:
            ws = t - cmn_window
            we = t + 1
        if ws < 0:
            we -= ws
            ws = 0
        if not center and we > t:
            we = max(t + 1, min_window)
        if we > T:
            ws -= we - T
            we = T
            if ws < 0:
                ws = 0
        out[t] = x[t] - x[ws:we].mean(axis=0)
    return out.astype(np.float32)


def select_voiced(x, vad):
    """Keep frame t iff vad[t] != 0 (select-voiced-frames.cc)."""
    vad = np.asarray(vad)
    assert vad.shape[0] == x.shape[0]
    return x[vad != 0]
