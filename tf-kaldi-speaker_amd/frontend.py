"""GPU feature front-end: sliding-window CMN + voiced-frame selection on a packed batch (csrc/frontend.hip),
the counterpart of the Kaldi pipe `apply-cmvn-sliding --norm-vars=false --center=true --cmn-window=300 |
select-voiced-frames` of egs/voxceleb/v1/nnet/run_extract_embeddings.sh:47."""
import ctypes as C

import numpy as np

from . import _lib


def cmn_select_packed(feats_dev, offsets, vads, cmn_window=300, center=True, min_window=100, dim=None, device_index=None,
                      min_frames=0):
    """feats_dev: CUDA float32 [frames, ld] (raw features of B utterances back to back); offsets: B+1 frame
    offsets; vads: list of B per-frame VAD arrays (non-zero = keep; None = utterance without decisions: dropped) or None to
    keep every frame.
    Utterances left with fewer than min_frames frames are dropped (the reference's extract.py:65-67 sees the
    lengths after select-voiced-frames).
    Returns (out CUDA float32 [kept_frames, dim], new offsets int32 [B'+1], indices of the B' kept utterances)."""
    import torch
    lib = _lib.load()
    offsets = np.ascontiguousarray(offsets, dtype=np.int32)
    B = len(offsets) - 1
    dim = int(dim or feats_dev.shape[1])
    keep, counts, kept = [], [], []
    for b in range(B):
        n = int(offsets[b + 1] - offsets[b])
        if vads is None:
            idx = np.arange(offsets[b], offsets[b + 1], dtype=np.int32)
        elif vads[b] is None:                # no VAD decisions for this utterance: dropped, as select-voiced-frames does
            continue
        else:
            v = np.asarray(vads[b])
            if v.shape[0] != n:
                raise ValueError("utterance %d: %d VAD decisions for %d frames" % (b, v.shape[0], n))
            idx = np.flatnonzero(v != 0).astype(np.int32) + offsets[b]
        if idx.shape[0] < max(min_frames, 1):
            continue
        keep.append(idx)
        counts.append(idx.shape[0])
        kept.append(b)
    src = np.concatenate(keep) if keep else np.zeros(0, np.int32)
    new_off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    kept = np.asarray(kept, dtype=np.int64)
    dev = feats_dev.device
    idx = dev.index if device_index is None else device_index
    src_dev = torch.from_numpy(src).to(dev)
    off_dev = torch.from_numpy(offsets).to(dev)
    scratch = torch.empty((int(offsets[-1]) + B) * dim, dtype=torch.float64, device=dev)
    out = torch.empty((int(src.shape[0]), dim), dtype=torch.float32, device=dev)
    if src.shape[0] == 0:
        return out, new_off, kept
    stream = torch.cuda.current_stream(idx).cuda_stream
    rc = lib.xv_frontend_cmn_select(idx, C.c_void_p(feats_dev.data_ptr()), int(feats_dev.shape[1]), dim,
                                    C.c_void_p(off_dev.data_ptr()), B, C.c_void_p(src_dev.data_ptr()), int(src.shape[0]),
                                    int(cmn_window), int(bool(center)), int(min_window),
                                    C.c_void_p(scratch.data_ptr()), C.c_void_p(out.data_ptr()), C.c_void_p(stream))
    _lib.check(rc)
    return out, new_off, kept
