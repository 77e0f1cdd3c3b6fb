"""Batch ark I/O over the native reader / formatter of libxvec_hip.so (csrc/ark_io.cpp).

`ArkBatchReader` yields whole device batches -- (keys, frame offsets, [frames, dim] float32 array living
in a caller-visible staging buffer) -- parsed outside the GIL, instead of one (key, matrix) per Python call
as dataset/kaldi_io.py read_mat_ark (:974-994) does.  `format_vectors` is the batch form of write_vec_flt
(:915-946).  gzip'ed arks and text arks are not handled here: callers fall back to kaldi_io.read_mat_ark."""
import ctypes as C

import numpy as np

from . import _lib
from .kaldi_io import open_or_fd


class ArkBatchReader(object):
    def __init__(self, rspecifier, batch_frames=76800, min_frames=0, max_utts=4096, buffers=None, capacity=None):
        """buffers: optional list of >= 2 float32 arrays (e.g. numpy views of pinned torch tensors) that are
        filled round-robin; a batch stays valid until `len(buffers) - 1` further batches have been read."""
        self._lib = _lib.load()
        self._fd = None
        self._r = C.c_void_p()
        spec = rspecifier.strip()
        plain = spec.split(":", 1)[1] if spec.startswith("ark:") or spec.startswith("ark,") else spec
        if plain.endswith("|"):
            self._fd = open_or_fd(rspecifier)                       # subprocess stdout (kaldi_io.popen)
            rc = self._lib.xv_ark_open(None, self._fd.fileno(), C.byref(self._r))
        else:
            if plain.endswith(".gz") or plain.endswith(".scp"):
                raise ValueError("native ark reader: %s is not a plain binary ark" % rspecifier)
            rc = self._lib.xv_ark_open(plain.encode(), -1, C.byref(self._r))
        if rc < 0:
            raise IOError("cannot open %s" % rspecifier)
        self.batch_frames = int(batch_frames)
        self.min_frames = int(min_frames)
        self.max_utts = int(max_utts)
        cap = int(capacity or (self.batch_frames + 65536) * 64)
        self._bufs = buffers if buffers is not None else [np.empty(cap, np.float32), np.empty(cap, np.float32)]
        self._turn = 0
        self._offsets = np.empty(self.max_utts + 1, np.int32)
        self._keys = C.create_string_buffer(self.max_utts * 64)

    @property
    def skipped(self):
        return int(self._lib.xv_ark_skipped(self._r))

    def next_batch(self):
        """-> (keys list, offsets int32 [n+1], features [frames, dim]) or None at end of stream."""
        buf = self._bufs[self._turn % len(self._bufs)]
        self._turn += 1
        flat = buf.reshape(-1)
        n, dim = C.c_int(0), C.c_int(0)
        rc = self._lib.xv_ark_next_batch(self._r, self.batch_frames, self.max_utts, self.min_frames,
                                         flat.ctypes.data_as(C.c_void_p), flat.size,
                                         self._offsets.ctypes.data_as(C.c_void_p), self._keys, len(self._keys),
                                         C.byref(n), C.byref(dim))
        if rc < 0:
            raise IOError("ark read failed: %s" % self._lib.xv_ark_error(self._r).decode("utf-8", "replace"))
        if rc == 0:
            return None
        nk = n.value
        raw = C.string_at(self._keys, _keys_len(self._keys, nk))
        keys = raw.decode("latin1").split("\n")[:nk]
        offsets = self._offsets[:nk + 1].copy()
        feats = flat[:int(offsets[-1]) * dim.value].reshape(int(offsets[-1]), dim.value)
        return keys, offsets, feats

    def __iter__(self):
        while True:
            b = self.next_batch()
            if b is None:
                return
            yield b

    def close(self):
        if self._r:
            self._lib.xv_ark_close(self._r)
            self._r = C.c_void_p()
        if self._fd is not None:
            self._fd.close()
            self._fd = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _keys_len(buf, n):
    """Length in bytes of the first n newline-terminated keys in the ctypes buffer."""
    data = buf.raw
    pos = 0
    for _ in range(n):
        pos = data.index(b"\n", pos) + 1
    return pos


_fmt_buf = None


def format_vectors(keys, vectors):
    """Binary Kaldi vector-ark bytes of `vectors` ([n, dim] float32) with `keys` (byte-identical to n calls
    of write_vec_flt, dataset/kaldi_io.py:915-946)."""
    global _fmt_buf
    lib = _lib.load()
    v = np.ascontiguousarray(vectors, dtype=np.float32)
    n, dim = v.shape
    kb = ("\n".join(keys) + "\n").encode("latin1") if n else b""
    need = len(kb) + n * (11 + 4 * dim) + 16
    if _fmt_buf is None or len(_fmt_buf) < need:
        _fmt_buf = C.create_string_buffer(max(need, 1 << 20))
    got = lib.xv_ark_format_vectors(kb, n, v.ctypes.data_as(C.c_void_p), dim, dim, _fmt_buf, len(_fmt_buf))
    if got < 0:
        raise IOError("xv_ark_format_vectors failed (%d)" % got)
    return C.string_at(_fmt_buf, got)
