"""Batch ark I/O over the native reader / formatter of libxvec_hip.so (csrc/ark_io.cpp).

`ArkBatchReader` (rspecifiers `ark:file`, `ark:cmd |`, `scp:file.scp`) yields whole device batches -- (keys, frame offsets, [frames, dim] float32 array living
in a caller-visible staging buffer) -- parsed outside the GIL, instead of one (key, matrix) per Python call
as dataset/kaldi_io.py read_mat_ark (:974-994) / read_mat_scp (:953-972) do.  `format_vectors` is the batch form of
write_vec_flt (:915-946); `VectorWriter` is the wspecifier side (`ark:file`, `ark:| cmd`, `ark,scp:a.ark,a.scp` -- the
last is what `ark:| copy-vector ark:- ark,scp:...` does in run_extract_embeddings.sh:70).  gzip'ed arks and text arks
are not handled here: callers fall back to kaldi_io.read_mat_ark."""
import ctypes as C
import os

import numpy as np

from . import _lib
from .kaldi_io import open_or_fd


class ArkBatchReader(object):
    def __init__(self, rspecifier, batch_frames=76800, min_frames=0, max_utts=4096, buffers=None, capacity=None, copy_threads=None):
        """buffers: optional list of >= 2 float32 arrays (e.g. numpy views of pinned torch tensors) that are
        filled round-robin; a batch stays valid until `len(buffers) - 1` further batches have been read."""
        self._lib = _lib.load()
        self._fd = None
        self._r = C.c_void_p()
        spec = rspecifier.strip()
        self.is_scp = spec.startswith("scp:") or spec.startswith("scp,")
        plain = spec.split(":", 1)[1] if spec.startswith(("ark:", "ark,", "scp:", "scp,")) else spec
        if self.is_scp:
            rc = self._lib.xv_ark_open_scp(plain.strip().encode(), C.byref(self._r))
            if rc < 0:
                raise IOError("cannot read script file %s (%s)" % (plain, "ranges / pipes are not supported natively"
                                                                  if rc == -2 else "missing or malformed"))
        elif plain.endswith("|"):
            self._fd = open_or_fd(rspecifier)                       # subprocess stdout (kaldi_io.popen)
            rc = self._lib.xv_ark_open(None, self._fd.fileno(), C.byref(self._r))
        else:
            if plain.endswith(".gz") or plain.endswith(".scp"):
                raise ValueError("native ark reader: %s is not a plain binary ark" % rspecifier)
            rc = self._lib.xv_ark_open(plain.encode(), -1, C.byref(self._r))
        if rc < 0:
            raise IOError("cannot open %s" % rspecifier)
        if copy_threads is not None:
            self._lib.xv_ark_set_copy_threads(self._r, int(copy_threads))
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

    def __len__(self):
        n = int(self._lib.xv_ark_scp_count(self._r))
        if n < 0:
            raise TypeError("only a script-file reader knows its record count")
        return n

    def shapes(self):
        """(rows, cols) int32 arrays of every record of the script file (headers only); rewinds the table."""
        n = len(self)
        rows, cols = np.empty(n, np.int32), np.empty(n, np.int32)
        rc = self._lib.xv_ark_scp_shapes(self._r, rows.ctypes.data_as(C.c_void_p), cols.ctypes.data_as(C.c_void_p), n)
        if rc < 0:
            raise IOError("scp scan failed: %s" % self._lib.xv_ark_error(self._r).decode("utf-8", "replace"))
        return rows, cols

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
            rows, cols = C.c_int(0), C.c_int(0)
            if self._lib.xv_ark_pending_shape(self._r, C.byref(rows), C.byref(cols)) == 0 and \
                    int(rows.value) * int(cols.value) > flat.size:
                # one utterance larger than the staging buffer (the reference chunks arbitrarily long recordings,
                # extract.py:68-86): deliver it alone in a one-off buffer of its own size
                flat = np.empty(int(rows.value) * int(cols.value), np.float32)
                rc = self._lib.xv_ark_next_batch(self._r, self.batch_frames, 1, self.min_frames,
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


def read_scp_table(path):
    """[(key, rxfilename)] of a Kaldi script file (text)."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if line:
                key, rx = line.split(None, 1)
                out.append((key, rx.strip()))
    return out


def scp_lengths(scp_path, utt2num_frames=None):
    """Frame count of every utterance of feats.scp, in table order: from Kaldi's `utt2num_frames` when the data
    directory has one, else from the matrix headers (one seek + short read each)."""
    if utt2num_frames and os.path.isfile(utt2num_frames):
        table = {}
        with open(utt2num_frames) as f:
            for line in f:
                p = line.split()
                if len(p) == 2:
                    table[p[0]] = int(p[1])
        keys = [k for k, _ in read_scp_table(scp_path)]
        if all(k in table for k in keys):
            return np.array([table[k] for k in keys], dtype=np.int64)
    r = ArkBatchReader("scp:" + scp_path, buffers=[np.empty(1, np.float32)])
    try:
        rows, _ = r.shapes()
    finally:
        r.close()
    return rows.astype(np.int64)


class VectorWriter(object):
    """Batch writer of float-vector tables.  wspecifier: `ark:file`, `ark:| cmd`, a bare file name, or
    `ark,scp:file.ark,file.scp` (binary ark + script file whose offsets point behind "key ", as Kaldi writes them)."""

    def __init__(self, wspecifier):
        spec = wspecifier.strip()
        self._scp = None
        self._ark_path = None
        self._pos = 0
        head = spec.split(":", 1)[0] if ":" in spec else ""
        opts = head.split(",")
        if opts[0] == "ark" and "scp" in opts[1:]:
            ark_path, scp_path = [x.strip() for x in spec.split(":", 1)[1].split(",", 1)]
            self._ark_path = ark_path
            self._fd = open(ark_path, "wb")
            self._scp = open(scp_path, "w")
        else:
            self._fd = open_or_fd(wspecifier, "wb")

    def write(self, keys, vectors):
        v = np.ascontiguousarray(vectors, dtype=np.float32)
        if v.ndim != 2 or v.shape[0] != len(keys):
            raise ValueError("VectorWriter.write: %d keys for an array of shape %s" % (len(keys), v.shape))
        if not len(keys):
            return
        self._fd.write(format_vectors(keys, v))
        if self._scp is not None:
            klen = np.fromiter((len(k.encode("latin1")) for k in keys), dtype=np.int64, count=len(keys))
            rec = klen + 1 + 10 + 4 * v.shape[1]                 # "key " + "\0BFV \4<i32>" + payload
            start = self._pos + np.concatenate([[0], np.cumsum(rec)[:-1]]) + klen + 1
            self._scp.write("".join("%s %s:%d\n" % (k, self._ark_path, o) for k, o in zip(keys, start)))
            self._pos += int(rec.sum())

    def close(self):
        """Close the table; returns the exit code of an output pipe's command (0 for files)."""
        rc = 0
        if self._fd is not None:
            self._fd.close()
            proc = getattr(self._fd, "_xv_proc", None)
            if proc is not None:
                rc = proc.wait()
            self._fd = None
        if self._scp is not None:
            self._scp.close()
            self._scp = None
        return rc
