"""Kaldi table I/O for the extraction path: matrix-ark in, vector-ark out.

Keeps the function surface of the reference's dataset/kaldi_io.py that extract.py uses
(`open_or_fd` :625, `read_key` :694, `read_mat_ark` :974, `read_mat` :996,
`write_vec_flt` :915, `write_mat` :1175, `read_vec_flt_ark` :838, `read_vec_flt` :855)
and the same wire format, so Kaldi binaries on either side of the pipe are unchanged:

  matrix record : key SP \\0 B  'FM '|'DM '  \\4 <i32 rows> \\4 <i32 cols>  row-major payload
                  key SP \\0 B  'CM '  <f32 min><f32 range><i32 rows><i32 cols>
                                       cols x 4 x u16 percentiles, then col-major u8   (:1071-1115)
  vector record : key SP \\0 B  'FV '|'DV '  \\4 <i32 dim>  payload

The implementation is its own: a chunk-buffered record parser (the reference reads the
key one byte per `fd.read(1)` call) and a vectorised compressed-matrix decoder.
"""
import gzip
import io
import re
import struct
import subprocess
import threading

import numpy as np


class UnsupportedDataType(Exception):
    pass


class UnknownVectorHeader(Exception):
    pass


class UnknownMatrixHeader(Exception):
    pass


class BadInputFormat(Exception):
    pass


class SubprocessFailed(Exception):
    pass


_SPEC_PREFIX = re.compile(r'^(ark|scp)(,scp|,b|,t|,n?f|,n?p|,b?o|,n?s|,n?cs)*:')


def popen(cmd, mode="rb"):
    """Run `cmd` through the shell and return its stdout ('rb'/'r') or stdin ('wb'/'w').
    A watcher thread raises SubprocessFailed if the command exits non-zero
    (reference behaviour, dataset/kaldi_io.py:658-688)."""
    if not isinstance(cmd, str):
        raise TypeError("invalid cmd type (%s, expected string)" % type(cmd))
    if mode not in ("r", "w", "rb", "wb"):
        raise ValueError("invalid mode %s" % mode)
    reading = mode[0] == "r"
    proc = subprocess.Popen(cmd, shell=True,
                            stdout=subprocess.PIPE if reading else None,
                            stdin=None if reading else subprocess.PIPE)

    def _watch():
        ret = proc.wait()
        if ret > 0:
            raise SubprocessFailed('cmd %s returned %d !' % (cmd, ret))

    threading.Thread(target=_watch, daemon=True).start()
    pipe = proc.stdout if reading else proc.stdin
    pipe._xv_proc = proc                      # lets callers wait for a writer pipe to drain
    return io.TextIOWrapper(pipe) if len(mode) == 1 else pipe


def open_or_fd(file, mode='rb'):
    """Open a file, gzipped file, `cmd |` input pipe, `| cmd` output pipe, or pass an
    already opened descriptor through.  An optional `ark:`/`scp:` prefix is stripped and a
    trailing `:offset` seeks (dataset/kaldi_io.py:625-655)."""
    if not isinstance(file, str):
        return file
    offset = None
    if _SPEC_PREFIX.search(file):
        file = file.split(':', 1)[1]
    if re.search(r':[0-9]+$', file):
        file, offset = file.rsplit(':', 1)
    file = file.strip() if (file.strip().endswith('|') or file.strip().startswith('|')) else file
    if file[-1] == '|':
        fd = popen(file[:-1], 'rb')
    elif file[0] == '|':
        fd = popen(file[1:], 'wb')
    elif file.split('.')[-1] == 'gz':
        fd = gzip.open(file, mode)
    else:
        fd = open(file, mode)
    if offset is not None:
        fd.seek(int(offset))
    return fd


class _Stream(object):
    """Chunk-buffered reader over a binary descriptor (file, gzip or pipe)."""

    def __init__(self, fd, chunk=1 << 20):
        # chunk=1: consume exactly what each record needs (a caller-owned descriptor may be
        # read again by the caller afterwards); the ark generators use 1 MiB read-ahead.
        self.fd = fd
        self.buf = b""
        self.pos = 0
        self.CHUNK = chunk

    def _fill(self, n):
        """Ensure n bytes are available from pos (fewer only at EOF)."""
        have = len(self.buf) - self.pos
        if have >= n:
            return
        parts = [self.buf[self.pos:]]
        while have < n:
            chunk = self.fd.read(max(self.CHUNK, n - have) if self.CHUNK > 1 else n - have)
            if not chunk:
                break
            parts.append(chunk)
            have += len(chunk)
        self.buf = b"".join(parts)
        self.pos = 0

    def read(self, n):
        self._fill(n)
        out = self.buf[self.pos:self.pos + n]
        self.pos += len(out)
        return out

    def read_exact(self, n):
        out = self.read(n)
        if len(out) != n:
            raise BadInputFormat("unexpected end of stream (wanted %d bytes, got %d)" % (n, len(out)))
        return out

    def read_token(self):
        """Bytes up to (not including) the next space; the space is consumed.  Returns b''
        at end of stream."""
        while True:
            i = self.buf.find(b" ", self.pos)
            if i >= 0:
                tok = self.buf[self.pos:i]
                self.pos = i + 1
                return tok
            before = len(self.buf) - self.pos
            self._fill(before + self.CHUNK)
            if len(self.buf) - self.pos == before:       # EOF without a space
                tok = self.buf[self.pos:]
                self.pos = len(self.buf)
                return tok

    def readline(self):
        while True:
            i = self.buf.find(b"\n", self.pos)
            if i >= 0:
                line = self.buf[self.pos:i + 1]
                self.pos = i + 1
                return line
            before = len(self.buf) - self.pos
            self._fill(before + self.CHUNK)
            if len(self.buf) - self.pos == before:
                line = self.buf[self.pos:]
                self.pos = len(self.buf)
                return line


def _as_stream(fd):
    return fd if isinstance(fd, _Stream) else _Stream(fd, chunk=1)


def read_key(fd):
    """Utterance key of the next record, or None at end of stream (dataset/kaldi_io.py:694-707)."""
    s = _as_stream(fd)
    key = s.read_token().decode("latin1").strip()
    if key == '':
        return None
    if re.match(r'^\S+$', key) is None:
        raise BadInputFormat("key contains whitespace: %r" % key)
    return key


# ----------------------------------------------------------------------------- vectors
def _read_vec_binary(s):
    header = s.read_exact(3)
    if header == b'FV ':
        dtype = np.dtype('<f4')
    elif header == b'DV ':
        dtype = np.dtype('<f8')
    else:
        raise UnknownVectorHeader("The header contained '%s'" % header.decode("latin1"))
    if s.read_exact(1) != b'\x04':
        raise BadInputFormat("vector: int-size marker missing")
    dim = struct.unpack('<i', s.read_exact(4))[0]
    return np.frombuffer(s.read_exact(dim * dtype.itemsize), dtype=dtype).copy()


def read_vec_flt(file_or_fd):
    """One Kaldi float vector, binary or ascii (dataset/kaldi_io.py:855-885)."""
    fd = open_or_fd(file_or_fd)
    s = _as_stream(fd)
    flag = s.read(2)
    if flag == b'\0B':
        ans = _read_vec_binary(s)
    else:
        arr = (flag + s.readline()).decode().strip().split()
        arr = [a for a in arr if a not in ('[', ']')]
        ans = np.array(arr, dtype=float)
    if fd is not file_or_fd and not isinstance(file_or_fd, _Stream):
        fd.close()
    return ans


def read_vec_flt_ark(file_or_fd):
    """Generator of (key, vector) over a vector ark (dataset/kaldi_io.py:838-853)."""
    fd = open_or_fd(file_or_fd)
    s = _Stream(fd)
    try:
        key = read_key(s)
        while key:
            yield key, read_vec_flt(s)
            key = read_key(s)
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_vec_flt(file_or_fd, v, key=''):
    """Write one binary Kaldi vector record (dataset/kaldi_io.py:915-946)."""
    fd = open_or_fd(file_or_fd, mode='wb')
    try:
        v = np.asarray(v)
        if v.dtype == np.float32:
            tag = b'FV '
        elif v.dtype == np.float64:
            tag = b'DV '
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % v.dtype)
        rec = (key + ' ').encode("latin1") if key != '' else b''
        rec += b'\0B' + tag + b'\x04' + struct.pack('<I', v.shape[0])
        fd.write(rec + np.ascontiguousarray(v).astype(v.dtype.newbyteorder('<'), copy=False).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()


# ----------------------------------------------------------------------------- matrices
_U16 = 1.52590218966964e-05      # 1/65535, the constant the reference uses (:1083)


def _decode_cm1(s):
    """'CM ' compressed matrix (per-column percentile headers + column-major u8): the
    piecewise-linear decode of dataset/kaldi_io.py:1071-1115, in one vectorised pass.

    Arithmetic is done in float64 and rounded to float32 once, which is what Kaldi's
    CompressedMatrix does (float * double constant) and what the reference's expressions did
    under the NumPy 1.x it was written for (np.float32 scalar * python float -> float64).
    Under NumPy >= 2 (NEP 50) the same reference source keeps float32 intermediates and can
    differ from this by one float32 ulp; tests/test_kaldi_io.py allows exactly that."""
    gmin, grange, rows, cols = struct.unpack('<ffii', s.read_exact(16))
    hdr = np.frombuffer(s.read_exact(cols * 8), dtype='<u2').reshape(cols, 4)
    data = np.frombuffer(s.read_exact(cols * rows), dtype=np.uint8).reshape(cols, rows)
    p = (np.float64(gmin) + np.float64(grange) * _U16 * hdr.astype(np.float64)).astype(np.float32).astype(np.float64)
    p0, p25, p75, p100 = (p[:, i:i + 1] for i in range(4))
    d = data.astype(np.float64)
    lo = p0 + (p25 - p0) / 64. * d
    mid = p25 + (p75 - p25) / 128. * (d - 64.)
    hi = p75 + (p100 - p75) / 63. * (d - 192.)
    out = np.where(data <= 64, lo, np.where(data <= 192, mid, hi)).astype(np.float32)
    return np.ascontiguousarray(out.T)


def _decode_cm23(s, two_byte):
    """'CM2' (u16) / 'CM3' (u8) globally-scaled formats of Kaldi's CompressedMatrix.  The
    reference rejects them (assert at :1076); accepted here, documented as an extension."""
    gmin, grange, rows, cols = struct.unpack('<ffii', s.read_exact(16))
    if two_byte:
        raw = np.frombuffer(s.read_exact(rows * cols * 2), dtype='<u2')
        inc = np.float32(grange) * np.float32(1.0 / 65535.0)
    else:
        raw = np.frombuffer(s.read_exact(rows * cols), dtype=np.uint8)
        inc = np.float32(grange) * np.float32(1.0 / 255.0)
    return (np.float32(gmin) + inc * raw.astype(np.float32)).reshape(rows, cols)


def _read_mat_binary(s):
    header = s.read_exact(3)
    if header == b'CM ':
        return _decode_cm1(s)
    if header == b'CM2':
        s.read_exact(1)
        return _decode_cm23(s, True)
    if header == b'CM3':
        s.read_exact(1)
        return _decode_cm23(s, False)
    if header == b'FM ':
        dtype = np.dtype('<f4')
    elif header == b'DM ':
        dtype = np.dtype('<f8')
    else:
        raise UnknownMatrixHeader("The header contained '%s'" % header.decode("latin1"))
    s1, rows, s2, cols = struct.unpack('<bibi', s.read_exact(10))
    if s1 != 4 or s2 != 4:
        raise BadInputFormat("matrix: int-size markers are %d,%d" % (s1, s2))
    vec = np.frombuffer(s.read_exact(rows * cols * dtype.itemsize), dtype=dtype)
    return vec.reshape(rows, cols).copy()


def _read_mat_ascii(s):
    rows = []
    while True:
        line = s.readline().decode()
        if len(line) == 0:
            raise BadInputFormat("ascii matrix: end of stream before ']'")
        arr = line.strip().split()
        if not arr:
            continue
        if arr[-1] != ']':
            rows.append(np.array(arr, dtype='float32'))
        else:
            rows.append(np.array(arr[:-1], dtype='float32'))
            return np.vstack(rows)


def read_mat(file_or_fd):
    """One Kaldi matrix, binary ('FM ', 'DM ', 'CM ') or ascii (dataset/kaldi_io.py:996-1011)."""
    fd = open_or_fd(file_or_fd)
    s = _as_stream(fd)
    try:
        flag = s.read_exact(2)
        if flag == b'\0B':
            mat = _read_mat_binary(s)
        elif flag == b' [':
            mat = _read_mat_ascii(s)
        else:
            raise BadInputFormat("matrix: expected '\\0B' or ' [', got %r" % flag)
    finally:
        if fd is not file_or_fd and not isinstance(file_or_fd, _Stream):
            fd.close()
    return mat


def read_mat_ark(file_or_fd):
    """Generator of (key, matrix) over a matrix ark file / pipe (dataset/kaldi_io.py:974-994)."""
    fd = open_or_fd(file_or_fd)
    s = _Stream(fd)
    try:
        key = read_key(s)
        while key:
            yield key, read_mat(s)
            key = read_key(s)
    finally:
        if fd is not file_or_fd:
            fd.close()


def read_mat_scp(file_or_fd):
    """Generator of (key, matrix) over a Kaldi scp: lines `key rxfile`, rxfile = `file` or `file:offset`
    (dataset/kaldi_io.py:953-972)."""
    fd = open_or_fd(file_or_fd)
    try:
        for line in fd:
            line = line.decode() if isinstance(line, bytes) else line
            if not line.strip():
                continue
            key, rxfile = line.strip().split(' ', 1)
            yield key, read_mat(rxfile.strip())
    finally:
        if fd is not file_or_fd:
            fd.close()


def write_mat(file_or_fd, m, key=''):
    """Write one binary Kaldi matrix record (dataset/kaldi_io.py:1175-1208)."""
    fd = open_or_fd(file_or_fd, mode='wb')
    try:
        m = np.asarray(m)
        if m.dtype == np.float32:
            tag = b'FM '
        elif m.dtype == np.float64:
            tag = b'DM '
        else:
            raise UnsupportedDataType("'%s', please use 'float32' or 'float64'" % m.dtype)
        rec = (key + ' ').encode("latin1") if key != '' else b''
        rec += b'\0B' + tag + b'\x04' + struct.pack('<I', m.shape[0]) + b'\x04' + struct.pack('<I', m.shape[1])
        fd.write(rec + np.ascontiguousarray(m).tobytes())
    finally:
        if fd is not file_or_fd:
            fd.close()
