"""TensorFlow-free reader (and minimal writer) for TF checkpoint-V2 "tensor bundles".

The reference restores `<model>/nnet/model-<step>` with `tf.train.Saver.restore`
(model/trainer.py:277-295).  A V2 checkpoint is
    model-<step>.index                 an SSTable (LevelDB table format, TF's lib/io/table) mapping
                                       "" -> BundleHeaderProto and  variable name -> BundleEntryProto
    model-<step>.data-0000i-of-0000N   the raw little-endian tensor bytes, entry = (shard, offset, size)
This module parses both with nothing but numpy, so released models can be loaded on a machine
without TensorFlow.

STATUS: **parity unpinned** -- no real checkpoint ships with the reference (README.md:101-119 are
remote links) and TensorFlow is not installed here, so the parser is written from the published
format (tensor_bundle.proto, table_builder.cc / format.cc) and tested only against the writer below
(round trip, tests/test_tf_checkpoint.py; one index assembled byte by byte in the test itself).  Block compression:
none (what BundleWriter emits); snappy blocks raise.  Checksums ARE verified, as TensorFlow's BundleReader does: the masked
CRC-32C behind every table block (LevelDB format.cc: crc32c(block | type byte), Mask = rotr15 + 0xa282ead8) and the masked
CRC-32C of every tensor's bytes (BundleEntryProto.crc32c); a mismatch raises CheckpointFormatError("checksum").  Known-answer
vectors of the checksum itself (RFC 3720 B.4) are in the tests.
"""
import os
import struct

import numpy as np

_TABLE_MAGIC = 0xdb4775248b80fb57
_FOOTER_LEN = 48

# tensorflow/core/framework/types.proto
_DTYPES = {1: np.dtype("<f4"), 2: np.dtype("<f8"), 3: np.dtype("<i4"), 4: np.dtype("u1"), 5: np.dtype("<i2"),
           6: np.dtype("i1"), 9: np.dtype("<i8"), 10: np.dtype("bool"), 17: np.dtype("<u2"), 19: np.dtype("<f2"),
           22: np.dtype("<u4"), 23: np.dtype("<u8")}
_DT_CODE = {np.dtype("float32"): 1, np.dtype("float64"): 2, np.dtype("int32"): 3, np.dtype("int64"): 9}


class CheckpointFormatError(Exception):
    pass


# ------------------------------------------------------------------------------ CRC-32C
_CRC_TABLE = None
_native_crc = None


def _crc32c_py(crc, data):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in bytes(data):
        c = (c >> 8) ^ tab[(c ^ b) & 0xFF]
    return c ^ 0xFFFFFFFF


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli) of `data` (bytes-like), continuing from `crc`.  Uses the native routine of libxvec_hip.so
    (xv_crc32c: the SSE4.2 instruction) when the library is built -- a released model has ~20 MB of tensors -- and a
    table loop otherwise."""
    global _native_crc
    if _native_crc is None:
        try:
            import ctypes
            from . import _lib
            fn = _lib.load().xv_crc32c
            _native_crc = lambda c, d: int(fn(c, ctypes.c_char_p(d) if isinstance(d, bytes) else
                                              ctypes.cast((ctypes.c_char * len(d)).from_buffer_copy(d), ctypes.c_void_p), len(d)))
        except Exception:
            _native_crc = False
    data = bytes(data) if not isinstance(data, bytes) else data
    if _native_crc:
        return _native_crc(crc, data)
    return _crc32c_py(crc, data)


def mask_crc(crc):
    """LevelDB / TensorFlow crc32c::Mask: rotate right by 15 bits and add a constant (a CRC of data that contains CRCs)."""
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + 0xa282ead8) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - 0xa282ead8) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------------------ varint / proto helpers
def _varint(buf, pos):
    result = shift = 0
    while True:
        if pos >= len(buf):
            raise CheckpointFormatError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7f) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise CheckpointFormatError("varint too long")


def _enc_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7f
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yield (field number, wire type, value) of one protobuf message (value: int or bytes)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        field, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise CheckpointFormatError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_shape(buf):
    """TensorShapeProto: repeated Dim dim = 2 { int64 size = 1; }, bool unknown_rank = 3."""
    dims = []
    for field, _, v in _proto_fields(buf):
        if field == 2:
            size = 0
            for f2, _, v2 in _proto_fields(v):
                if f2 == 1:
                    size = _signed64(v2)
            dims.append(size)
    return dims


def _parse_entry(buf):
    """BundleEntryProto: dtype=1, shape=2, shard_id=3, offset=4, size=5, crc32c=6 (fixed32), slices=7."""
    e = {"dtype": 0, "shape": [], "shard_id": 0, "offset": 0, "size": 0, "slices": 0, "crc32c": None}
    for field, _, v in _proto_fields(buf):
        if field == 1:
            e["dtype"] = v
        elif field == 2:
            e["shape"] = _parse_shape(v)
        elif field == 3:
            e["shard_id"] = v
        elif field == 4:
            e["offset"] = v
        elif field == 5:
            e["size"] = v
        elif field == 6:
            e["crc32c"] = struct.unpack("<I", bytes(v))[0]
        elif field == 7:
            e["slices"] += 1
    return e


# ------------------------------------------------------------------------------ SSTable
def _block_handle(buf, pos):
    off, pos = _varint(buf, pos)
    size, pos = _varint(buf, pos)
    return off, size, pos


def _read_block(data, off, size):
    """A table block: entries, restart array, num_restarts; followed on disk by a 1-byte
    compression type and a 4-byte crc (not part of `size`)."""
    if off + size + 5 > len(data):
        raise CheckpointFormatError("block beyond end of index file")
    ctype = data[off + size]
    if ctype != 0:
        raise CheckpointFormatError("compressed table block (type %d): only uncompressed bundles are supported" % ctype)
    stored = struct.unpack("<I", data[off + size + 1:off + size + 5])[0]
    if unmask_crc(stored) != crc32c(data[off:off + size + 1]):            # block contents + the type byte (format.cc, ReadBlock)
        raise CheckpointFormatError("table block at offset %d: checksum mismatch (corrupt index file)" % off)
    block = data[off:off + size]
    if size < 4:
        raise CheckpointFormatError("block too small")
    num_restarts = struct.unpack("<I", block[-4:])[0]
    limit = size - 4 - 4 * num_restarts
    if limit < 0:
        raise CheckpointFormatError("bad restart array")
    pos, key = 0, b""
    while pos < limit:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_index(index_path):
    """-> (header dict, {name: entry dict}) of a `.index` file."""
    with open(index_path, "rb") as f:
        data = f.read()
    if len(data) < _FOOTER_LEN:
        raise CheckpointFormatError("%s is too small to be a checkpoint index" % index_path)
    footer = data[-_FOOTER_LEN:]
    if struct.unpack("<Q", footer[-8:])[0] != _TABLE_MAGIC:
        raise CheckpointFormatError("%s: bad table magic (not a TF checkpoint-V2 index)" % index_path)
    _, _, pos = _block_handle(footer, 0)              # metaindex handle (unused)
    ioff, isize, _ = _block_handle(footer, pos)       # index block handle
    header, entries = {}, {}
    for _, handle in _read_block(data, ioff, isize):
        boff, bsize, _ = _block_handle(handle, 0)
        for key, value in _read_block(data, boff, bsize):
            if key == b"":
                for field, _, v in _proto_fields(value):     # BundleHeaderProto: num_shards=1, endianness=2
                    if field == 1:
                        header["num_shards"] = v
                    elif field == 2:
                        header["endianness"] = v
            else:
                entries[key.decode("utf-8")] = _parse_entry(value)
    if header.get("endianness", 0) != 0:
        raise CheckpointFormatError("big-endian bundle")
    return header, entries


def read_bundle(prefix, name_filter=None):
    """Load every (numeric, unsliced) variable of the checkpoint `prefix` (`prefix.index` +
    `prefix.data-*`) -> {name: ndarray}.  `name_filter(name) -> bool` selects variables."""
    header, entries = read_index(prefix + ".index")
    shards = max(int(header.get("num_shards", 1)), 1)
    out, files = {}, {}
    try:
        for name, e in sorted(entries.items()):
            if name_filter is not None and not name_filter(name):
                continue
            if e["slices"] or e["dtype"] not in _DTYPES:
                continue                                   # partitioned variables / strings: not needed here
            sid = e["shard_id"]
            if sid not in files:
                files[sid] = open("%s.data-%05d-of-%05d" % (prefix, sid, shards), "rb")
            fh = files[sid]
            fh.seek(e["offset"])
            raw = fh.read(e["size"])
            dt = _DTYPES[e["dtype"]]
            n = int(np.prod(e["shape"])) if e["shape"] else 1
            if len(raw) != e["size"] or n * dt.itemsize != e["size"]:
                raise CheckpointFormatError("variable %s: size %d does not match shape %s" % (name, e["size"], e["shape"]))
            if e["crc32c"] is not None and unmask_crc(e["crc32c"]) != crc32c(raw):       # tensor_bundle.cc, BundleReader::GetValue
                raise CheckpointFormatError("variable %s: checksum mismatch (corrupt data shard %d)" % (name, sid))
            out[name] = np.frombuffer(raw, dtype=dt).reshape(e["shape"]).copy()
    finally:
        for fh in files.values():
            fh.close()
    return out


# ------------------------------------------------------------------------------ minimal writer (tests / conversion)
def _enc_field(field, wt, payload):
    return _enc_varint((field << 3) | wt) + payload


def _enc_entry(arr, offset, shard=0):
    shape = b"".join(_enc_field(2, 2, _enc_varint(len(d)) + d)
                     for d in (_enc_field(1, 0, _enc_varint(int(s))) for s in arr.shape))
    msg = _enc_field(1, 0, _enc_varint(_DT_CODE[arr.dtype]))
    msg += _enc_field(2, 2, _enc_varint(len(shape)) + shape)
    if shard:
        msg += _enc_field(3, 0, _enc_varint(shard))
    if offset:
        msg += _enc_field(4, 0, _enc_varint(offset))
    msg += _enc_field(5, 0, _enc_varint(arr.nbytes))
    msg += _enc_field(6, 5, struct.pack("<I", mask_crc(crc32c(arr.tobytes()))))
    return msg


def _build_block(items, restart_interval=16):
    out, restarts, last = bytearray(), [], b""
    for i, (key, value) in enumerate(items):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(key), len(last)) and key[shared] == last[shared]:
                shared += 1
        out += _enc_varint(shared) + _enc_varint(len(key) - shared) + _enc_varint(len(value)) + key[shared:] + value
        last = key
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack("<I", r)
    out += struct.pack("<I", len(restarts))
    return bytes(out)


def _block_trailer(block):
    """type byte (0 = uncompressed) + masked CRC-32C of block | type (table_builder.cc, WriteRawBlock)"""
    return b"\x00" + struct.pack("<I", mask_crc(crc32c(block + b"\x00")))


def write_bundle(prefix, tensors, block_entries=8, shards=1, restart_interval=16):
    """Write {name: ndarray} as a V2 bundle of `shards` data files in the layout read_bundle expects (uncompressed
    blocks, prefix-compressed keys with a restart point every `restart_interval` entries, block and tensor checksums).
    For tests and for converting an .npz back; NOT verified against TensorFlow's own reader."""
    names = sorted(tensors)
    items = [(b"", _enc_field(1, 0, _enc_varint(shards)) + _enc_field(2, 0, _enc_varint(0)))]
    files = [open("%s.data-%05d-of-%05d" % (prefix, i, shards), "wb") for i in range(shards)]
    offsets = [0] * shards
    try:
        for i, n in enumerate(names):
            a = np.asarray(tensors[n])                 # (ascontiguousarray would promote 0-d to 1-d)
            if a.dtype not in _DT_CODE:
                a = a.astype(np.float32)
            sid = i % shards
            items.append((n.encode("utf-8"), _enc_entry(a, offsets[sid], sid)))
            files[sid].write(a.tobytes())
            offsets[sid] += a.nbytes
    finally:
        for fh in files:
            fh.close()
    data, index_items = bytearray(), []
    for i in range(0, len(items), block_entries):
        chunk = items[i:i + block_entries]
        block = _build_block(chunk, restart_interval)
        index_items.append((chunk[-1][0], _enc_varint(len(data)) + _enc_varint(len(block))))
        data += block + _block_trailer(block)
    meta = _build_block([])
    meta_handle = _enc_varint(len(data)) + _enc_varint(len(meta))
    data += meta + _block_trailer(meta)
    index = _build_block(index_items, restart_interval=1)
    index_handle = _enc_varint(len(data)) + _enc_varint(len(index))
    data += index + _block_trailer(index)
    footer = meta_handle + index_handle
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", _TABLE_MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(bytes(data) + footer)


def is_bundle(prefix):
    return os.path.isfile(prefix + ".index")
