"""CPU: TensorFlow-free checkpoint-V2 bundle reader (tf-kaldi-speaker_amd/tf_checkpoint.py).

**parity unpinned**: no real TF checkpoint is available offline (the reference only links to them,
README.md:101-119), so the reader is exercised against the module's own writer, which emits the
published SSTable / BundleEntryProto layout."""
import os
import struct

import numpy as np
import pytest

from tf_kaldi_speaker_amd import model_io, synth, tf_checkpoint


def test_bundle_round_trip_multi_block(tmp_path):
    p = dict(synth.TDNN_ATT_PARAMS, num_nodes_pooling_layer=48, att_key_num_nodes=[24, 16], network_relu_type="prelu")
    w = synth.synth_weights(p, 13, seed=1, channels=32)
    extra = dict(w)
    extra["tdnn/tdnn1_conv/kernel/Momentum"] = np.zeros((1, 5, 13, 32), np.float32)      # optimizer slot
    extra["global_step"] = np.array(123456, np.int64)
    extra["softmax/output/kernel"] = np.ones((32, 7), np.float64)
    prefix = str(tmp_path / "model-2610000")
    tf_checkpoint.write_bundle(prefix, extra, block_entries=5)
    header, entries = tf_checkpoint.read_index(prefix + ".index")
    assert header["num_shards"] == 1 and set(entries) == set(extra)
    assert entries["tdnn/tdnn3_conv/kernel"]["shape"] == [1, 7, 32, 32]
    back = tf_checkpoint.read_bundle(prefix)
    assert set(back) == set(extra)
    for k, v in extra.items():
        assert back[k].dtype == np.asarray(v).dtype and back[k].shape == np.asarray(v).shape
        np.testing.assert_array_equal(back[k], v)
    assert back["global_step"] == 123456


def test_model_dir_with_tf_bundle(tmp_path):
    """nnet/checkpoint -> model-<step>.index/.data: loaded without TF; optimizer slots filtered out, the softmax
    kernel kept (extract_angle.py needs it; the predict graph ignores it)."""
    p = dict(synth.TDNN_STAT_PARAMS, num_nodes_pooling_layer=16)
    w = synth.synth_weights(p, 6, seed=2, channels=16)
    nnet = model_io.save_model(str(tmp_path / "exp"), p, 6, w, step=42)
    os.remove(os.path.join(nnet, "model-42.npz"))
    extra = dict(w)
    extra["tdnn/tdnn2_conv/bias/Momentum"] = np.zeros(16, np.float32)
    extra["softmax/output/kernel"] = np.ones((16, 3), np.float32)
    extra["softmax/output/kernel/Momentum"] = np.ones((16, 3), np.float32)
    tf_checkpoint.write_bundle(os.path.join(nnet, "model-42"), extra)
    got, step = model_io.load_weights(nnet)
    assert step == 42 and set(got) == set(w) | {model_io.SOFTMAX_KERNEL}
    for k in w:
        np.testing.assert_array_equal(got[k], w[k])


def test_bad_index_files(tmp_path):
    p = tmp_path / "m.index"
    p.write_bytes(b"short")
    with pytest.raises(tf_checkpoint.CheckpointFormatError):
        tf_checkpoint.read_index(str(p))
    p.write_bytes(b"\\0" * 40 + struct.pack("<Q", 0x1234))
    with pytest.raises(tf_checkpoint.CheckpointFormatError):
        tf_checkpoint.read_index(str(p))
    # compressed block type is refused, not misparsed
    prefix = str(tmp_path / "c")
    tf_checkpoint.write_bundle(prefix, {"tdnn/x/kernel": np.ones((2, 2), np.float32)})
    raw = bytearray(open(prefix + ".index", "rb").read())
    footer = bytes(raw[-48:])
    _, _, pos = tf_checkpoint._block_handle(footer, 0)
    ioff, isize, _ = tf_checkpoint._block_handle(footer, pos)
    _, handle = next(tf_checkpoint._read_block(bytes(raw), ioff, isize))
    boff, bsize, _ = tf_checkpoint._block_handle(handle, 0)
    raw[boff + bsize] = 1                                                 # compression byte of the first data block: snappy
    open(prefix + ".index", "wb").write(bytes(raw))
    with pytest.raises(tf_checkpoint.CheckpointFormatError):
        tf_checkpoint.read_bundle(prefix)


def _crc32c_bitwise(data):
    """Independent bit-by-bit CRC-32C (reflected 0x82F63B78) for the hand-assembled index below."""
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
    return c ^ 0xFFFFFFFF


def test_crc32c_known_answers():
    """RFC 3720 appendix B.4 vectors + the classic check value; the native routine (SSE4.2 / slicing-by-8) and the table loop agree,
    also when the checksum is continued across pieces; LevelDB's mask round-trips."""
    ramp = bytes(range(32))
    kats = [(b"123456789", 0xE3069283), (bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43), (ramp, 0x46DD794E),
            (ramp[::-1], 0x113FDB5C)]
    for data, want in kats:
        assert tf_checkpoint._crc32c_py(0, data) == want
        assert tf_checkpoint.crc32c(data) == want
        assert _crc32c_bitwise(data) == want
    big = np.random.RandomState(0).bytes(100003)
    whole = tf_checkpoint.crc32c(big)
    assert whole == tf_checkpoint._crc32c_py(0, big)
    assert tf_checkpoint.crc32c(big[4097:], tf_checkpoint.crc32c(big[:4097])) == whole
    for c in (0, 1, 0xE3069283, 0xFFFFFFFF):
        assert tf_checkpoint.unmask_crc(tf_checkpoint.mask_crc(c)) == c
    assert tf_checkpoint.mask_crc(0xE3069283) != 0xE3069283


def test_two_shards_restart_points_and_corruption(tmp_path):
    """A bundle of two data shards whose index blocks hold more entries than the restart interval (so keys are rebuilt from shared
    prefixes across restart points), every checksum verified on the way in; then one flipped byte in a data shard, and one in an
    index block, must each be reported as a checksum error -- not returned as weights."""
    rs = np.random.RandomState(3)
    tensors = {"tdnn/tdnn%d_%s/%s" % (i, kind, leaf): rs.standard_normal((3, 1 + i)).astype(np.float32)
               for i in range(1, 8) for kind in ("conv", "bn") for leaf in ("kernel", "bias", "gamma")}
    tensors["global_step"] = np.array(7, np.int64)
    prefix = str(tmp_path / "model-7")
    tf_checkpoint.write_bundle(prefix, tensors, block_entries=40, shards=2, restart_interval=4)
    assert os.path.isfile(prefix + ".data-00000-of-00002") and os.path.isfile(prefix + ".data-00001-of-00002")
    header, entries = tf_checkpoint.read_index(prefix + ".index")
    assert header["num_shards"] == 2 and {e["shard_id"] for e in entries.values()} == {0, 1}
    assert all(e["crc32c"] is not None for e in entries.values())
    back = tf_checkpoint.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        np.testing.assert_array_equal(back[k], v)
    shard = prefix + ".data-00001-of-00002"
    raw = bytearray(open(shard, "rb").read())
    raw[len(raw) // 2] ^= 0x10
    open(shard, "wb").write(bytes(raw))
    with pytest.raises(tf_checkpoint.CheckpointFormatError, match="checksum"):
        tf_checkpoint.read_bundle(prefix)
    raw[len(raw) // 2] ^= 0x10
    open(shard, "wb").write(bytes(raw))
    tf_checkpoint.read_bundle(prefix)
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[20] ^= 0x01                                                       # inside the first data block
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(tf_checkpoint.CheckpointFormatError, match="checksum"):
        tf_checkpoint.read_index(prefix + ".index")


def test_hand_assembled_index(tmp_path):
    """An index written out byte by byte from the published layout (LevelDB table_format.md; tensor_bundle.proto), NOT through
    write_bundle: one data block {"" -> BundleHeaderProto, "a/b" -> entry, "a/c" -> entry (key sharing the prefix "a/")}, an empty
    metaindex block, an index block, the 48-byte footer.  Checksums come from the bit-by-bit CRC above."""
    def trailer(block):
        return b"\x00" + struct.pack("<I", tf_checkpoint.mask_crc(_crc32c_bitwise(block + b"\x00")))
    t_b = np.array([1.5, -2.0], np.float32)
    t_c = np.array([[3]], np.int32)
    data = t_b.tobytes() + t_c.tobytes()
    open(str(tmp_path / "m.data-00000-of-00001"), "wb").write(data)
    crc_b = struct.pack("<I", tf_checkpoint.mask_crc(_crc32c_bitwise(t_b.tobytes())))
    crc_c = struct.pack("<I", tf_checkpoint.mask_crc(_crc32c_bitwise(t_c.tobytes())))
    header = b"\x08\x01" + b"\x10\x00"                                   # num_shards = 1, endianness = LITTLE
    #            dtype=DT_FLOAT  shape{dim{size 2}}            size=8     crc32c (fixed32, field 6)
    entry_b = b"\x08\x01" + b"\x12\x04\x12\x02\x08\x02" + b"\x28\x08" + b"\x35" + crc_b
    #            dtype=DT_INT32  shape{dim{1} dim{1}}                         offset=8   size=4
    entry_c = b"\x08\x03" + b"\x12\x08\x12\x02\x08\x01\x12\x02\x08\x01" + b"\x20\x08" + b"\x28\x04" + b"\x35" + crc_c
    block = (b"\x00\x00" + bytes([len(header)]) + header                  # shared 0, non-shared 0 (key ""), value
             + b"\x00\x03" + bytes([len(entry_b)]) + b"a/b" + entry_b    # shared 0, non-shared 3
             + b"\x02\x01" + bytes([len(entry_c)]) + b"c" + entry_c      # shared 2 ("a/"), non-shared 1
             + struct.pack("<I", 0) + struct.pack("<I", 1))               # restart array [0], num_restarts = 1
    meta = struct.pack("<I", 0) + struct.pack("<I", 1)                     # empty block: one restart at 0
    meta_off = len(block) + 5
    index_off = meta_off + len(meta) + 5
    handle = bytes([0, len(block)])                                        # BlockHandle(offset 0, size): varints < 128
    index = (b"\x00\x03" + bytes([len(handle)]) + b"a/c" + handle + struct.pack("<I", 0) + struct.pack("<I", 1))
    footer = bytes([meta_off, len(meta)]) + bytes([index_off, len(index)])
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xdb4775248b80fb57)
    assert len(block) < 128 and index_off < 128
    open(str(tmp_path / "m.index"), "wb").write(block + trailer(block) + meta + trailer(meta) + index + trailer(index) + footer)
    hdr, entries = tf_checkpoint.read_index(str(tmp_path / "m.index"))
    assert hdr == {"num_shards": 1, "endianness": 0}
    assert entries["a/b"]["shape"] == [2] and entries["a/b"]["offset"] == 0 and entries["a/c"]["shape"] == [1, 1]
    assert entries["a/c"]["offset"] == 8 and entries["a/c"]["dtype"] == 3
    got = tf_checkpoint.read_bundle(str(tmp_path / "m"))
    np.testing.assert_array_equal(got["a/b"], t_b)
    np.testing.assert_array_equal(got["a/c"], t_c)
    # and the module's own writer produces a byte-identical data block layout for the same content
    tf_checkpoint.write_bundle(str(tmp_path / "w"), {"a/b": t_b, "a/c": t_c})
    assert tf_checkpoint.read_index(str(tmp_path / "w.index"))[1] == entries
