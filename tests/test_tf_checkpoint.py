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
