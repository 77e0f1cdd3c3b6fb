"""CPU: ark reader/writer against fixtures produced by the reference's dataset/kaldi_io.py
(tests/golden/feats.ark, vectors.ark + *_expected.npz; see make_golden.py)."""
import io
import os

import numpy as np
import pytest

from tf_kaldi_speaker_amd import kaldi_io

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_read_mat_ark_fm_dm_cm_matches_reference_decoder():
    exp = np.load(os.path.join(GOLD, "feats_expected.npz"))
    got = list(kaldi_io.read_mat_ark(os.path.join(GOLD, "feats.ark")))
    assert [k for k, _ in got] == list(exp["keys"])
    for i, (_, m) in enumerate(got):
        ref = exp["m%d" % i]
        assert m.dtype == ref.dtype and m.shape == ref.shape
        if i < 2:
            np.testing.assert_array_equal(m, ref)        # FM / DM payloads: bit exact
        else:
            # 'CM ': the fixture was decoded by the reference under NumPy 2 (float32 intermediates);
            # ours rounds once from float64 like Kaldi itself -> within one float32 ulp of the
            # column's value range (|x| < 16 here -> 9.6e-7).
            np.testing.assert_allclose(m, ref, rtol=0, atol=9.6e-7)


def test_write_vec_flt_is_byte_identical_to_reference_writer():
    exp = np.load(os.path.join(GOLD, "vectors_expected.npz"))
    buf = io.BytesIO()
    for i, k in enumerate(exp["keys"]):
        kaldi_io.write_vec_flt(buf, exp["v%d" % i], key=str(k))
    with open(os.path.join(GOLD, "vectors.ark"), "rb") as f:
        assert buf.getvalue() == f.read()
    back = list(kaldi_io.read_vec_flt_ark(os.path.join(GOLD, "vectors.ark")))
    assert [k for k, _ in back] == list(exp["keys"])
    for i, (_, v) in enumerate(back):
        np.testing.assert_array_equal(v, exp["v%d" % i])


def test_write_mat_round_trip_and_specifiers(tmp_path):
    rs = np.random.RandomState(0)
    mats = [("a", rs.standard_normal((4, 3)).astype(np.float32)), ("b_2", rs.standard_normal((1, 5)).astype(np.float64))]
    path = str(tmp_path / "m.ark")
    with open(path, "wb") as f:
        for k, m in mats:
            kaldi_io.write_mat(f, m, key=k)
    for spec in (path, "ark:" + path, "ark:cat %s |" % path):
        got = list(kaldi_io.read_mat_ark(spec))
        assert [k for k, _ in got] == ["a", "b_2"]
        for (_, g), (_, m) in zip(got, mats):
            np.testing.assert_array_equal(g, m)
    # output pipe  "ark:| cmd"
    out = str(tmp_path / "v.ark")
    fd = kaldi_io.open_or_fd("ark:| cat > %s" % out, "wb")
    kaldi_io.write_vec_flt(fd, np.arange(4, dtype=np.float32), key="k")
    fd.close()
    fd._xv_proc.wait()
    assert list(kaldi_io.read_vec_flt_ark(out))[0][1].tolist() == [0, 1, 2, 3]
    # gz + offset forms of open_or_fd (dataset/kaldi_io.py:633-655)
    import gzip
    with gzip.open(str(tmp_path / "m.ark.gz"), "wb") as f:
        f.write(open(path, "rb").read())
    assert len(list(kaldi_io.read_mat_ark(str(tmp_path / "m.ark.gz")))) == 2
    fd = kaldi_io.open_or_fd(path + ":2")              # skip key "a "
    np.testing.assert_array_equal(kaldi_io.read_mat(fd), mats[0][1])


def test_ascii_matrix_and_vector():
    m = kaldi_io.read_mat(io.BytesIO(b" [\n 1 2 3\n 4 5 6 ]\n"))
    np.testing.assert_array_equal(m, np.array([[1, 2, 3], [4, 5, 6]], np.float32))
    v = kaldi_io.read_vec_flt(io.BytesIO(b" [ 1.5 2 ]\n"))
    assert v.tolist() == [1.5, 2.0]


def test_bad_inputs_raise():
    with pytest.raises(kaldi_io.UnknownMatrixHeader):
        kaldi_io.read_mat(io.BytesIO(b"\0BXX \x04\0\0\0\0"))
    with pytest.raises(kaldi_io.UnsupportedDataType):
        kaldi_io.write_vec_flt(io.BytesIO(), np.zeros(3, np.int32), key="k")
    with pytest.raises(kaldi_io.BadInputFormat):
        kaldi_io.read_mat(io.BytesIO(b"\0BFM \x04\x05\0\0\0\x04\x02\0\0\0abc"))     # truncated payload
    assert list(kaldi_io.read_mat_ark(io.BytesIO(b""))) == []                    # empty ark


def test_sequential_reads_on_caller_owned_descriptor():
    """read_key/read_mat on a raw descriptor consume exactly one record (no read-ahead)."""
    buf = io.BytesIO()
    kaldi_io.write_mat(buf, np.ones((2, 2), np.float32), key="u1")
    kaldi_io.write_mat(buf, np.zeros((1, 2), np.float32), key="u2")
    buf.seek(0)
    assert kaldi_io.read_key(buf) == "u1"
    assert kaldi_io.read_mat(buf).shape == (2, 2)
    assert kaldi_io.read_key(buf) == "u2"
    assert kaldi_io.read_mat(buf).shape == (1, 2)
    assert kaldi_io.read_key(buf) is None
