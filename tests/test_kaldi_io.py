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


# ------------------------------------------------------------------ native batch reader / formatter (csrc/ark_io.cpp)
@pytest.fixture(scope="module")
def native():
    import __graft_entry__ as g
    g.build()
    from tf_kaldi_speaker_amd import native_ark
    return native_ark


def test_native_reader_decodes_reference_fixture(native):
    """FM / DM / CM records of tests/golden/feats.ark (written / decoded by the reference's kaldi_io.py)."""
    exp = np.load(os.path.join(GOLD, "feats_expected.npz"))
    got = []
    for keys, off, feats in native.ArkBatchReader(os.path.join(GOLD, "feats.ark"), batch_frames=1000):
        for i, k in enumerate(keys):                       # the three records differ in width -> one batch each
            got.append((k, feats[off[i]:off[i + 1]].copy()))
    assert [k for k, _ in got] == list(exp["keys"])
    np.testing.assert_array_equal(got[0][1], exp["m0"])                              # FM: bit exact
    np.testing.assert_array_equal(got[1][1], exp["m1"].astype(np.float32))           # DM: rounded to float32
    np.testing.assert_allclose(got[2][1], exp["m2"], rtol=0, atol=9.6e-7)            # CM: see test above
    py = [m for _, m in kaldi_io.read_mat_ark(os.path.join(GOLD, "feats.ark"))]
    np.testing.assert_array_equal(got[2][1], py[2])                                  # identical to the Python decoder


def test_native_reader_batching_skip_and_pipe(native, tmp_path):
    rs = np.random.RandomState(1)
    lens = [30, 10, 200, 25, 24, 999, 50, 61]
    mats = [rs.standard_normal((t, 5)).astype(np.float32) for t in lens]
    path = str(tmp_path / "f.ark")
    with open(path, "wb") as f:
        for i, m in enumerate(mats):
            kaldi_io.write_mat(f, m, key="k%d" % i)
    for spec in (path, "ark:" + path, "ark:cat %s |" % path):
        r = native.ArkBatchReader(spec, batch_frames=230, min_frames=25)
        batches = list((k, o.copy(), f.copy()) for k, o, f in r)
        assert r.skipped == 2                                                        # T=10 and T=24 (extract.py:65-67)
        keys = [k for b in batches for k in b[0]]
        assert keys == ["k0", "k2", "k3", "k5", "k6", "k7"]                          # input order
        assert [len(b[0]) for b in batches] == [2, 2, 2]                             # stop once >= 230 frames
        for ks, off, feats in batches:
            for i, k in enumerate(ks):
                np.testing.assert_array_equal(feats[off[i]:off[i + 1]], mats[int(k[1:])])
        r.close()


def test_native_reader_errors(native, tmp_path):
    p = tmp_path / "bad.ark"
    p.write_bytes(b"key \0BFM \x04\x05\0\0\0\x04\x02\0\0\0abc")                     # truncated payload
    with pytest.raises(IOError):
        list(native.ArkBatchReader(str(p)))
    p.write_bytes(b"key \0BXX \x04\x05\0\0\0")
    with pytest.raises(IOError):
        list(native.ArkBatchReader(str(p)))
    p.write_bytes(b"")
    assert list(native.ArkBatchReader(str(p))) == []
    with pytest.raises(ValueError):
        native.ArkBatchReader("feats.ark.gz")


@pytest.mark.parametrize("threads", [1, 3, 16])
def test_native_reader_mapped_file_with_copy_threads(native, tmp_path, threads):
    """An ark opened by name is mapped and the payloads of a batch are pread() by several threads, each taking a
    contiguous share of the batch's bytes (a share may begin inside a record): every byte must arrive, for any thread
    count, in ark and in scp mode (two arks, so that the mapping changes inside a batch), and a file that ends inside a
    payload is an error, not a short copy."""
    rs = np.random.RandomState(5)
    lens = rs.randint(100, 900, size=60)
    mats = [rs.standard_normal((int(t), 30)).astype(np.float32) for t in lens]      # 3.6 MB: above the threading threshold
    paths = [str(tmp_path / "a.ark"), str(tmp_path / "b.ark")]
    lines = []
    for j, path in enumerate(paths):
        with open(path, "wb") as f:
            for i in range(j, len(mats), 2):
                f.write(("u%03d " % i).encode())
                lines.append((i, "u%03d %s:%d" % (i, path, f.tell())))
                kaldi_io.write_mat(f, mats[i])
    scp = str(tmp_path / "feats.scp")
    with open(scp, "w") as f:
        f.write("\n".join(line for _, line in sorted(lines)) + "\n")
    for spec, order in (("ark:" + paths[0], list(range(0, len(mats), 2))), ("scp:" + scp, list(range(len(mats))))):
        r = native.ArkBatchReader(spec, batch_frames=10 ** 6, copy_threads=threads)
        got = [(k, feats[off[i]:off[i + 1]].copy()) for ks, off, feats in r for i, k in enumerate(ks)]
        r.close()
        assert [k for k, _ in got] == ["u%03d" % i for i in order]
        for (k, m), i in zip(got, order):
            np.testing.assert_array_equal(m, mats[i])
    with open(paths[0], "r+b") as f:
        f.truncate(os.path.getsize(paths[0]) - 1000)
    with pytest.raises(IOError):
        list(native.ArkBatchReader("ark:" + paths[0], batch_frames=10 ** 6, copy_threads=threads))


def test_native_format_vectors_is_byte_identical(native):
    exp = np.load(os.path.join(GOLD, "vectors_expected.npz"))
    with open(os.path.join(GOLD, "vectors.ark"), "rb") as f:
        ref = f.read()
    first = native.format_vectors([str(exp["keys"][0])], exp["v0"][None])
    assert ref.startswith(first)                                                     # the float32 record of the fixture
    rs = np.random.RandomState(0)
    v = rs.standard_normal((7, 33)).astype(np.float32)
    keys = ["spk%d-utt_%d" % (i, i * i) for i in range(7)]
    buf = io.BytesIO()
    for k, x in zip(keys, v):
        kaldi_io.write_vec_flt(buf, x, key=k)
    assert native.format_vectors(keys, v) == buf.getvalue()


def test_read_mat_scp_with_offsets(tmp_path):
    """scp lines `key file:offset` (dataset/kaldi_io.py:953-972): offsets point at the byte after `key `."""
    from tf_kaldi_speaker_amd import kaldi_io
    rng = np.random.default_rng(3)
    mats = {"uttA": rng.standard_normal((7, 5)).astype(np.float32), "uttB": rng.standard_normal((3, 5)).astype(np.float32)}
    ark = str(tmp_path / "feats.ark")
    offsets = {}
    with open(ark, "wb") as f:
        for k, m in mats.items():
            f.write((k + " ").encode())
            offsets[k] = f.tell()
            kaldi_io.write_mat(f, m)
    scp = str(tmp_path / "feats.scp")
    with open(scp, "w") as f:
        for k in mats:
            f.write("%s %s:%d\n" % (k, ark, offsets[k]))
    got = dict(kaldi_io.read_mat_scp(scp))
    assert list(got) == list(mats)
    for k in mats:
        np.testing.assert_array_equal(got[k], mats[k])
