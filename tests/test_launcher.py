"""CPU: the multi-GPU launcher (run_extract_embeddings.sh counterpart) -- native scp reading, `ark,scp:` writing,
LPT sharding into per-job tables, one child process per job with per-GPU slots, logs, failure propagation, ordered
concatenation, checkpoint selection, and `--gpu -1` device mapping.  The jobs are tests/helpers/fake_extract_job.py
(same command line as the real driver, no GPU); stages 2-3 are checked against oracle/ref_post.py here and run on
the GPU in tests/test_gpu_launcher.py."""
import os
import sys

import numpy as np
import pytest

from oracle import ref_post
from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, native_ark, run_extract, synth

HELPERS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers")
sys.path.insert(0, HELPERS)
import fake_extract_job  # noqa: E402


def make_data_dir(tmp_path, lens, dim=6, seed=3, n_arks=2, vad=True, fail_key=None):
    """A Kaldi data directory: feats.scp over `n_arks` arks (round-robin, so the table jumps between files), vad.scp,
    spk2utt, utt2num_frames.  Returns (data dir, {key: features}, {key: vad})."""
    data = tmp_path / "data"
    data.mkdir()
    utts = synth.synth_features(len(lens), lens, dim, seed=seed)
    rs = np.random.RandomState(seed)
    feats, vads, lines, vlines = {}, {}, [], []
    arks = [open(data / ("raw.%d.ark" % a), "wb") for a in range(n_arks)]
    vark = open(data / "vad.ark", "wb")
    for i, u in enumerate(utts):
        key = "%sspk%d-utt%03d" % ("FAIL" if fail_key == i else "", i % 3, i)
        f = arks[i % n_arks]
        f.write((key + " ").encode())
        lines.append("%s %s:%d" % (key, data / ("raw.%d.ark" % (i % n_arks)), f.tell()))
        kaldi_io.write_mat(f, u)
        v = (rs.rand(u.shape[0]) > 0.2).astype(np.float32)
        vark.write((key + " ").encode())
        vlines.append("%s %s:%d" % (key, data / "vad.ark", vark.tell()))
        kaldi_io.write_vec_flt(vark, v)
        feats[key], vads[key] = u, v
    for f in arks + [vark]:
        f.close()
    (data / "feats.scp").write_text("\n".join(lines) + "\n")
    if vad:
        (data / "vad.scp").write_text("\n".join(vlines) + "\n")
    spk = {}
    for k in feats:
        spk.setdefault(k.replace("FAIL", "").split("-")[0], []).append(k)
    (data / "spk2utt").write_text("".join("%s %s\n" % (s, " ".join(u)) for s, u in sorted(spk.items())))
    (data / "utt2num_frames").write_text("".join("%s %d\n" % (k, v.shape[0]) for k, v in feats.items()))
    return str(data), feats, vads


def test_native_scp_reader_seeks_across_arks(tmp_path):
    lens = [30, 7, 120, 64, 1, 250, 33]
    data, feats, vads = make_data_dir(tmp_path, lens)
    r = native_ark.ArkBatchReader("scp:" + os.path.join(data, "feats.scp"), batch_frames=200, min_frames=5)
    assert len(r) == len(lens)
    rows, cols = r.shapes()
    assert list(rows) == lens and set(cols) == {6}
    got = {}
    for keys, offsets, data_ in r:
        assert int(offsets[-1]) <= 200 + max(lens)
        for i, k in enumerate(keys):
            got[k] = data_[offsets[i]:offsets[i + 1]].copy()
    assert r.skipped == 1                                             # the 1-frame utterance (< min_frames)
    assert list(got) == [k for k in feats if feats[k].shape[0] >= 5]     # table order
    for k, v in got.items():
        np.testing.assert_array_equal(v, feats[k])
    r.close()
    # reversed table: every record needs a backward seek
    table = native_ark.read_scp_table(os.path.join(data, "feats.scp"))[::-1]
    (tmp_path / "rev.scp").write_text("".join("%s %s\n" % kv for kv in table))
    r = native_ark.ArkBatchReader("scp:" + str(tmp_path / "rev.scp"), batch_frames=10 ** 6)
    keys, offsets, d = r.next_batch()
    assert keys == [k for k, _ in table]
    for i, k in enumerate(keys):
        np.testing.assert_array_equal(d[offsets[i]:offsets[i + 1]], feats[k])
    assert r.next_batch() is None
    r.close()
    # float-vector tables (vad.scp) arrive as [dim, 1] matrices, via scp and via the plain ark
    for spec in ("scp:" + os.path.join(data, "vad.scp"), "ark:" + os.path.join(data, "vad.ark")):
        recs = dict(extract._vad_records(spec))
        assert list(recs) == list(vads)
        for k in vads:
            np.testing.assert_array_equal(recs[k], vads[k])
    assert np.array_equal(native_ark.scp_lengths(os.path.join(data, "feats.scp")), lens)
    assert np.array_equal(native_ark.scp_lengths(os.path.join(data, "feats.scp"), os.path.join(data, "utt2num_frames")), lens)


def test_native_reader_grows_for_an_oversize_utterance(tmp_path):
    big = synth.synth_features(1, [500], 8, seed=1)[0]
    ark = tmp_path / "big.ark"
    with open(ark, "wb") as f:
        kaldi_io.write_mat(f, np.ones((3, 8), np.float32), key="small")
        kaldi_io.write_mat(f, big, key="big")
    r = native_ark.ArkBatchReader("ark:" + str(ark), batch_frames=100, capacity=1024)     # 128 rows of 8 floats
    out = [(k, d.copy()) for keys, off, d in r for k in keys]
    assert [k for k, _ in out] == ["small", "big"]
    np.testing.assert_array_equal(out[1][1], big)


def test_scp_table_with_ranges_is_refused(tmp_path):
    (tmp_path / "r.scp").write_text("utt1 /x/y.ark:12[0:10]\n")
    with pytest.raises(IOError):
        native_ark.ArkBatchReader("scp:" + str(tmp_path / "r.scp"))


def test_vector_writer_ark_scp_offsets(tmp_path):
    keys = ["a", "spk1-utt000002", "z" * 40]
    x = np.random.RandomState(0).randn(3, 17).astype(np.float32)
    ark, scp = str(tmp_path / "v.ark"), str(tmp_path / "v.scp")
    w = native_ark.VectorWriter("ark,scp:%s,%s" % (ark, scp))
    w.write(keys[:1], x[:1])
    w.write(keys[1:], x[1:])
    assert w.close() == 0
    assert [k for k, _ in kaldi_io.read_vec_flt_ark(ark)] == keys
    for i, (k, rx) in enumerate(native_ark.read_scp_table(scp)):
        assert k == keys[i] and rx.startswith(ark + ":")
        np.testing.assert_array_equal(kaldi_io.read_vec_flt(rx), x[i])          # file:offset -> the record body
    # a failing output command is reported (extract.py returns non-zero on it)
    w = native_ark.VectorWriter("ark:| exit 7")
    try:
        w.write(["k"], x[:1])
    except (BrokenPipeError, OSError):
        pass
    try:
        assert w.close() == 7
    except (BrokenPipeError, OSError):
        pass


def test_auto_device_reads_the_job_index():
    f = extract.auto_device
    feat = "ark:apply-cmvn-sliding scp:data/split8/%d/feats.scp ark:- | select-voiced-frames ark:- scp,s,cs:data/split8/%d/vad.scp ark:- |"
    out = "ark:| copy-vector ark:- ark,scp:exp/xvector.%d.ark,exp/xvector.%d.scp"
    assert [f(feat % (j, j), out % (j, j), device_count=8, environ={}) for j in range(1, 9)] == list(range(8))
    assert [f(feat % (j, j), out % (j, j), device_count=4, environ={}) for j in (1, 4, 5, 32)] == [0, 3, 0, 3]
    assert f(feat % (3, 3), "ark:out.ark", device_count=8, environ={}) == 2          # from the rspecifier alone
    assert f("ark:feats.ark", "ark:out.ark", device_count=8, environ={"LOCAL_RANK": "5"}) == 5
    assert f("ark:feats.ark", "ark:out.ark", device_count=8, environ={}) == 0


def test_set_checkpoint_follows_get_checkpoint(tmp_path):
    """misc/utils.py:251-304."""
    nnet = tmp_path / "nnet"
    nnet.mkdir()
    (nnet / "config.json").write_text('{"num_steps_per_epoch": 100}')
    state = 'model_checkpoint_path: "/old/place/model-300"\n' + "".join(
        'all_model_checkpoint_paths: "/old/place/model-%d"\n' % s for s in (100, 200, 300))
    (nnet / "checkpoint").write_text(state)
    (nnet / "valid_loss").write_text("0 2.5 0.1\n1 1.5 0.08\n2 1.9 0.09\n")
    assert run_extract.set_checkpoint(str(nnet), "-1") == str(nnet / "model-200")      # epoch 1 is best -> (1+1)*100
    assert model_io.read_checkpoint_state(str(nnet)) == "model-200"
    lines = (nnet / "checkpoint").read_text().splitlines()
    assert lines[1:] == ['all_model_checkpoint_paths: "%s"' % (nnet / ("model-%d" % s)) for s in (100, 200, 300)]
    assert run_extract.set_checkpoint(str(nnet), "last").endswith("model-300")
    assert run_extract.set_checkpoint(str(nnet), "100").endswith("model-100")
    with pytest.raises(AssertionError):
        run_extract.set_checkpoint(str(nnet), "150")


def _oracle_post(monkeypatch):
    """Stages 2-3 need the GPU; for the CPU orchestration test they are replaced by the oracle (test only)."""
    from tf_kaldi_speaker_amd import postprocess

    def length_normalize(x, scaleup=False, device=0):
        return ref_post.normalize_length(x, scaleup)

    def speaker_mean(keys, x, spk2utt, device=0):
        means, counts = ref_post.speaker_mean(dict(zip(keys, x)), spk2utt)
        return [s for s, _ in means], np.stack([m for _, m in means]), np.array([n for _, n in counts])

    monkeypatch.setattr(postprocess, "length_normalize", length_normalize)
    monkeypatch.setattr(postprocess, "speaker_mean", speaker_mean)


@pytest.mark.parametrize("normalize", [False, True])
def test_launcher_fans_out_and_concatenates_in_input_order(tmp_path, monkeypatch, normalize):
    lens = list(np.random.RandomState(5).randint(30, 400, size=37)) + [12]        # the last one is too short
    data, feats, vads = make_data_dir(tmp_path, lens)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, dict(synth.TDNN_STAT_PARAMS), 6, {}, step=5)
    out = str(tmp_path / "xv")
    monkeypatch.setenv("PYTHONPATH", HELPERS + os.pathsep + os.environ.get("PYTHONPATH", ""))
    _oracle_post(monkeypatch)
    argv = ["--nj", "5", "--gpus", "0,1", "--min-chunk-size", "20", "--normalize", "true" if normalize else "false",
            "--node", "tdnn6_dense", "--job-module", "fake_extract_job", model_dir, data, out]
    assert run_extract.main(argv) == 0
    # per-job tables: an LPT partition of feats.scp, each in table order, loads within one utterance of each other
    keys = list(feats)
    seen, loads = [], []
    for j in range(1, 6):
        tab = [k for k, _ in native_ark.read_scp_table(os.path.join(out, "split5", str(j), "feats.scp"))]
        assert tab == [k for k, _ in native_ark.read_scp_table(os.path.join(out, "split5", str(j), "vad.scp"))]
        assert [keys.index(k) for k in tab] == sorted(keys.index(k) for k in tab)
        seen += tab
        loads.append(sum(feats[k].shape[0] for k in tab))
        log = open(os.path.join(out, "log", "extract.%d.log" % j)).read()
        assert "fake job on device %d" % ((j - 1) % 2) in log                       # job j -> GPU (j-1) mod #GPUs
    assert sorted(seen) == sorted(keys) and max(loads) - min(loads) <= max(lens)
    # xvector.scp: input order, the short utterance dropped, every vector what the job computed
    table = native_ark.read_scp_table(os.path.join(out, "xvector.scp"))
    kept = [k for k in keys if int((vads[k] != 0).sum()) >= 20]
    assert [k for k, _ in table] == kept and len(kept) == len(keys) - 1
    vecs = {}
    for k, rx in table:
        v = kaldi_io.read_vec_flt(rx if not normalize else dict(native_ark.read_scp_table(os.path.join(out, "xvector_before_norm.scp")))[k])
        want = fake_extract_job.embed(feats[k][vads[k] != 0], 0)
        np.testing.assert_array_equal(v[:-1], want[:-1])
        vecs[k] = v
    # stages 2-3 wiring (run_extract_embeddings.sh:80-103)
    spk2utt = [(l.split()[0], l.split()[1:]) for l in open(os.path.join(data, "spk2utt"))]
    x = np.stack([vecs[k] for k in kept])
    xin = ref_post.normalize_length(x) if normalize else x
    means, counts = ref_post.speaker_mean(dict(zip(kept, xin)), spk2utt)
    got = native_ark.read_scp_table(os.path.join(out, "spk_xvector.scp"))
    assert [k for k, _ in got] == [s for s, _ in means]
    for (s, rx), (_, m) in zip(got, means):
        np.testing.assert_array_equal(kaldi_io.read_vec_flt(rx), ref_post.normalize_length(m[None])[0] if normalize else m)
    assert open(os.path.join(out, "num_utts.ark")).read() == "".join("%s %d\n" % sc for sc in counts)
    if normalize:
        for k, rx in table:
            np.testing.assert_array_equal(kaldi_io.read_vec_flt(rx), ref_post.normalize_length(vecs[k][None])[0])


def test_launcher_reports_a_failed_job(tmp_path, monkeypatch, capsys):
    data, feats, _ = make_data_dir(tmp_path, [40, 50, 60, 70], fail_key=2)
    model_dir = str(tmp_path / "exp")
    model_io.save_model(model_dir, dict(synth.TDNN_STAT_PARAMS), 6, {}, step=5)
    monkeypatch.setenv("PYTHONPATH", HELPERS + os.pathsep + os.environ.get("PYTHONPATH", ""))
    rc = run_extract.main(["--nj", "2", "--gpus", "0", "--job-module", "fake_extract_job", model_dir, data, str(tmp_path / "xv")])
    assert rc == 1 and "failed" in capsys.readouterr().out
    assert not os.path.exists(str(tmp_path / "xv" / "xvector.scp"))


def test_launcher_checks_its_inputs(tmp_path, capsys):
    assert run_extract.main(["--gpus", "0", str(tmp_path / "nomodel"), str(tmp_path), str(tmp_path / "o")]) == 1
    assert "No such file" in capsys.readouterr().out


def test_run_jobs_serialises_a_slot_and_overlaps_slots(tmp_path):
    """Jobs of one GPU run one after the other, different GPUs concurrently (4 x 0.4 s on 2 slots ~ 0.8 s)."""
    import time
    cmds = [[sys.executable, "-c", "import time; time.sleep(0.4); print(%d)" % i] for i in range(4)]
    logs = [str(tmp_path / ("%d.log" % i)) for i in range(4)]
    t0 = time.time()
    codes = run_extract.run_jobs(cmds, logs, [0, 1, 0, 1])
    el = time.time() - t0
    assert codes == [0, 0, 0, 0] and 0.75 <= el < 1.6
    assert [open(l).read().splitlines()[-1] for l in logs] == ["0", "1", "2", "3"]


def test_utterances_without_vad_are_skipped_not_fatal(tmp_path):
    """select-voiced-frames ark:- scp,s,cs:vad.scp (run_extract_embeddings.sh:47) warns about an utterance that has no VAD
    decisions and goes on.  Same here: the launcher's shards leave it out, and the job's lock-step lookup returns None for it
    (the front-end then drops it) without losing its place in the sorted table."""
    lens = [40, 50, 60, 70, 80]
    data, feats, vads = make_data_dir(tmp_path, lens)
    keys = list(feats)
    lines = open(os.path.join(data, "vad.scp")).read().splitlines()
    with open(os.path.join(data, "vad.scp"), "w") as f:                 # drop the decisions of the 2nd and the last utterance
        f.write("\n".join(l for l in lines if l.split()[0] not in (keys[1], keys[4])) + "\n")
    out = str(tmp_path / "xv")
    skeys, lengths, shards = run_extract.make_shards(data, out, 2, use_vad=True)
    assert skeys == [keys[0], keys[2], keys[3]] and list(lengths) == [40, 60, 70]
    tabs = []
    for j in (1, 2):
        ft = [k for k, _ in native_ark.read_scp_table(os.path.join(out, "split2", str(j), "feats.scp"))]
        assert ft == [k for k, _ in native_ark.read_scp_table(os.path.join(out, "split2", str(j), "vad.scp"))]
        tabs += ft
    assert sorted(tabs) == sorted(skeys)
    look = extract._vad_lookup("scp:" + os.path.join(data, "vad.scp"))
    got = [look(k) for k in keys]                                          # features of all five arrive at the job
    assert got[1] is None and got[4] is None
    for i in (0, 2, 3):
        np.testing.assert_array_equal(got[i], vads[keys[i]])
    with pytest.raises(ValueError):                                        # nothing left at all: that is an error
        open(os.path.join(data, "vad.scp"), "w").write("nobody %s:0\n" % os.path.join(data, "vad.ark"))
        run_extract.make_shards(data, out, 2, use_vad=True)
