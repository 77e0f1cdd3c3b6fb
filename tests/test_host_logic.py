"""CPU: host-side logic that needs no GPU -- Params, model dir contract, extract.py driver
behaviour (skip / chunk / weight / normalise / order) against oracle.ref_numpy.extract_utterance,
and the C-ABI library's exported symbols."""
import ctypes
import io
import json
import os
import re

import numpy as np
import pytest

from oracle import ref_numpy
from tf_kaldi_speaker_amd import extract, kaldi_io, model_io, synth
from tf_kaldi_speaker_amd.params import Params


def _fake_embed(utts):
    """Deterministic stand-in for the network: depends on content and length."""
    return np.stack([np.concatenate([u.mean(0), u.std(0) + 0.01 * u.shape[0]]) for u in utts])


def _fake_predict(x):
    x = np.asarray(x)
    return _fake_embed(list(x)) if x.ndim == 3 else _fake_embed([x])[0]


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("chunk", [40, 41, 10000])
def test_extract_stream_matches_reference_driver_semantics(normalize, chunk):
    """egs/voxceleb/v1/nnet/lib/extract.py:64-93 incl. S=40 chunking cases T in {39,40,41,61,100}."""
    lens = [39, 40, 41, 61, 100, 24, 25, 200]
    items = [("utt%d" % i, u) for i, u in enumerate(synth.synth_features(len(lens), lens, 5, seed=1))]
    out = []
    done, skipped = extract.extract_stream(_fake_embed, iter(items), lambda k, v: out.append((k, v)),
                                           min_chunk_size=25, chunk_size=chunk, normalize=normalize, batch_frames=150)
    expect = []
    for k, f in items:
        e = ref_numpy.extract_utterance(f, _fake_predict, 25, chunk, normalize)
        if e is not None:
            expect.append((k, e))
    assert skipped == 1 and done == len(expect)
    assert [k for k, _ in out] == [k for k, _ in expect]            # input order kept across batches
    for (_, a), (_, b) in zip(out, expect):
        assert a.dtype == np.float32
        np.testing.assert_allclose(a, b, rtol=1e-6, atol=1e-7)


def test_split_chunks_formula():
    """num_chunks = ceil((T-S)/(S//2)) + 1, starts i*(S//2), last length T-start (extract.py:70-76)."""
    assert extract.split_chunks(41, 40) == [(0, 40), (20, 21)]
    assert extract.split_chunks(100, 40) == [(0, 40), (20, 40), (40, 40), (60, 40)]
    assert extract.split_chunks(101, 40) == [(0, 40), (20, 40), (40, 40), (60, 40), (80, 21)]
    assert extract.split_chunks(61, 41) == [(0, 41), (20, 41)]     # odd S: hop = 20


def test_cli_parser_keeps_reference_flags():
    a = extract.build_parser().parse_args(["-g", "3", "-m", "50", "-s", "3000", "-n", "--node", "output", "m", "ark:in", "ark:out"])
    assert (a.gpu, a.min_chunk_size, a.chunk_size, a.normalize, a.node) == (3, 50, 3000, True, "output")
    assert (a.model_dir, a.rspecifier, a.wspecifier) == ("m", "ark:in", "ark:out")
    d = extract.build_parser().parse_args(["m", "r", "w"])
    assert (d.gpu, d.min_chunk_size, d.chunk_size, d.normalize, d.node) == (-1, 25, 10000, False, "")


def test_params_attribute_and_dict_access(tmp_path):
    """misc/utils.py:13-41."""
    path = tmp_path / "c.json"
    path.write_text(json.dumps({"network_type": "tdnn", "embedding_node": "tdnn6_dense"}))
    p = Params(str(path))
    assert p.network_type == "tdnn" and "pooling_type" not in p.dict
    p.embedding_node = "output"
    assert p.dict["embedding_node"] == "output"
    p.dict["num_nodes_pooling_layer"] = 1500
    assert p.num_nodes_pooling_layer == 1500
    p.save(str(tmp_path / "d.json"))
    assert Params(str(tmp_path / "d.json")).embedding_node == "output"


def test_model_dir_round_trip(tmp_path):
    p = dict(synth.TDNN_STAT_PARAMS)
    p["num_nodes_pooling_layer"] = 8
    w = synth.synth_weights(p, 5, seed=1, channels=16)
    nnet = model_io.save_model(str(tmp_path / "exp"), p, 5, w, step=420000)
    assert sorted(os.listdir(nnet)) == ["checkpoint", "config.json", "feature_dim", "model-420000.npz"]
    w2, step = model_io.load_weights(nnet)
    assert step == 420000 and set(w2) == set(w)
    for k in w:
        np.testing.assert_array_equal(w[k], w2[k])
    assert model_io.load_weights(str(tmp_path)) == (None, None)


def test_synth_weights_cover_reference_variable_names():
    w = synth.synth_weights(dict(synth.TDNN_ATT_PARAMS, network_relu_type="prelu", att_apply_nonlinear=True), 30, seed=0)
    assert w["tdnn/tdnn1_conv/kernel"].shape == (1, 5, 30, 512)
    assert w["tdnn/tdnn3_conv/kernel"].shape == (1, 7, 512, 512)
    assert w["tdnn/tdnn5_dense/kernel"].shape == (512, 1500)
    assert w["tdnn/tdnn6_dense/kernel"].shape == (3000, 512)
    assert w["tdnn/attention/att_key0/att_key0_dense/kernel"].shape == (512, 1500)
    assert w["tdnn/attention/att_key1/att_key1_dense/kernel"].shape == (1500, 1500)
    assert "tdnn/attention/att_key1/att_key1_bn/gamma" not in w          # type 1 = affine + relu
    assert w["tdnn/attention/query"].shape == (1, 1500)
    assert w["tdnn/tdnn4_relu/alpha"].shape == (512,)
    assert w["tdnn/attention/att_post_bn/gamma"].shape == (3000,)


def test_c_abi_library_exports_every_declared_symbol(repo_root):
    """The shared library loads on a GPU-less box and exports what include/xvec_hip.h declares."""
    import __graft_entry__ as g
    g.build()
    hdr = open(os.path.join(repo_root, "include", "xvec_hip.h")).read()
    declared = set(re.findall(r"\b(xv_[a-z0-9_]+)\s*\(", hdr))
    from tf_kaldi_speaker_amd import _lib
    assert declared == set(_lib.EXPORTS)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    lib.xv_version.restype = ctypes.c_char_p
    assert b"gfx950" in lib.xv_version()
    # the ctypes mirrors have the size the C compiler gives the structs of the header
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        src = os.path.join(tmp, "sz.c")
        with open(src, "w") as f:
            f.write('#include <stdio.h>\n#include "xvec_hip.h"\nint main(void) { printf("%zu %zu %zu\\n", sizeof(xv_model_desc), '
                    'sizeof(xv_plan_info), sizeof(xv_kernel_time)); return 0; }\n')
        exe = os.path.join(tmp, "sz")
        subprocess.run(["gcc", "-I", os.path.join(repo_root, "include"), src, "-o", exe], check=True)
        sizes = [int(x) for x in subprocess.run([exe], capture_output=True, text=True, check=True).stdout.split()]
    assert sizes == [ctypes.sizeof(_lib.ModelDesc), ctypes.sizeof(_lib.PlanInfo), ctypes.sizeof(_lib.KernelTime)]


def test_trainer_refuses_unsupported_graphs_and_missing_gpu():
    from tf_kaldi_speaker_amd.trainer import Trainer
    with pytest.raises(NotImplementedError):
        Trainer(Params(**dict(synth.TDNN_STAT_PARAMS, network_type="tdnn-s")), None, 30)
    rn = Trainer(Params(**dict(synth.RESNET_PARAMS, pooling_type="self_attention")), None, 40)
    with pytest.raises(NotImplementedError):
        rn.build("predict")
    with pytest.raises(NotImplementedError):
        Trainer(Params(**dict(synth.TDNN_STAT_PARAMS, network_type="nonsense")), None, 30)
    tr = Trainer(Params(**dict(synth.TDNN_STAT_PARAMS, pooling_type="ghost_vlad")), None, 30)
    with pytest.raises(NotImplementedError):
        tr.build("predict")
    import torch
    if not torch.cuda.is_available():
        tr = Trainer(Params(**dict(synth.TDNN_STAT_PARAMS)), None, 30)
        tr.build("predict")
        with pytest.raises(RuntimeError):                       # no silent CPU fallback
            tr.load_weights(synth.synth_weights(synth.TDNN_STAT_PARAMS, 30, channels=8))


def test_extract_frame_stream_padding_and_chunking():
    """egs/voxceleb/v1/nnet/lib/extract_frame.py:64-94 with a stand-in network of context 14."""
    from tf_kaldi_speaker_amd import extract_frame

    def embed(utts):              # frame-level stand-in: valid window of 15 frames -> T - 14 rows
        return [np.stack([u[t:t + 15].mean(0) for t in range(u.shape[0] - 14)]) for u in utts]

    lens = [24, 25, 40, 100, 130, 230]     # (a trailing chunk shorter than the context fails in the reference too)
    items = [("u%d" % i, u) for i, u in enumerate(synth.synth_features(len(lens), lens, 4, seed=2))]
    out = []
    done, skipped = extract_frame.extract_frames_stream(embed, iter(items), lambda k, m: out.append((k, m)),
                                                        min_chunk_size=25, chunk_size=100, batch_frames=150)
    assert (done, skipped) == (5, 1) and [k for k, _ in out] == ["u1", "u2", "u3", "u4", "u5"]
    for (k, m), (_, f) in zip(out, items[1:]):
        assert m.shape == (f.shape[0], 4) and m.dtype == np.float32
    # reference arithmetic for one chunked utterance (T=230, S=100: chunks 100, 100, 30)
    f = items[5][1]
    parts = []
    for s0, n in ((0, 100), (100, 100), (200, 30)):
        e = embed([f[s0:s0 + n]])[0]
        pad = (n - e.shape[0]) // 2
        parts.append(np.concatenate([np.tile(e[0], [pad, 1]), e, np.tile(e[-1], [pad, 1])], axis=0))
    np.testing.assert_allclose(out[4][1], np.concatenate(parts, axis=0), rtol=1e-6)
    assert extract_frame.split_plain(230, 100) == [(0, 100), (100, 100), (200, 30)]
    a = extract_frame.build_parser().parse_args(["--node", "tdnn4_relu", "m", "r", "w"])
    assert (a.gpu, a.min_chunk_size, a.chunk_size, a.node) == (-1, 25, 10000, "tdnn4_relu")


def test_asm_guard_catches_an_in_flight_register_and_a_compiler_load(repo_root):
    """tools/asm_guard.py (run by __graft_entry__.build on the emitted gfx950 assembly of the hand-scheduled kernels): a register
    that is the destination of a load still in flight must not be touched before the counted wait that retires it -- also across
    a loop's back edge -- and the K loop must contain no vector-memory instruction outside the kernels' asm statements."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("asm_guard", os.path.join(repo_root, "tools", "asm_guard.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    ok = """
	s_waitcnt vmcnt(0)
.LBB0_1:
	;;#ASMSTART
	global_load_dwordx4 v[10:13], v1, s[2:3] offset:0
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_mfma_f32_16x16x32_f16 v[20:23], v[14:17], v[30:33], v[20:23]
	;;#ASMSTART
	global_load_dwordx4 v[14:17], v1, s[2:3] offset:1024
	;;#ASMEND
	;;#ASMSTART
	s_waitcnt vmcnt(1)
	;;#ASMEND
	v_mfma_f32_16x16x32_f16 v[20:23], v[10:13], v[30:33], v[20:23]
	s_cbranch_scc1 .LBB0_1
	s_endpgm
""".splitlines()
    viol, st = g.check_kernel(ok)
    assert viol == [] and st["mfma"] == 2 and st["asm_loads"] == 1, (viol, st)
    # the same loop with the second wait one too lax: on the NEXT iteration v[10:13] is reloaded while ... no: the MFMA reads
    # v[10:13] while its load from the top of this iteration may still be in flight
    lax = [l.replace("s_waitcnt vmcnt(1)", "s_waitcnt vmcnt(2)") if i > 10 else l for i, l in enumerate(ok)]
    viol, _ = g.check_kernel(lax)
    assert any("v[10" in v or "[10," in v for v in viol), viol
    # a compiler-generated spill reload between the MFMAs
    spill = list(ok)
    spill.insert(10, "	scratch_load_dword v40, off, off offset:16")
    viol, st = g.check_kernel(spill)
    assert st["compiler_vmem_in_loop"] == 1 and any("compiler-generated" in v for v in viol)
    # the round-2 fault: a v_mov into a destination right behind its load (block-local form)
    reuse = list(ok)
    reuse.insert(6, "	v_mov_b32_e32 v11, v3")
    assert any("v[11]" in v for v in g.check_kernel_local(reuse)[0])
    assert any("v[11]" in v for v in g.check_kernel(reuse)[0])
    # and the report of the real build exists with zero violations for every hand-scheduled kernel
    rep = os.path.join(repo_root, "profiles", "r03", "asm_guard.txt")
    if os.path.isfile(rep):
        rows = [l for l in open(rep) if l.startswith("_Z")]
        assert len(rows) >= 20 and all(l.rstrip().endswith("violations 0") for l in rows)
        assert sum("gemm_f6v2_kernel" in l and "all-paths" in l for l in rows) == 8      # 5, 7, 9 taps x two output formats + the two ResNet forms
