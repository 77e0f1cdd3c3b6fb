#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own numpy-only
modules in the build container (never on the GPU box, never at test time):

  /root/reference/model/test_utils.py   compute_self_attention (:321-392), compute_phone_pooling (:1112-1114)
  /root/reference/dataset/kaldi_io.py   write_mat / write_vec_flt / read_mat_ark / read_vec_flt_ark

Only inputs and expected outputs (data) are stored; no reference source travels.
usage: python tests/golden/make_golden.py   (needs /root/reference)
"""
import io
import os
import struct
import sys

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)
from model import test_utils as ref_tu            # noqa: E402
from dataset import kaldi_io as ref_kio           # noqa: E402


class P(object):
    pass


def attention_fixture(seed, b, l, dv, dk, heads, split, use_scale):
    """Reference compute_self_attention.  Its split branch divides ints with `/` (Python-2
    code, TypeError under Python 3), so a split case is produced head by head through the
    no-split branch on the head's key/value slices (identical arithmetic: model/pooling.py:151-157
    splits channels contiguously and scales by the per-head key width)."""
    rs = np.random.RandomState(seed)
    value = rs.standard_normal((b, l, dv))
    value[:, :, 0] = 0.25                                  # a constant channel: exercises the 1e-12 floor
    key = rs.standard_normal((b, l, dk))
    q_dim = dk // heads if split else dk
    query = rs.standard_normal((heads, q_dim))
    p = P()
    p.att_split_value = False
    p.att_split_key = False
    p.att_use_scale = use_scale
    p.att_penalty_term = 0.0
    if not split:
        att, _ = ref_tu.compute_self_attention(value, key, query, p)
    else:
        dvh, dkh = dv // heads, dk // heads
        means, stds = [], []
        for h in range(heads):
            a, _ = ref_tu.compute_self_attention(value[:, :, h * dvh:(h + 1) * dvh], key[:, :, h * dkh:(h + 1) * dkh],
                                                 query[h:h + 1], p)
            means.append(a[:, :dvh])
            stds.append(a[:, dvh:])
        att = np.concatenate(means + stds, axis=1)
    return dict(value=value, key=key, query=query, heads=heads, split=int(split), use_scale=int(use_scale), att=att)


def stat_pool_fixture(seed):
    rs = np.random.RandomState(seed)
    x = rs.standard_normal((3, 17, 12))
    x[:, :, 5] = -1.5                                       # zero variance channel
    post = np.ones((3, 17, 1))
    out = ref_tu.compute_phone_pooling(x, post)[:, :24]     # p1 = [mean, std] (test_utils.py:1114)
    return dict(x=x, mean_std=out)


def encode_cm1(mat):
    """Own encoder of Kaldi's 'CM ' compressed format (per-column header, u8 data) -- only to
    produce a byte stream for the REFERENCE decoder to read."""
    rows, cols = mat.shape
    gmin, gmax = float(mat.min()), float(mat.max())
    grange = max(gmax - gmin, 1e-5)
    buf = io.BytesIO()
    buf.write(struct.pack('<ffii', gmin, grange, rows, cols))
    hdrs, datas = [], []
    for c in range(cols):
        col = np.sort(mat[:, c])
        pct = [col[0], col[rows // 4], col[(3 * rows) // 4], col[-1]]
        u = [int(round((p - gmin) / grange * 65535.0)) for p in pct]
        u[1] = max(u[1], u[0] + 1); u[2] = max(u[2], u[1] + 1); u[3] = max(u[3], u[2] + 1)
        hdrs.append(struct.pack('<4H', *u))
        p0, p25, p75, p100 = [gmin + grange * (1.0 / 65535.0) * v for v in u]
        x = mat[:, c]
        d = np.where(x < p25, (x - p0) / (p25 - p0) * 64.0,
                     np.where(x < p75, 64.0 + (x - p25) / (p75 - p25) * 128.0, 192.0 + (x - p75) / (p100 - p75) * 63.0))
        datas.append(np.clip(np.round(d), 0, 255).astype(np.uint8).tobytes())
    buf.write(b"".join(hdrs))
    buf.write(b"".join(datas))
    return buf.getvalue()


def kaldi_fixture():
    rs = np.random.RandomState(3)
    mats = [("utt_fm", rs.standard_normal((7, 30)).astype(np.float32)),
            ("utt-dm.2", rs.standard_normal((5, 4)).astype(np.float64)),
            ("uttC", (3.0 * rs.standard_normal((40, 6))).astype(np.float32))]
    ark = io.BytesIO()
    ark.mode = 'wb'                                          # reference asserts fd.mode == 'wb'
    ref_kio.write_mat(ark, mats[0][1], key=mats[0][0])
    ref_kio.write_mat(ark, mats[1][1], key=mats[1][0])
    ark.write((mats[2][0] + ' ').encode() + b'\0B' + b'CM ' + encode_cm1(mats[2][1]))
    data = ark.getvalue()
    with open(os.path.join(HERE, "feats.ark"), "wb") as f:
        f.write(data)
    decoded = {k: m for k, m in ref_kio.read_mat_ark(io.BytesIO(data))}
    np.savez(os.path.join(HERE, "feats_expected.npz"), **{"m%d" % i: decoded[k] for i, (k, _) in enumerate(mats)},
             keys=np.array([k for k, _ in mats]))
    vecs = [("spk1-utt1", rs.standard_normal(512).astype(np.float32)), ("x", rs.standard_normal(3).astype(np.float64))]
    vark = io.BytesIO()
    vark.mode = 'wb'
    for k, v in vecs:
        ref_kio.write_vec_flt(vark, v, key=k)
    with open(os.path.join(HERE, "vectors.ark"), "wb") as f:
        f.write(vark.getvalue())
    np.savez(os.path.join(HERE, "vectors_expected.npz"), v0=vecs[0][1], v1=vecs[1][1], keys=np.array([k for k, _ in vecs]))


def main():
    cases = [attention_fixture(1, 2, 9, 8, 6, 1, False, True), attention_fixture(2, 2, 11, 6, 8, 3, False, False),
             attention_fixture(3, 3, 7, 12, 8, 4, True, True), attention_fixture(4, 1, 5, 4, 4, 1, True, False)]
    for i, c in enumerate(cases):
        np.savez(os.path.join(HERE, "attention_%d.npz" % i), **c)
    np.savez(os.path.join(HERE, "stat_pool.npz"), **stat_pool_fixture(7))
    kaldi_fixture()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
