"""Stages 2-3 of the extraction recipe (egs/voxceleb/v1/nnet/run_extract_embeddings.sh:80-103) without the Kaldi
binaries: speaker-level means of the utterance x-vectors (`ivector-mean ark:$data/spk2utt ...`) and length
normalisation (`ivector-normalize-length --scaleup=false`), computed on the GPU by csrc/post.hip through the C ABI
(xv_length_normalize / xv_speaker_mean).  Tables are read and written natively (native_ark).  No CPU path."""
import ctypes as C
import os
import shutil

import numpy as np

from . import _lib
from . import native_ark


def read_vectors(rspecifier):
    """-> (keys, [n, dim] float32) of a float-vector table (`scp:file` or `ark:file`), table order."""
    reader = native_ark.ArkBatchReader(rspecifier, batch_frames=1 << 21, max_utts=4096, capacity=(1 << 21) + (1 << 16))
    keys, parts, dim = [], [], None
    try:
        for k, offsets, data in reader:
            d = int(offsets[1] - offsets[0])
            if dim is None:
                dim = d
            if np.any(np.diff(offsets) != dim):
                raise ValueError("%s: vectors of different dimensions" % rspecifier)
            keys.extend(k)
            parts.append(data.reshape(len(k), dim).copy())
    finally:
        reader.close()
    if not parts:
        return [], np.zeros((0, 0), np.float32)
    return keys, np.concatenate(parts, axis=0)


def read_spk2utt(path):
    """[(spk, [utt, ...])] in file order (Kaldi `ark:spk2utt` text table)."""
    out = []
    with open(path) as f:
        for line in f:
            p = line.split()
            if p:
                out.append((p[0], p[1:]))
    return out


def _device_tensor(x, device):
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no HIP device visible: the post-step has no CPU fallback")
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to("cuda:%d" % device)


def length_normalize(x, scaleup=False, device=0):
    """`ivector-normalize-length [--scaleup]` on an [n, dim] float32 array -> float32 array (GPU)."""
    import torch
    lib = _lib.load()
    x = np.asarray(x, dtype=np.float32)
    if x.shape[0] == 0:
        return x.copy()
    with torch.cuda.device(device):
        d = _device_tensor(x, device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.xv_length_normalize(device, C.c_void_p(d.data_ptr()), x.shape[1], x.shape[0], x.shape[1],
                                           int(bool(scaleup)), C.c_void_p(d.data_ptr()), x.shape[1], C.c_void_p(stream)))
        return d.cpu().numpy()


def speaker_mean(keys, x, spk2utt, device=0):
    """`ivector-mean ark:spk2utt`: -> (speakers, [S, dim] float32 means, counts) for the speakers with at least one
    vector among `keys`; utterances of spk2utt that have no vector are skipped (Kaldi warns and continues)."""
    import torch
    lib = _lib.load()
    x = np.asarray(x, dtype=np.float32)
    row = {k: i for i, k in enumerate(keys)}
    spks, off, idx = [], [0], []
    for spk, utts in spk2utt:
        rows = [row[u] for u in utts if u in row]
        if not rows:
            continue
        spks.append(spk)
        idx.extend(rows)
        off.append(len(idx))
    counts = np.diff(np.asarray(off, dtype=np.int64))
    if not spks:
        return [], np.zeros((0, x.shape[1] if x.ndim == 2 else 0), np.float32), counts
    with torch.cuda.device(device):
        d = _device_tensor(x, device)
        d_off = torch.from_numpy(np.asarray(off, dtype=np.int32)).to(d.device)
        d_idx = torch.from_numpy(np.asarray(idx, dtype=np.int32)).to(d.device)
        out = torch.empty((len(spks), x.shape[1]), dtype=torch.float32, device=d.device)
        stream = torch.cuda.current_stream(device).cuda_stream
        _lib.check(lib.xv_speaker_mean(device, C.c_void_p(d.data_ptr()), x.shape[1], x.shape[1], C.c_void_p(d_off.data_ptr()),
                                       C.c_void_p(d_idx.data_ptr()), len(spks), C.c_void_p(out.data_ptr()), x.shape[1],
                                       C.c_void_p(stream)))
        return spks, out.cpu().numpy(), counts


def write_vectors(wspecifier, keys, x):
    w = native_ark.VectorWriter(wspecifier)
    for i in range(0, len(keys), 4096):
        w.write(keys[i:i + 4096], x[i:i + 4096])
    return w.close()


def stage_speaker_mean(data_dir, out_dir, normalize, device=0):
    """run_extract_embeddings.sh:80-95 (stage 2): spk_xvector.{ark,scp} + num_utts.ark from xvector.scp and
    $data/spk2utt; with `normalize`, vectors are length-normalised before the mean and the means again after it."""
    keys, x = read_vectors("scp:" + os.path.join(out_dir, "xvector.scp"))
    if normalize:
        x = length_normalize(x, False, device)
    spks, means, counts = speaker_mean(keys, x, read_spk2utt(os.path.join(data_dir, "spk2utt")), device)
    if normalize:
        means = length_normalize(means, False, device)
    write_vectors("ark,scp:%s,%s" % (os.path.join(out_dir, "spk_xvector.ark"), os.path.join(out_dir, "spk_xvector.scp")),
                  spks, means)
    with open(os.path.join(out_dir, "num_utts.ark"), "w") as f:         # `ark,t:` int32 table
        for s, n in zip(spks, counts):
            f.write("%s %d\n" % (s, int(n)))
    return len(spks)


def stage_length_norm(out_dir, device=0):
    """run_extract_embeddings.sh:97-103 (stage 3, only with --normalize true): xvector.scp -> xvector_before_norm.scp,
    length-normalised copies in xvector.ark / xvector.scp."""
    scp = os.path.join(out_dir, "xvector.scp")
    before = os.path.join(out_dir, "xvector_before_norm.scp")
    shutil.copyfile(scp, before)
    keys, x = read_vectors("scp:" + before)
    x = length_normalize(x, False, device)
    write_vectors("ark,scp:%s,%s" % (os.path.join(out_dir, "xvector.ark"), scp), keys, x)
    return len(keys)
