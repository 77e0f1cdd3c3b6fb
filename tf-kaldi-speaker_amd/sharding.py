"""Multi-GPU = shard by utterance, no exchange step.  The reference does the
same one level up: `utils/split_data.sh $data $nj` + `run.pl JOB=1:nj`
(egs/voxceleb/v1/nnet/run_extract_embeddings.sh:43,68) and `cat xvector.*.scp` (:75-78).

Used by the launcher (run_extract.py: one job per GPU, `lpt_shards` over the frame counts of feats.scp), by
bench.py (config 4: a fixed set of variable-length utterances sharded over WORLD_SIZE ranks) and by the gloo
tests; the only collectives anywhere are the timing barrier / max-over-ranks and, after the timed region, the
ordered gather of the results -- never the data path."""
import time

import numpy as np


def lpt_shards(lengths, world_size):
    """Longest-processing-time-first assignment of utterances to ranks by frame count.
    Returns a list (per rank) of index arrays, each in ascending (input) order."""
    lengths = np.asarray(lengths)
    order = np.argsort(-lengths, kind="stable")
    load = np.zeros(world_size, dtype=np.int64)
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = int(np.argmin(load))
        shards[r].append(int(i))
        load[r] += int(lengths[i])
    return [np.array(sorted(s), dtype=np.int64) for s in shards]


def pack_batches(indices, lengths, batch_frames):
    """Split one rank's utterances into ragged batches of at most `batch_frames` frames
    (at least one utterance each), similar lengths together to keep tiles dense."""
    idx = sorted((int(i) for i in indices), key=lambda i: int(lengths[i]))
    batches, cur, frames = [], [], 0
    for i in idx:
        t = int(lengths[i])
        if cur and frames + t > batch_frames:
            batches.append(cur)
            cur, frames = [], 0
        cur.append(i)
        frames += t
    if cur:
        batches.append(cur)
    return batches


def config4_lengths(n=8192, lo=200, hi=1000, seed=2024):
    """N = 8192 utterances, T ~ integer U[200, 1000], seed 2024 (same on every rank)."""
    return np.random.RandomState(seed).randint(lo, hi + 1, size=n)


def rank_batches(lengths, world_size, rank, batch_frames):
    """This rank's share of a fixed utterance set: -> (indices in input order, ragged batches of global indices)."""
    mine = lpt_shards(lengths, world_size)[rank]
    return mine, pack_batches(mine, lengths, batch_frames)


def gather_in_order(indices, values, n_total, dist=None, dst=0):
    """Ordered concatenation of the per-rank results (what `cat xvector.$j.scp` does for the per-job arks,
    run_extract_embeddings.sh:75-78): `values[i]` belongs to global utterance `indices[i]`.  Returns the
    [n_total, ...] array in input order on rank `dst` (None on the others).  Host-side, outside any timed region."""
    indices = np.asarray(indices, dtype=np.int64)
    values = np.asarray(values)
    parts = [(indices, values)]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        gathered = [None] * dist.get_world_size() if dist.get_rank() == dst else None
        dist.gather_object((indices, values), gathered, dst=dst)
        if dist.get_rank() != dst:
            return None
        parts = gathered
    out = np.empty((n_total,) + values.shape[1:], dtype=values.dtype)
    filled = np.zeros(n_total, dtype=bool)
    for idx, val in parts:
        if np.any(filled[idx]) or len(np.unique(idx)) != len(idx):
            raise ValueError("gather_in_order: an utterance was produced by two ranks")
        out[idx] = val
        filled[idx] = True
    if not filled.all():
        raise ValueError("gather_in_order: %d utterances were produced by no rank" % int((~filled).sum()))
    return out


def timed_steps(step_fn, steps, sync_fn, dist=None, device=None):
    """Time exactly `steps` calls of step_fn between two (barrier + device sync) points and
    return the MAX over ranks, as the bench contract requires."""
    def fence():
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
            dist.barrier()
        sync_fn()

    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    return elapsed
