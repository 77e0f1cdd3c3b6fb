"""Multi-GPU = shard by utterance, no exchange step.  The reference does the
same one level up: `utils/split_data.sh $data $nj` + `run.pl JOB=1:nj`
(egs/voxceleb/v1/nnet/run_extract_embeddings.sh:43,68) and `cat xvector.*.scp` (:75-78).

Helpers used by bench.py and by the gloo tests; the only collective anywhere is the timing
barrier / max-over-ranks, never the data path."""
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
