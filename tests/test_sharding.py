"""CPU, world_size 2, gloo: the N>1 host path (shard by utterance, no data-path collective,
barrier + max-over-ranks timing) used by bench.py."""
import os
import socket
import time

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tf_kaldi_speaker_amd import sharding


def test_lpt_shards_cover_and_balance():
    lens = np.random.RandomState(2024).randint(200, 1001, size=512)
    for world in (1, 2, 4, 8):
        shards = sharding.lpt_shards(lens, world)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(512))               # every utterance exactly once
        loads = np.array([lens[s].sum() for s in shards])
        assert loads.max() - loads.min() <= lens.max()              # LPT bound
        for s in shards:
            assert np.all(np.diff(s) > 0)                           # input order kept inside a shard


def test_pack_batches_respects_budget():
    lens = np.random.RandomState(1).randint(25, 1200, size=100)
    batches = sharding.pack_batches(range(100), lens, 4000)
    assert sorted(i for b in batches for i in b) == list(range(100))
    for b in batches:
        assert len(b) == 1 or sum(lens[i] for i in b) <= 4000


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = np.random.RandomState(7).randint(200, 1001, size=64)
    mine = sharding.lpt_shards(lens, world)[rank]
    # stand-in workload: one "embedding" per utterance, rank 1 is slower
    emb = {int(i): np.full(4, float(lens[i]), np.float32) for i in mine}
    elapsed = sharding.timed_steps(lambda: time.sleep(0.02 * (rank + 1)), 3, lambda: None, dist=dist)
    gathered = [None] * world
    dist.all_gather_object(gathered, (sorted(emb), elapsed))        # test-side check only
    if rank == 0:
        keys = sorted(k for g in gathered for k in g[0])
        out.put((keys, [g[1] for g in gathered]))
    dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_max_timing():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    keys, times = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert keys == list(range(64))                                  # shards are a partition
    assert abs(times[0] - times[1]) < 1e-9                          # both report the max over ranks
    assert times[0] >= 3 * 0.04 - 0.005                             # = the slower rank's time
