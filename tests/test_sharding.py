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


def _fake_embed(utts):
    """Deterministic stand-in for the network (content- and length-dependent), one row per utterance."""
    return np.stack([np.concatenate([u.mean(0), u.std(0) + 0.01 * u.shape[0]]) for u in utts]).astype(np.float32)


def _utterance(i, t, dim=5):
    return np.random.RandomState(1000 + i).standard_normal((t, dim)).astype(np.float32)


def _worker(rank, world, port, out):
    """The N>1 host path of bench.py's config 4 with the device work replaced by _fake_embed: shard the fixed
    utterance set (LPT), pack the shard into ragged batches, run K timed passes over them, gather in input order."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lens = sharding.config4_lengths(n=96)
    mine, batches = sharding.rank_batches(lens, world, rank, batch_frames=4000)
    assert sorted(i for b in batches for i in b) == list(mine)
    feats = {int(i): _utterance(int(i), int(lens[i])) for i in mine}
    result = {}

    def one_pass():
        for b in batches:
            emb = _fake_embed([feats[i] for i in b])
            for i, e in zip(b, emb):
                result[i] = e
        time.sleep(0.01 * rank)                      # ranks finish at different times

    elapsed = sharding.timed_steps(one_pass, 3, lambda: None, dist=dist)
    idx = np.array(sorted(result), dtype=np.int64)
    full = sharding.gather_in_order(idx, np.stack([result[int(i)] for i in idx]), len(lens), dist=dist)
    times = [None] * world
    dist.all_gather_object(times, (elapsed, int(lens[mine].sum())))       # test-side check only
    if rank == 0:
        out.put((full, times))
    else:
        assert full is None
    dist.destroy_process_group()


def test_two_rank_gloo_config4_host_path():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    full, times = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    lens = sharding.config4_lengths(n=96)
    assert lens.min() >= 200 and lens.max() <= 1000
    assert np.array_equal(lens[:4], np.random.RandomState(2024).randint(200, 1001, size=8192)[:4])   # a prefix of the full set
    want = _fake_embed([_utterance(i, int(lens[i])) for i in range(96)])
    np.testing.assert_array_equal(full, want)                        # every utterance once, input order
    assert abs(times[0][0] - times[1][0]) < 1e-9                    # both ranks report the max over ranks
    assert times[0][0] >= 3 * 0.01 - 0.002
    loads = [t[1] for t in times]
    assert abs(loads[0] - loads[1]) <= lens.max()                    # LPT balance


def test_gather_in_order_rejects_gaps_and_duplicates():
    import pytest
    with pytest.raises(ValueError):
        sharding.gather_in_order([0, 2], np.zeros((2, 3)), 3)
    with pytest.raises(ValueError):
        sharding.gather_in_order([0, 0, 1], np.zeros((3, 3)), 2)
    out = sharding.gather_in_order([2, 0, 1], np.arange(3, dtype=np.float32)[:, None], 3)
    assert out[:, 0].tolist() == [1.0, 2.0, 0.0]
