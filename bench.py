#!/usr/bin/env python
"""bench.py -- x-vector extraction throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (TDNN frame layers -> statistics pooling -> segment
affine, node tdnn6_dense) over one batch of 256 synthetic 30-dim x 300-frame utterances
that are already resident in HBM (BASELINE.json configs[1]).  With --gpus N each rank runs
the same per-GPU batch on its own GPU (utterances shard with no exchange step -> weak
scaling, no collective on the data path; torch.distributed is used for the timing barrier
and the max-over-ranks only).

Prints ONE JSON line on rank 0 (contract in the task statement), carrying
  roofline     : dominant kernel (tdnn3_conv GEMM), algorithmic FLOPs / hipEvent-measured
                 mean launch duration inside the timed region, against the dense MFMA peak;
  cpu_baseline : oracle/ref_torch.py (fp32 torch-CPU restatement, 1 thread, batch 1 --
                 mirrors model/trainer.py:135-139 + extract.py:89) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FLOP_PER_UTT_300 = 2452865024          # BASELINE.md section 2 (L1..L6, T=300)
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0}   # MI355X_MICROARCH.md: fp32 MFMA / bf16 MFMA dense
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--dim", type=int, default=30)
    ap.add_argument("--precision", default=os.environ.get("XVEC_PRECISION", ""), help="f32 | bf16x3 (default: library default)")
    ap.add_argument("--pooling", default="statistics_pooling", choices=["statistics_pooling", "self_attention"])
    ap.add_argument("--network", default="tdnn", choices=["tdnn", "extended_tdnn", "resnet_18"],
                    help="tdnn = BASELINE configs 1-4; resnet_18 = config 5 (use --dim 40 --batch 64)")
    ap.add_argument("--varlen", action="store_true", help="config 4: T ~ U[200,1000] (seed 2024)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward from a captured hipGraph (per-kernel events are then not recorded)")
    return ap.parse_args()


def pmc_traffic(kernel, precision, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (profiles/<round>/traffic.json; FETCH_SIZE/WRITE_SIZE need their own profiler runs, so
    they cannot be sampled inside the timed region).  None when no matching profile exists."""
    if (precision != "bf16x3" or args.varlen or args.pooling != "statistics_pooling" or args.batch != 256 or
            args.frames != 300 or args.network != "tdnn"):
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic.json")), reverse=True):
        try:
            with open(path) as f:
                k = json.load(f)["kernels"].get(kernel)
            if k:
                return k["traffic_bytes"]
        except (OSError, ValueError, KeyError):
            continue
    return None


def cpu_baseline(weights, params, dim, frames, budget_s):
    """oracle/ref_torch.py on this box's host cores: 1 thread, 1 utterance per call."""
    import torch
    from oracle import ref_torch
    from tf_kaldi_speaker_amd import synth
    prev = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        model = (ref_torch.TorchResnet18 if params.network_type == "resnet_18" else ref_torch.TorchTdnn)(weights, params)
        utts = synth.synth_features(4, frames, dim, seed=99)
        model.predict(utts[0], dim)                       # warm-up
        n, t0 = 0, time.perf_counter()
        while True:
            model.predict(utts[n % len(utts)], dim)
            n += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n >= 2000:
                break
    finally:
        torch.set_num_threads(prev)
    return {"value": round(n / el, 3), "unit": "utterances/s", "cores": 1, "kind": "port",
            "sample": "%d utterances of %dx%d, oracle/ref_torch.py fp32, 1 thread, batch 1, %.1f s" % (n, frames, dim, el)}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist
    from tf_kaldi_speaker_amd import synth
    from tf_kaldi_speaker_amd.params import Params
    from tf_kaldi_speaker_amd import trainer as trainer_mod
    from tf_kaldi_speaker_amd.trainer import Trainer

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("[bench] note: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run for N>1" % (args.gpus, world),
              file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    precision = args.precision or trainer_mod.DEFAULT_PRECISION

    base = synth.TDNN_ATT_PARAMS if args.pooling == "self_attention" else synth.TDNN_STAT_PARAMS
    if args.network == "resnet_18":
        if args.dim != 40:
            args.dim = 40
        params = Params(**dict(synth.RESNET_PARAMS))
        weights = synth.synth_resnet_weights(params, seed=0)
    else:
        params = Params(**dict(base, network_type=args.network))
        if args.network == "extended_tdnn":
            params.embedding_node = "tdnn12_dense"
            if args.pooling == "self_attention":
                params.att_key_input, params.att_value_input = "tdnn9_relu", "tdnn10_relu"
        weights = synth.synth_weights(params, args.dim, seed=0)
    tr = Trainer(params, None, args.dim, single_cpu=True, device=local_rank, precision=precision)
    tr.build("predict")
    tr.load_weights(weights)

    # synthetic batch, distinct per rank, resident in HBM before the timed region
    if args.varlen:
        lens = np.random.RandomState(2024 + rank).randint(200, 1001, size=args.batch)
    else:
        lens = np.full(args.batch, args.frames)
    utts = synth.synth_features(args.batch, lens, args.dim, seed=1234 + rank)
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    feats = torch.from_numpy(np.concatenate(utts, axis=0)).to(dev)
    info = tr.plan_info(offsets)
    out = torch.empty((int(info["out_rows"]), int(info["out_cols"])), dtype=torch.float32, device=dev)

    from tf_kaldi_speaker_amd import sharding

    def sync_dev():
        torch.cuda.synchronize(dev)

    graph = None
    if args.graph:
        graph, _ = tr.capture_graph(feats, offsets, out=out)
        args.no_profile = True

    def step():
        if graph is not None:
            graph.replay()
        else:
            tr.predict_packed(feats, offsets, out=out)

    for _ in range(args.warmup):
        step()
    sync_dev()
    if not args.no_profile:
        tr.profile_begin(max_events=2 * 48 * (args.steps + 1))
    # barrier + synchronize, exactly K steps, barrier + synchronize, max over ranks
    elapsed = sharding.timed_steps(step, args.steps, sync_dev, dist=dist if world > 1 else None, device=dev)
    kernels = []
    if not args.no_profile:
        kernels, _ = tr.profile_end()
    # extra leg (not `value`): the same K steps replayed from a captured hipGraph
    graph_rate = None
    if graph is None and not args.varlen:
        g2, _ = tr.capture_graph(feats, offsets, out=out)
        for _ in range(args.warmup):
            g2.replay()
        el2 = sharding.timed_steps(g2.replay, args.steps, sync_dev, dist=dist if world > 1 else None, device=dev)
        graph_rate = n_gpus * args.batch * args.steps / el2
    result = None
    if rank == 0:
        emb = out.cpu().numpy()
        total_utts = n_gpus * args.batch * args.steps
        value = total_utts / elapsed
        # parity spot check against the float64 oracle on a few utterances of this batch
        from oracle import ref_numpy
        idx = list(range(0, args.batch, max(1, args.batch // 4)))[:(2 if args.network == "resnet_18" else 4)]
        errs = []
        for i in idx:
            ref = ref_numpy.predict(utts[i], weights, params, args.dim)
            errs.append(float(np.linalg.norm(emb[i] - ref) / np.linalg.norm(ref)))
        # roofline of the dominant kernel
        roof = None
        if kernels:
            dom = max(kernels, key=lambda k: k["ms"])
            tf = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
            peak = PEAK_TFLOPS[precision]
            roof = {"kernel": dom["name"], "bound": "mfma", "achieved": round(tf, 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(tf / peak, 4), "traffic": pmc_traffic(dom["name"], precision, args),
                    "launch_ms": round(dom["ms"], 4),
                    "hbm_frac_algorithmic": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
            roof["algorithmic_bytes"] = dom["bytes"]
            if precision == "bf16x3":
                roof["mfma_issue_frac"] = round(3 * tf / peak, 4)
                roof["note"] = ("bf16x3: 3 bf16 MFMAs per algorithmic product, so the ceiling of `frac` is 1/3; "
                                "mfma_issue_frac = issued MFMA FLOPs / bf16 dense peak")
        flops_step = float(info["flops"])
        result = {
            "metric": "utterances/sec x-vector extraction (30-dim x 300-frame)",
            "value": round(value, 1), "unit": "utterances/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if precision == "f32" else "bf16x3(f32-split)", "data": "synthetic",
            "config": {"workload": "%s x-vector (%s), %s, %d utt/GPU/step of %s frames x %d dims"
                       % (args.network, params.embedding_node, args.pooling, args.batch,
                          "U[200,1000]" if args.varlen else str(args.frames), args.dim),
                       "batch_per_gpu": args.batch, "frames": "varlen" if args.varlen else args.frames,
                       "node": params.embedding_node, "precision": precision, "parallelism": "utterance-shard x%d" % n_gpus,
                       "launch": "hipGraph replay" if args.graph else "stream launches"},
            "tflops_algorithmic": round(flops_step * args.steps * n_gpus / elapsed / 1e12, 2),
            "graph_replay_value": round(graph_rate, 1) if graph_rate else None,
            "parity_rel_l2_max": max(errs),
            "kernels": [{"name": k["name"], "ms": round(k["ms"], 4),
                         "tflops": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2) if k["ms"] > 0 else None,
                         "gbs_algorithmic": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1) if k["ms"] > 0 else None}
                        for k in kernels],
            "roofline": roof,
        }
        if args.cpu_seconds > 0 and n_gpus == 1:
            result["cpu_baseline"] = cpu_baseline(weights, params, args.dim, args.frames, args.cpu_seconds)
        else:
            result["cpu_baseline"] = None
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


if __name__ == "__main__":
    main()
