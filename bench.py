#!/usr/bin/env python
"""bench.py -- x-vector extraction throughput on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path (TDNN frame layers -> statistics pooling -> segment
affine, node tdnn6_dense) over one batch of 256 synthetic 30-dim x 300-frame utterances
that are already resident in HBM (BASELINE.json configs[1]).  With --gpus N each rank runs
the same per-GPU batch on its own GPU (utterances shard with no exchange step -> weak
scaling, no collective on the data path; torch.distributed is used for the timing barrier
and the max-over-ranks only).

Prints ONE JSON line on rank 0 (contract in the task statement).  `value` is the config-2 rate above; beside it
  roofline     : dominant kernel (tdnn3_conv GEMM), algorithmic FLOPs / hipEvent-measured mean launch duration
                 over a second pass of the same K steps (the events cost 6-9 % of a step, so the region `value`
                 is taken from carries none), against the dense MFMA peak;
  cpu_baseline : the CPU stand-in of the reference's extraction job (oracle/cpu_extract.py: ark -> ark,
                 1 thread, batch 1) run as nj = min(host cores, 32) fresh processes side by side
                 (run_extract_embeddings.sh:3,68) for a bounded time; N = 1 only;
  value_reps   : min / median / max of 5 more repetitions of the same K-step timed region;
  e2e_value    : host numpy -> pinned staging -> H2D -> kernels -> D2H with two batches in flight (Trainer.submit_list / collect);
                 e2e_sync_value: one blocking Trainer.predict_list per batch; N = 1 only;
  cli          : ark -> ark through the command-line driver in a fresh process, N = 1 only;
  launcher     : the config-4 set through run_extract.py (the run_extract_embeddings.sh counterpart), nj = 1, wall clock incl.
                 process start-up, N = 1 only;
  config4      : BASELINE configs[3] / SURVEY 8(d) config 4 -- a FIXED set of 8192 utterances, T ~ U[200,1000]
                 (seed 2024), LPT-sharded over WORLD_SIZE ranks, resident in HBM, ragged batches; utt/s and
                 frames/s of the whole set (strong scaling), frames per rank.  Runs at every N.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

FLOP_PER_UTT_300 = 2452865024          # BASELINE.md section 2 (L1..L6, T=300)
PEAK_TFLOPS = {"f32": 157.3, "bf16x3": 2500.0, "f16x3": 2500.0, "f16f6": 2500.0}   # MI355X_MICROARCH.md: fp32 MFMA / bf16 = f16 MFMA dense
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="utterances per GPU per step")
    ap.add_argument("--frames", type=int, default=300)
    ap.add_argument("--dim", type=int, default=30)
    ap.add_argument("--precision", default=os.environ.get("XVEC_PRECISION", ""), help="f32 | bf16x3 | f16x3 | f16f6 (default: library default)")
    ap.add_argument("--pooling", default="statistics_pooling", choices=["statistics_pooling", "self_attention"])
    ap.add_argument("--network", default="tdnn", choices=["tdnn", "extended_tdnn", "resnet_18"],
                    help="tdnn = BASELINE configs 1-4; resnet_18 = config 5 (use --dim 40 --batch 64)")
    ap.add_argument("--varlen", action="store_true", help="T ~ U[200,1000] per-GPU batch (config-4 shape as the main workload)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--cpu-nj", type=int, default=0, help="processes of the cpu_baseline leg (0 = min(host cores, 32))")
    ap.add_argument("--no-profile", action="store_true", help="skip per-kernel hipEvent timing")
    ap.add_argument("--no-extra", action="store_true", help="only the main timed region (no reps / e2e / cli / config4 legs)")
    ap.add_argument("--prewarm", type=float, default=0.5, help="seconds of untimed forwards before the warm-up steps (clock ramp)")
    ap.add_argument("--reps", type=int, default=5, help="extra repetitions of the timed region (value_reps)")
    ap.add_argument("--c4-utts", type=int, default=8192, help="config 4: size of the fixed utterance set")
    ap.add_argument("--c4-steps", type=int, default=3, help="config 4: timed passes over the set")
    ap.add_argument("--c4-batch-frames", type=int, default=153600, help="config 4: frames per ragged device batch")
    ap.add_argument("--cli-utts", type=int, default=131072, help="cli leg: utterances in the ark")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend of the timing barrier (nccl = RCCL; gloo only to rehearse N > 1 on one GPU)")
    ap.add_argument("--one-device", action="store_true", help="rehearsal: every rank uses device 0 (with --dist-backend gloo)")
    ap.add_argument("--graph", action="store_true",
                    help="replay the forward from a captured hipGraph (per-kernel events are then not recorded)")
    return ap.parse_args()


def pmc_traffic(kernel, precision, args):
    """HBM-side bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same
    command (profiles/<round>/traffic.json; FETCH_SIZE/WRITE_SIZE need their own profiler runs, so
    they cannot be sampled inside the timed region).  None when no matching profile exists."""
    if (precision == "f32" or args.varlen or args.pooling != "statistics_pooling" or args.batch != 256 or
            args.frames != 300 or args.network != "tdnn"):
        return None
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic*.json")), reverse=True):
        try:
            with open(path) as f:
                doc = json.load(f)
            if precision not in doc.get("config", ""):
                continue
            k = doc["kernels"].get(kernel)
            if k:
                return k["traffic_bytes"]
        except (OSError, ValueError, KeyError):
            continue
    return None


# ------------------------------------------------------------------------------------------ CPU baseline
def host_cores():
    """Cores this process may use: scheduler affinity, capped by the cgroup CPU quota when one is set."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            with open(path) as f:
                parts = f.read().split()
            if path.endswith("cpu.max"):
                if parts[0] != "max":
                    n = min(n, max(1, int(int(parts[0]) / int(parts[1]))))
            else:
                quota = int(parts[0])
                if quota > 0:
                    with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                        n = min(n, max(1, quota // int(g.read().split()[0])))
        except (OSError, ValueError, IndexError):
            continue
    return max(n, 1)


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def write_feature_ark(path, n, frames, dim, seed=0):
    """n utterances of [frames, dim] float32 as one binary Kaldi matrix ark (windows of one random matrix)."""
    from tf_kaldi_speaker_amd import kaldi_io
    base = np.random.RandomState(seed).standard_normal((frames + 64, dim)).astype(np.float32)
    with open(path, "wb") as f:
        for i in range(n):
            kaldi_io.write_mat(f, base[i % 64:i % 64 + frames], key="utt%07d" % i)


def cpu_baseline(weights, params, dim, frames, budget_s, nj):
    """nj fresh single-thread processes of oracle/cpu_extract.py (ark -> ark, batch 1), each for `budget_s` seconds
    of extraction; all read the same feature ark (a job's CPU time does not depend on which split it reads)."""
    from tf_kaldi_speaker_amd import model_io
    nj = nj if nj > 0 else min(host_cores(), 32)           # 32 = the reference's default nj (run_extract_embeddings.sh:3)
    tmp = tempfile.mkdtemp(prefix="xvcpu_", dir="/tmp")
    try:
        model_dir = os.path.join(tmp, "exp")
        model_io.save_model(model_dir, dict(params.dict), dim, weights, step=1)
        ark = os.path.join(tmp, "feats.ark")
        per_job = int(max(64, budget_s * (300.0 if params.network_type != "resnet_18" else 12.0) * 300.0 / max(frames, 1)))
        write_feature_ark(ark, per_job, frames, dim, seed=99)
        env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        procs = []
        t0 = time.perf_counter()
        for j in range(nj):
            cmd = [sys.executable, os.path.join(ROOT, "oracle", "cpu_extract.py"), "--max-seconds", str(budget_s),
                   "--min-chunk-size", "25", model_dir, "ark:" + ark, "ark:" + os.path.join(tmp, "xvector.%d.ark" % (j + 1))]
            procs.append(subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, text=True))
        recs = []
        for p in procs:
            out, err = p.communicate(timeout=budget_s * 4 + 600)
            if p.returncode != 0:
                raise RuntimeError("cpu_extract failed: %s" % err[-500:])
            recs.append(json.loads(out.strip().splitlines()[-1]))
        wall = time.perf_counter() - t0
    finally:
        subprocess.call(["rm", "-rf", tmp])
    n = sum(r["utterances"] for r in recs)
    el = max(r["seconds"] for r in recs)
    total = n / el
    return {"value": round(total, 2), "unit": "utterances/s", "cores": nj, "kind": "port",
            "value_total": round(total, 2), "value_per_core": round(total / nj, 3), "cpu_model": cpu_model(),
            "host_cores": host_cores(), "wall_s": round(wall, 1),
            "sample": "%d processes x 1 thread of oracle/cpu_extract.py (fp32 torch-CPU restatement, ark->ark, batch 1): %d "
                      "utterances of %dx%d in %.1f s of extraction each (process start-up excluded)" % (nj, n, frames, dim, el)}


# ------------------------------------------------------------------------------------------ extra legs
def cli_leg(weights, params, dim, frames, n_utts, precision):
    """ark -> ark through `python -m tf_kaldi_speaker_amd.extract` in a fresh process (ark parsing, H2D, kernels, D2H,
    vector-ark formatting); the ark sits in the page cache."""
    from tf_kaldi_speaker_amd import model_io
    tmp = tempfile.mkdtemp(prefix="xvcli_", dir="/tmp")
    try:
        model_dir = os.path.join(tmp, "exp")
        model_io.save_model(model_dir, dict(params.dict), dim, weights, step=1)
        ark = os.path.join(tmp, "feats.ark")
        write_feature_ark(ark, n_utts, frames, dim, seed=5)
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        cmd = [sys.executable, "-m", "tf_kaldi_speaker_amd.extract", "--gpu", "0", "--precision", precision, model_dir,
               "ark:" + ark, "ark:" + os.path.join(tmp, "xvector.ark")]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": r.stderr[-300:]}
        loop, waits = None, None
        for line in r.stderr.splitlines():
            if line.startswith("Extracted ") and " in " in line:
                loop = float(line.rsplit(" in ", 1)[1].split()[0])
            if "driver loop:" in line:
                waits = line.split("driver loop:", 1)[1].strip()
        size = os.path.getsize(os.path.join(tmp, "xvector.ark"))
    finally:
        subprocess.call(["rm", "-rf", tmp])
    return {"utterances": n_utts, "wall_s": round(wall, 2), "loop_s": loop,
            "value": round(n_utts / loop, 1) if loop else None, "value_incl_startup": round(n_utts / wall, 1),
            "unit": "utterances/s", "out_bytes": size, "loop_waits": waits,
            "note": "value = utterances / time of the driver's read-embed-write loop; value_incl_startup adds process "
                    "start-up, model upload and the first-touch of the GPU"}


def launcher_leg(weights, params, dim, n_utts, precision):
    """The config-4 set (T ~ U[200,1000], seed 2024) through the LAUNCHER -- `run_extract.py` = run_extract_embeddings.sh:
    stage 0 (shards, one fresh extraction process per job), stage 1 (ordered scp merge), stages 2-3 (speaker means, length norm) --
    with nj = one job on this GPU, wall clock of the whole command incl. both process start-ups: what a job costs before an
    8-GPU node ever runs eight of them side by side."""
    from tf_kaldi_speaker_amd import kaldi_io, model_io, sharding
    tmp = tempfile.mkdtemp(prefix="xvlaunch_", dir="/tmp")
    try:
        model_dir = os.path.join(tmp, "exp")
        model_io.save_model(model_dir, dict(params.dict), dim, weights, step=1)
        data = os.path.join(tmp, "data")
        os.makedirs(data)
        lens = sharding.config4_lengths(n=n_utts)
        base = np.random.RandomState(7).standard_normal((int(lens.max()) + 64, dim)).astype(np.float32)
        ark = os.path.join(data, "feats.ark")
        with open(ark, "wb") as f, open(os.path.join(data, "feats.scp"), "w") as scp, \
                open(os.path.join(data, "utt2num_frames"), "w") as u2n, open(os.path.join(data, "spk2utt"), "w") as s2u:
            spk = {}
            for i, t in enumerate(lens):
                key = "spk%03d-utt%06d" % (i % 128, i)
                f.write((key + " ").encode())
                scp.write("%s %s:%d\n" % (key, ark, f.tell()))
                kaldi_io.write_mat(f, base[i % 64:i % 64 + int(t)], key="")
                u2n.write("%s %d\n" % (key, int(t)))
                spk.setdefault("spk%03d" % (i % 128), []).append(key)
            for k in sorted(spk):
                s2u.write("%s %s\n" % (k, " ".join(spk[k])))
        env = dict(os.environ)
        env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
        out = os.path.join(tmp, "xv")
        cmd = [sys.executable, "-m", "tf_kaldi_speaker_amd.run_extract", "--nj", "1", "--gpus", "0", "--apply-vad", "false",
               "--cmn-window", "0", "--min-chunk-size", "25", "--node", params.embedding_node, "--precision", precision,
               "--batch-frames", "153600", model_dir, data, out]
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"error": (r.stdout + r.stderr)[-400:]}
        loop = None
        try:
            for line in open(os.path.join(out, "log", "extract.1.log")):
                if line.startswith("Extracted ") and " in " in line:
                    loop = float(line.rsplit(" in ", 1)[1].split()[0])
        except OSError:
            pass
        n_out = sum(1 for _ in open(os.path.join(out, "xvector.scp")))
    finally:
        subprocess.call(["rm", "-rf", tmp])
    frames = int(lens.sum())
    return {"workload": "%d utterances, T ~ U[200,1000] seed 2024, through run_extract.py (nj 1, one GPU, stages 0-3)" % n_utts,
            "vectors": n_out, "wall_s": round(wall, 2), "job_loop_s": loop,
            "value_incl_startup": round(n_utts / wall, 1), "frames_per_s_incl_startup": round(frames / wall, 1),
            "value_job_loop": round(n_utts / loop, 1) if loop else None,
            "frames_per_s_job_loop": round(frames / loop, 1) if loop else None, "unit": "utterances/s",
            "note": "wall = launcher + job process start-up, model upload, extraction loop, merge and the two post stages"}


def config4_leg(tr, args, dist, world, rank, dev, weights, params):
    """the fixed 8192-utterance variable-length set, sharded by utterance (LPT on frames)."""
    import torch
    from tf_kaldi_speaker_amd import sharding
    lens = sharding.config4_lengths(n=args.c4_utts)
    mine, batches = sharding.rank_batches(lens, world, rank, args.c4_batch_frames)
    feats, offs, outs, host0 = [], [], [], None
    for bi, b in enumerate(batches):
        bl = lens[b]
        off = np.concatenate([[0], np.cumsum(bl)]).astype(np.int32)
        x = np.random.RandomState(50000 + int(b[0])).standard_normal((int(off[-1]), args.dim)).astype(np.float32)
        if bi == 0:
            host0 = x
        feats.append(torch.from_numpy(x).to(dev))
        offs.append(off)
        info = tr.plan_info(off)
        outs.append(torch.empty((int(info["out_rows"]), int(info["out_cols"])), dtype=torch.float32, device=dev))

    def one_pass():
        for x, off, o in zip(feats, offs, outs):
            tr.predict_packed(x, off, out=o)

    def sync_dev():
        torch.cuda.synchronize(dev)

    one_pass()                                         # warm-up: every geometry planned, every kernel loaded
    sync_dev()
    d = dist if world > 1 else None
    elapsed = sharding.timed_steps(one_pass, args.c4_steps, sync_dev, dist=d, device=dev if args.dist_backend == "nccl" else None)
    emb = torch.cat(outs, dim=0).cpu().numpy()
    order = np.array([i for b in batches for i in b], dtype=np.int64)
    full = sharding.gather_in_order(order, emb, len(lens), dist=d)
    loads = [int(lens[mine].sum())]
    if world > 1:
        got = [None] * world
        dist.all_gather_object(got, loads[0])
        loads = got
    if rank != 0:
        return None
    from oracle import ref_numpy
    errs = []
    b0, off0 = batches[0], offs[0]
    for j in (0, len(b0) // 2, len(b0) - 1):
        ref = ref_numpy.predict(host0[off0[j]:off0[j + 1]], weights, params, args.dim)
        got = full[b0[j]]
        errs.append(float(np.linalg.norm(got - ref) / np.linalg.norm(ref)))
    total_frames = int(lens.sum())
    return {"workload": "%d utterances, T ~ U[200,1000] seed 2024, %d dims, LPT-sharded over %d GPU(s), ragged batches of <= %d frames"
                        % (len(lens), args.dim, world, args.c4_batch_frames),
            "value": round(len(lens) * args.c4_steps / elapsed, 1), "unit": "utterances/s",
            "frames_per_s": round(total_frames * args.c4_steps / elapsed, 1), "scaling": "strong", "n_gpus": world,
            "steps": args.c4_steps, "ms_per_pass": round(elapsed / args.c4_steps * 1e3, 3),
            "frames_per_rank": loads, "batches_rank0": len(batches), "parity_rel_l2_max": max(errs),
            "checksum": float(np.sum(full.astype(np.float64)))}


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
        if args.one_device:
            local_rank = 0
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    n_gpus = world if world > 1 else 1
    if args.gpus != n_gpus and rank == 0:
        print("[bench] note: --gpus %d but WORLD_SIZE=%d; launch with torch.distributed.run for N>1" % (args.gpus, world),
              file=sys.stderr)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # The bench runs what a default invocation of the library, the CLI and the launcher runs (trainer.default_precision: f16f6);
    # --precision / XVEC_PRECISION override.  The default line also carries the rate of the full-range precision (value_bf16x3).
    precision = args.precision or trainer_mod.default_precision(args.network)

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
    d = dist if world > 1 else None
    red_dev = dev if args.dist_backend == "nccl" else None     # where the max-over-ranks all_reduce lives

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

    # clock ramp: the first K steps after an idle period run ~5 % slower than any later repetition of the same region
    # (value_reps), so the device is kept busy for a moment before the W warm-up steps (untimed, reported as prewarm_s)
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm:
        for _ in range(10):
            step()
        sync_dev()
    for _ in range(args.warmup):
        step()
    sync_dev()
    # `value` comes from a region WITHOUT hipEvents: an event pair around even one layer keeps the next kernel from
    # starting while the previous one drains (bracketing only tdnn3_conv cost 5.8 % of the step, all 14 launches ~9 %).
    # `roofline` and the per-kernel table come from a SECOND timed region of the same K steps in which every launch is
    # bracketed (its per-kernel averages agree with the rocprofv3 trace in profiles/).
    # barrier + synchronize, exactly K steps, barrier + synchronize, max over ranks
    elapsed = sharding.timed_steps(step, args.steps, sync_dev, dist=d, device=red_dev)
    kernels = []
    if not args.no_profile:
        tr.profile_begin(max_events=2 * 48 * (args.steps + 1))
        sharding.timed_steps(step, args.steps, sync_dev, dist=d, device=red_dev)
        kernels, _ = tr.profile_end()
    emb = out.cpu().numpy() if rank == 0 else None
    extra = not args.no_extra
    # extra legs (never `value`): repetitions of the same region, the same K steps replayed from a captured hipGraph
    reps = []
    if extra and args.reps > 0:
        for _ in range(args.reps):
            reps.append(sharding.timed_steps(step, args.steps, sync_dev, dist=d, device=red_dev))
    graph_rate = None
    if extra and graph is None and not args.varlen:
        g2, _ = tr.capture_graph(feats, offsets, out=out)
        for _ in range(args.warmup):
            g2.replay()
        el2 = sharding.timed_steps(g2.replay, args.steps, sync_dev, dist=d, device=red_dev)
        graph_rate = n_gpus * args.batch * args.steps / el2
        del g2
        tr.release_graphs()                                # the graph is gone: its plan / workspace may be reused
    tdnn_default = args.network == "tdnn" and args.pooling == "statistics_pooling"
    # the same K-step region in the full-range precision (bf16x3: no fp16 range to guard), for reference beside `value`
    bf_rate = None
    if extra and precision != "bf16x3" and not args.graph:
        tr2 = Trainer(params, None, args.dim, single_cpu=True, device=local_rank, precision="bf16x3")
        tr2.build("predict")
        tr2.load_weights(weights)
        out2 = torch.empty_like(out)

        def step2():
            tr2.predict_packed(feats, offsets, out=out2)

        for _ in range(args.warmup + 10):
            step2()
        sync_dev()
        el_bf = sharding.timed_steps(step2, args.steps, sync_dev, dist=d, device=red_dev)
        bf_rate = n_gpus * args.batch * args.steps / el_bf
        tr2.close()
        del out2
    c4 = None
    if extra and tdnn_default and args.c4_steps > 0:
        c4 = config4_leg(tr, args, dist, world, rank, dev, weights, params)
    e2e_rate = e2e_sync = None
    if extra and n_gpus == 1 and rank == 0:
        # host numpy -> pinned staging -> H2D -> kernels -> D2H.  e2e_value: the caller keeps two batches in flight
        # (Trainer.submit_list / collect: what the command-line driver does); e2e_sync_value: one blocking predict_list per batch
        for _ in range(3):
            tr.predict_list(utts)                          # warm-up (staging slots, copy stream)
        n_e2e = max(8, args.steps)
        t1 = time.perf_counter()
        tickets, t_sub, t_col = [], 0.0, 0.0
        for i in range(n_e2e):
            t2 = time.perf_counter()
            tickets.append(tr.submit_list(utts))
            t3 = time.perf_counter()
            if len(tickets) == 2:
                tr.collect(tickets.pop(0))
            t_sub += t3 - t2
            t_col += time.perf_counter() - t3
        while tickets:
            tr.collect(tickets.pop(0))
        e2e_rate = n_e2e * args.batch / (time.perf_counter() - t1)
        print("[bench] e2e pipelined: %.3f ms per batch (submit %.3f, collect %.3f)"
              % ((time.perf_counter() - t1) / n_e2e * 1e3, t_sub / n_e2e * 1e3, t_col / n_e2e * 1e3), file=sys.stderr)
        calls = []
        for i in range(max(3, args.steps // 2)):
            t1 = time.perf_counter()
            tr.predict_list(utts)
            calls.append(time.perf_counter() - t1)
        e2e_sync = args.batch / sorted(calls)[len(calls) // 2]     # median call: one stalled call must not set the rate
    result = None
    if rank == 0:
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
            if precision == "f16f6" and dom["name"].endswith("_conv"):
                roof["mfma_issue_frac"] = round(1.5 * tf / peak, 4)
                roof["note"] = ("two-unit split: hi*hi on the f16 MFMA + two cross terms on the block-scaled fp6 MFMA (4x rate) = 1.5 f16 "
                                "MFMA units per algorithmic product (K groups formed over quads of channel blocks: no zero-weight groups), so the ceiling of "
                                "`frac` is 2/3; mfma_issue_frac = those units / f16 dense peak")
            elif precision != "f32":
                roof["mfma_issue_frac"] = round(3 * tf / peak, 4)
                roof["note"] = ("split precision: 3 MFMAs per algorithmic product, so the ceiling of `frac` is 1/3; "
                                "mfma_issue_frac = issued MFMA FLOPs / bf16 dense peak")
        flops_step = float(info["flops"])
        result = {
            "metric": "utterances/sec x-vector extraction (30-dim x 300-frame)",
            "value": round(value, 1), "unit": "utterances/s", "n_gpus": n_gpus, "steps": args.steps,
            "warmup": args.warmup, "prewarm_s": args.prewarm, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16x3": "bf16x3(f32-split)", "f16x3": "f16x3(f32-split)", "f16f6": "f16+fp6-cross(f32-split)"}[precision], "data": "synthetic",
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
            "kernels_note": "per-kernel hipEvent times of a second region of the same %d steps (all launches bracketed)" % args.steps,
            "roofline": roof,
        }
        if reps:
            rates = sorted(total_utts / r for r in reps)
            result["value_reps"] = {"n": len(rates), "min": round(rates[0], 1), "median": round(rates[len(rates) // 2], 1),
                                    "max": round(rates[-1], 1), "what": "repetitions of the same %d-step timed region" % args.steps}
        result["e2e_value"] = round(e2e_rate, 1) if e2e_rate else None
        result["e2e_sync_value"] = round(e2e_sync, 1) if e2e_sync else None
        result["value_bf16x3"] = round(bf_rate, 1) if bf_rate else None
        result["config4"] = c4
    tr.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        torch.cuda.synchronize(dev)
        result["cli"] = None
        if extra and n_gpus == 1 and tdnn_default and not args.varlen and args.cli_utts > 0:
            try:
                result["cli"] = cli_leg(weights, params, args.dim, args.frames, args.cli_utts, precision)
            except Exception as e:              # the extra leg must not cost the bench line
                result["cli"] = {"error": str(e)[:300]}
        result["launcher"] = None
        if extra and n_gpus == 1 and tdnn_default and not args.varlen and args.c4_utts > 0:
            try:
                result["launcher"] = launcher_leg(weights, params, args.dim, args.c4_utts, precision)
            except Exception as e:
                result["launcher"] = {"error": str(e)[:300]}
        if args.cpu_seconds > 0 and n_gpus == 1:
            result["cpu_baseline"] = cpu_baseline(weights, params, args.dim, args.frames, args.cpu_seconds, args.cpu_nj)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))


if __name__ == "__main__":
    main()
