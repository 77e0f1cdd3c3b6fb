"""Embedding extraction driver: Kaldi matrix ark in -> x-vector ark out.

Same command line and per-utterance behaviour as the reference driver
egs/voxceleb/v1/nnet/lib/extract.py (argparse :11-24; model dir contract :42-55; scp
rspecifier refused :59-61; min-length skip :65-67; long-utterance half-overlap chunking and
length-weighted average :68-86; optional L2 normalisation :84-85,91-92; vector ark out :93):

    extract.py [-g GPU] [-m MIN] [-s CHUNK] [-n] [--node NODE] model_dir rspecifier wspecifier

What differs is how the device is fed: instead of one sess.run per utterance (extract.py:89)
utterances (and the chunks of long ones) are packed back to back into ragged batches of up
to --batch-frames frames and sent through `Trainer.predict_list` in one launch sequence;
results are written in input order, so the output ark is the one the reference would write.
"""
import argparse
import logging
import os
import sys

import numpy as np

from .kaldi_io import open_or_fd, read_mat_ark, write_vec_flt
from .params import Params

log = logging.getLogger("xvec.extract")


def build_parser():
    parser = argparse.ArgumentParser()
    parser.add_argument("-g", "--gpu", type=int, default=-1,
                        help="The GPU id.  -1 (the reference's 'GPU disabled'; what run_extract_embeddings.sh passes to "
                             "every one of its nj jobs) picks a device for this job: (JOB-1) mod visible GPUs when the job "
                             "index can be read off the specifiers (xvector.JOB.ark / splitN/JOB/), else LOCAL_RANK, "
                             "else 0.  This implementation has no CPU path.")
    parser.add_argument("-m", "--min-chunk-size", type=int, default=25,
                        help="The minimum length of the segments. Any segment shorted than this value will be ignored.")
    parser.add_argument("-s", "--chunk-size", type=int, default=10000,
                        help="The length of the segments used to extract the embeddings. Segments longer than this value "
                             "will be splited before extraction. Then the splited embeddings will be averaged to get the "
                             "final embedding. L2 normalizaion will be applied before the averaging if specified.")
    parser.add_argument("-n", "--normalize", action="store_true", help="Normalize the embedding before averaging and output.")
    parser.add_argument("--node", type=str, default="", help="The node to output the embeddings.")
    parser.add_argument("--batch-frames", type=int, default=153600,
                        help="Frames packed into one device batch (extension; 153600 = 512 utterances x 300 frames: the fixed cost per "
                             "batch -- plan look-up, pipeline hand-over -- is paid half as often as with 76800).")
    parser.add_argument("--precision", type=str, default="", help="f32 | bf16x3 | f16x3 | f16f6 (extension; default: library default)")
    parser.add_argument("--scp-input", action="store_true",
                        help="Accept `scp:<file>` as the rspecifier and read its records natively by seeking (extension; "
                             "the reference refuses scp input, extract.py:59-61, because Kaldi binaries expand it upstream).")
    parser.add_argument("--python-reader", action="store_true",
                        help="Parse the ark with the pure-Python reader instead of the native batch reader (extension).")
    parser.add_argument("--cmn-window", type=int, default=0,
                        help="Apply centred sliding-window CMN of this many frames on the GPU before the network "
                             "(extension; replaces `apply-cmvn-sliding --norm-vars=false --center=true` of "
                             "run_extract_embeddings.sh:47; 0 = features are already normalised).")
    parser.add_argument("--vad-rspecifier", type=str, default="",
                        help="Kaldi vector ark of per-frame VAD decisions, same key order as the features; voiced frames "
                             "are selected on the GPU (extension; replaces `select-voiced-frames`).")
    parser.add_argument("model_dir", type=str, help="The model directory.")
    parser.add_argument("rspecifier", type=str, help="Kaldi feature rspecifier (or ark file).")
    parser.add_argument("wspecifier", type=str, help="Kaldi output wspecifier (or ark file).")
    return parser


def split_chunks(num_frames, chunk_size):
    """(start, length) of the pieces of an utterance longer than chunk_size: chunks of
    chunk_size at hop chunk_size//2, the last one shorter (extract.py:70-77; `chunk_size / 2`
    there is Python-2 integer division)."""
    half = chunk_size // 2
    num_chunks = int(np.ceil(float(num_frames - chunk_size) / half)) + 1
    pieces = []
    for i in range(num_chunks):
        start = i * half
        this = chunk_size if num_frames - start > chunk_size else num_frames - start
        pieces.append((start, this))
    return pieces


def combine_chunks(embeddings, lengths, normalize):
    """extract.py:83-86: optional per-chunk L2 normalisation, then length-weighted mean."""
    embeddings = np.asarray(embeddings)
    lengths = np.expand_dims(np.asarray(lengths), axis=1)
    if normalize:
        embeddings = embeddings / np.sqrt(np.sum(np.square(embeddings), axis=1, keepdims=True))
    return np.sum(embeddings * lengths, axis=0) / np.sum(lengths)


def _batches(items, min_chunk_size, chunk_size, batch_frames, counters):
    """Group the (key, matrix) stream into device batches: yields (pieces, pending) where `pieces` is
    the list of feature matrices to embed and `pending` = [(key, piece indices, chunk lengths | None)]."""
    pending, pieces, frames = [], [], 0
    for key, feature in items:
        t = feature.shape[0]
        if t < min_chunk_size:
            log.info("[INFO] Key %s length too short, %d < %d, skip.", key, t, min_chunk_size)
            counters["skipped"] += 1
            continue
        if t > chunk_size:
            parts = split_chunks(t, chunk_size)
            log.info("[INFO] Key %s length %d > %d, split to %d segments.", key, t, chunk_size, len(parts))
            idx = []
            for start, length in parts:
                idx.append(len(pieces))
                pieces.append(feature[start:start + length])
            pending.append((key, idx, [p[1] for p in parts]))
        else:
            log.info("[INFO] Key %s length %d.", key, t)
            pending.append((key, [len(pieces)], None))
            pieces.append(feature)
        frames += t
        if frames >= batch_frames:
            yield pieces, pending
            pending, pieces, frames = [], [], 0
    if pending:
        yield pieces, pending


def extract_stream(embed_fn, items, write_fn, min_chunk_size=25, chunk_size=10000, normalize=False,
                   batch_frames=76800, prefetch=2):
    """Core loop.  `items` yields (key, [T,d] matrix); `embed_fn(list of [T_i,d])` returns an
    [n,E] array; `write_fn(key, vector)` is called once per kept utterance, in input order.
    With prefetch > 0 a producer thread parses the ark and groups the next batches while the caller's
    thread embeds and writes the current one (ark parsing, device work and output overlap).
    Returns (#written, #skipped)."""
    counters = {"skipped": 0}
    gen = _batches(items, min_chunk_size, chunk_size, batch_frames, counters)
    if prefetch > 0:
        import queue
        import threading
        q = queue.Queue(maxsize=prefetch)
        done_marker = object()

        def producer():
            try:
                for b in gen:
                    q.put(b)
                q.put(done_marker)
            except BaseException as e:          # surface parser errors in the consumer thread
                q.put(e)

        threading.Thread(target=producer, daemon=True).start()

        def batches():
            while True:
                b = q.get()
                if b is done_marker:
                    return
                if isinstance(b, BaseException):
                    raise b
                yield b
        source = batches()
    else:
        source = gen
    done = 0
    for pieces, pending in source:
        emb = np.asarray(embed_fn(pieces))
        for key, idx, lengths in pending:
            if lengths is None:
                e = emb[idx[0]]
            else:
                e = combine_chunks(emb[idx], lengths, normalize)
            if normalize:
                e = e / np.sqrt(np.sum(np.square(e)))                      # extract.py:91-92
            write_fn(key, np.asarray(e, dtype=np.float32))
            done += 1
    return done, counters["skipped"]


def auto_device(rspecifier, wspecifier, device_count=None, environ=None):
    """Device of a job started with `--gpu -1`.  The reference's launcher gives every job `--gpuid -1`
    (run_extract_embeddings.sh:68-71) and run.pl substitutes the job index only into the specifier strings
    (`.../split8/3/feats.scp`, `xvector.3.ark`), so that is where it is read from; jobs then spread over the
    visible GPUs instead of all landing on device 0."""
    import re
    environ = os.environ if environ is None else environ
    if device_count is None:
        import torch
        device_count = torch.cuda.device_count()        # does not initialise the GPU
    device_count = max(int(device_count), 1)
    for spec, pat in ((wspecifier, r"xvector\.(\d+)\.(?:ark|scp)"), (rspecifier, r"/split\d+[a-z]*/(\d+)/"),
                      (wspecifier, r"\.(\d+)\.ark")):
        m = re.search(pat, spec or "")
        if m:
            return (int(m.group(1)) - 1) % device_count
    if "LOCAL_RANK" in environ:
        return int(environ["LOCAL_RANK"]) % device_count
    return 0


def _vad_records(vad_rspecifier):
    """(key, vector) stream of a VAD table (`ark:` / `scp:`), parsed in batches by the native reader (float-vector
    records arrive as [dim, 1] matrices); .gz and text tables go through the Python reader."""
    spec = vad_rspecifier.strip()
    plain = spec.split(":", 1)[-1].strip()
    if plain.endswith(".gz"):
        from .kaldi_io import read_vec_flt_ark
        for kv in read_vec_flt_ark(vad_rspecifier):
            yield kv
        return
    from . import native_ark
    reader = native_ark.ArkBatchReader(vad_rspecifier, batch_frames=1 << 20, min_frames=0, capacity=(1 << 20) + (1 << 18))
    try:
        for keys, offsets, data in reader:
            flat = data.reshape(-1).copy()
            for i, k in enumerate(keys):
                yield k, flat[offsets[i]:offsets[i + 1]]
    finally:
        reader.close()


def _vad_lookup(vad_rspecifier):
    """Lock-step lookup in a VAD vector table that is in the same (sorted) key order as the features (Kaldi's `scp,s,cs`
    contract of select-voiced-frames): returns f(key) -> vector, skipping VAD entries without features.  An utterance
    without VAD decisions gives None -- select-voiced-frames warns and drops it, it does not end the job -- and the
    caller counts it as skipped."""
    it = iter(_vad_records(vad_rspecifier))
    held = []                               # one VAD record read ahead of the features (its key sorts behind the one asked for)

    def lookup(key):
        while True:
            if held:
                k, v = held.pop()
            else:
                try:
                    k, v = next(it)
                except StopIteration:
                    return None
            if k == key:
                return v
            if k > key:                     # sorted tables: the VAD table has no entry for `key`
                held.append((k, v))
                return None
    return lookup


def run_native(trainer, rspecifier, writer, min_chunk_size, chunk_size, normalize, batch_frames, cmn_window=0,
               vad_rspecifier=""):
    """Fast path of the driver, three stages that overlap:
      reader thread   the native batch reader (csrc/ark_io.cpp) parses ark records straight into pinned staging buffers
                      (outside the GIL);
      this thread     copies batch i to the device on a COPY stream (beside the kernels of batch i - 1), enqueues the forward,
                      the copy of the result into a pinned slot and the read-out of the fp16 range flags (xv_flags_async: no
                      host synchronisation), then waits for batch i - 2 and hands it over;
      writer thread   range check, L2 normalisation, vector formatting (native, outside the GIL) and the write.
    Two batches are in flight on the device at any time, so neither the host-side output work nor the H2D copy of the next batch
    leaves it idle.  Batches containing an utterance longer than chunk_size, and the front-end path, go through the synchronous
    code on views of the same buffers (in order: everything in flight is written first)."""
    import queue
    import threading
    import time
    import torch
    from . import native_ark
    t_enter = time.perf_counter()
    cap = (batch_frames + 65536) * 64
    NPIN, NSLOT, KEEP = 6, 5, 2
    # six staging buffers: the reader is at most two batches ahead (one being filled, one queued), three are in flight on the
    # device or being copied (KEEP + the one just taken), one spare.  Five result slots: KEEP in flight, one queued for the
    # writer, one the writer is working on, one for the batch being enqueued.
    pins = [torch.empty(cap, dtype=torch.float32, pin_memory=True) for _ in range(NPIN)]
    frontend = cmn_window > 0 or bool(vad_rspecifier)
    vad_of = _vad_lookup(vad_rspecifier) if vad_rspecifier else None
    # with the front-end on, the min-length rule applies to the lengths after frame selection
    reader = native_ark.ArkBatchReader(rspecifier, batch_frames=batch_frames, min_frames=1 if frontend else min_chunk_size,
                                       buffers=[p.numpy() for p in pins])
    q = queue.Queue(maxsize=1)
    end = object()

    def producer():
        try:
            for b in reader:
                q.put(b)
            q.put(end)
        except BaseException as e:
            q.put(e)

    threading.Thread(target=producer, daemon=True).start()
    done = extra_skipped = 0
    dev_index = trainer._device_index
    devname = "cuda:%d" % dev_index
    pending = []                          # batches enqueued on the device, oldest first: (keys, pinned embeddings, pinned flags, event)
    wq = queue.Queue(maxsize=1)
    werr = []

    def consumer():
        while True:
            item = wq.get()
            try:
                if item is None:
                    return
                if werr:
                    continue                            # drain after a failure: the main thread raises it
                keys_p, host_p, flags_p, dev_p, offs_p = item
                # (a batch the fp16 precisions refuse runs again on the bf16x3 twin: its device copy is intact -- the main thread
                #  cannot reach the turn that reuses the slot before this batch has left the queue)
                emb = trainer.checked_or_rerun(host_p.numpy(), dev_p, offs_p, code=trainer.decode_flags(flags_p))
                if normalize:
                    emb = emb / np.sqrt(np.sum(np.square(emb), axis=1, keepdims=True))
                writer.write(keys_p, emb)
            except BaseException as e:
                werr.append(e)
            finally:
                wq.task_done()

    wthread = threading.Thread(target=consumer, daemon=True)
    wthread.start()

    waited = {"reader": 0.0, "device": 0.0, "writer": 0.0}     # where this thread sat still (logged at the end)

    def flush(keep=0):
        nonlocal done
        while len(pending) > keep:
            keys_p, host_p, flags_p, ev, dev_p, offs_p = pending.pop(0)
            t0 = time.perf_counter()
            ev.synchronize()
            t1 = time.perf_counter()
            if werr:
                raise werr[0]
            wq.put((keys_p, host_p, flags_p, dev_p, offs_p))      # blocks while the writer is a batch behind: its slots stay untouched
            waited["device"] += t1 - t0
            waited["writer"] += time.perf_counter() - t1
            done += len(keys_p)

    with torch.cuda.device(dev_index):
        comp = torch.cuda.current_stream(dev_index)
        copy_s = torch.cuda.Stream(device=dev_index)
        dev_in = [None] * NSLOT
        emb_pins = [None] * NSLOT
        flag_pins = [torch.zeros(2, dtype=torch.int32).pin_memory() for _ in range(NSLOT)]
        slot_done = [None] * NSLOT
    turn = 0
    t_loop0 = time.perf_counter()
    try:
        while True:
            t0 = time.perf_counter()
            b = q.get()
            waited["reader"] += time.perf_counter() - t0
            if b is end:
                break
            if isinstance(b, BaseException):
                raise b
            keys, offsets, feats = b
            lens_max = int(np.diff(offsets).max())
            if frontend or lens_max > chunk_size:
                flush()                                     # these paths write synchronously: keep the output order
                _writer_idle(wq, werr)
            if frontend:
                from .frontend import cmn_select_packed
                vads = [vad_of(k) for k in keys] if vad_of else None
                if vads is not None and any(v is None for v in vads):
                    missing = [k for k, v in zip(keys, vads) if v is None]
                    log.warning("[WARNING] no VAD decisions for %d utterance(s) (first: %s): skipped" % (len(missing), missing[0]))
                with torch.cuda.device(dev_index):
                    raw = torch.from_numpy(feats).to(devname, non_blocking=True)
                    dev, offsets, kept = cmn_select_packed(raw, offsets, vads, cmn_window=cmn_window,
                                                           min_frames=min_chunk_size)
                extra_skipped += len(keys) - len(kept)
                keys = [keys[i] for i in kept]
                if not keys:
                    continue
                if np.diff(offsets).max() > chunk_size:     # rare: finish on the host views of the processed batch
                    feats = dev.cpu().numpy()
                else:
                    with torch.cuda.device(dev_index):
                        emb = trainer.checked_or_rerun(trainer.predict_packed(dev, offsets).cpu().numpy(), dev, offsets)
                    if normalize:
                        emb = emb / np.sqrt(np.sum(np.square(emb), axis=1, keepdims=True))
                    writer.write(keys, emb)
                    done += len(keys)
                    continue
            lens = np.diff(offsets)
            if lens.max() > chunk_size:                     # rare: chunk / weight / average on views
                items = [(k, feats[offsets[i]:offsets[i + 1]]) for i, k in enumerate(keys)]
                out = []
                extract_stream(trainer.predict_list, iter(items), lambda k, v: out.append((k, v)), min_chunk_size,
                               chunk_size, normalize, batch_frames, prefetch=0)
                writer.write([k for k, _ in out], np.stack([v for _, v in out]))
                done += len(out)
                continue
            host = torch.from_numpy(feats)                  # view of the pinned staging buffer
            k = turn % NSLOT
            with torch.cuda.device(dev_index):
                if dev_in[k] is None or dev_in[k].shape[0] < host.shape[0] or dev_in[k].shape[1] != host.shape[1]:
                    dev_in[k] = torch.empty((max(int(host.shape[0] * 1.25), 4096), host.shape[1]), dtype=torch.float32, device=devname)
                    copy_s.wait_stream(comp)                # the block may still be in use by work queued on the compute stream
                elif slot_done[k] is not None:
                    copy_s.wait_event(slot_done[k])         # the forward that last read this buffer has finished
                with torch.cuda.stream(copy_s):
                    dev_in[k][:host.shape[0]].copy_(host, non_blocking=True)
                    h2d = torch.cuda.Event()
                    h2d.record(copy_s)
                comp.wait_event(h2d)
                info = trainer.plan_info(offsets)
                rows_o, cols_o = int(info["out_rows"]), int(info["out_cols"])
                flat = emb_pins[k]
                if flat is None or flat.numel() < rows_o * cols_o:
                    flat = emb_pins[k] = torch.empty(max(rows_o * cols_o, 4096 * cols_o), dtype=torch.float32, pin_memory=True)
                host_out = flat[:rows_o * cols_o].view(rows_o, cols_o)
                if info["frame_level"]:
                    host_out.copy_(trainer.predict_packed(dev_in[k][:host.shape[0]], offsets), non_blocking=True)
                else:                                       # the last kernel writes the embeddings straight into the pinned slot
                    trainer.predict_packed(dev_in[k][:host.shape[0]], offsets, out=host_out)
                trainer.flags_async(flag_pins[k])
                ev = torch.cuda.Event()
                ev.record(comp)
                slot_done[k] = ev
            turn += 1
            pending.append((list(keys), host_out, flag_pins[k], ev, dev_in[k][:host.shape[0]], np.array(offsets, dtype=np.int32)))
            flush(keep=KEEP)                                # hand batch i - 2 to the writer while the device runs i - 1 and i
        flush()
    finally:
        wq.put(None)
        wthread.join()
    if werr:
        raise werr[0]
    log.info("[INFO] driver loop: %d batches; waited %.3f s for the reader, %.3f s for the device, %.3f s for the writer; "
             "set-up (staging buffers, reader, threads) %.3f s, loop %.3f s"
             % (turn, waited["reader"], waited["device"], waited["writer"], t_loop0 - t_enter, time.perf_counter() - t_loop0))
    skipped = reader.skipped + extra_skipped
    # (closing unmaps the input -- gigabytes of page cache for a large ark, ~0.1 s -- : beside the caller's own teardown)
    threading.Thread(target=reader.close, daemon=True).start()
    return done, skipped


def _writer_idle(wq, werr):
    """Block until the writer thread has consumed everything queued so far (queue.join semantics via task counts)."""
    wq.join()
    if werr:
        raise werr[0]


def main(argv=None):
    args = build_parser().parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(message)s")
    nnet_dir = os.path.join(args.model_dir, "nnet")
    config_json = os.path.join(args.model_dir, "nnet/config.json")
    if not os.path.isfile(config_json):
        sys.exit("Cannot find params.json in %s" % config_json)
    params = Params(config_json)
    if len(args.node) != 0:                                  # extract.py:50-51
        params.embedding_node = args.node
    log.info("Extract embedding from %s" % params.embedding_node)
    with open(os.path.join(nnet_dir, "feature_dim"), "r") as f:
        dim = int(f.readline().strip())

    spec = args.rspecifier.strip()
    if spec.rsplit(".", 1)[-1] == "scp" and not (args.scp_input and spec.startswith(("scp:", "scp,"))):   # extract.py:59-61
        sys.exit("The rspecifier must be ark or input pipe")

    from .trainer import Trainer
    device = args.gpu if args.gpu >= 0 else auto_device(args.rspecifier, args.wspecifier)
    log.info("Using HIP device %d" % device)
    trainer = Trainer(params, args.model_dir, dim, single_cpu=True, device=device, precision=args.precision or None)
    trainer.build("predict")

    from . import native_ark
    writer = native_ark.VectorWriter(args.wspecifier)
    plain = spec.split(":", 1)[-1].strip()
    frontend = args.cmn_window > 0 or bool(args.vad_rspecifier)
    native = not args.python_reader and not plain.endswith(".gz")
    if native and not plain.endswith("|") and not spec.startswith(("scp:", "scp,")) and not _binary_ark(plain):
        log.info("[INFO] %s is not a binary ark: using the Python reader." % plain)
        native = False                      # text-mode ark: the reference's reader handles it (kaldi_io.py:1056-1068)
    if frontend and not native:
        sys.exit("--cmn-window / --vad-rspecifier need the native reader (binary ark, scp or pipe input)")
    import time
    if trainer.model is not None and os.path.isfile(os.path.join(trainer.model, "checkpoint")):
        trainer.load()                      # the reference restores lazily inside the first predict (trainer.py:891-895)
    t_loop = time.perf_counter()
    if native:
        done, skipped = run_native(trainer, args.rspecifier, writer, args.min_chunk_size, args.chunk_size,
                                   args.normalize, args.batch_frames, args.cmn_window, args.vad_rspecifier)
    else:
        if spec.startswith(("scp:", "scp,")):
            from .kaldi_io import read_mat_scp
            items = read_mat_scp(plain)
        else:
            items = read_mat_ark(args.rspecifier)
        done, skipped = extract_stream(
            trainer.predict_list, items, lambda key, vec: writer.write([key], vec[None, :]),
            min_chunk_size=args.min_chunk_size, chunk_size=args.chunk_size, normalize=args.normalize,
            batch_frames=args.batch_frames)
    rc = writer.close()
    elapsed = time.perf_counter() - t_loop          # read - embed - write incl. the output flush; not the teardown below
    trainer.close()
    log.info("Extracted %d embeddings (%d utterances skipped) in %.3f s" % (done, skipped, elapsed))
    if rc != 0:
        log.error("the output command of %s exited with code %d" % (args.wspecifier, rc))
        return 1
    return 0


def _binary_ark(path):
    """True when the first record of the file carries the binary marker `\\0B` after its key."""
    try:
        with open(path, "rb") as f:
            head = f.read(4096)
    except OSError:
        return True                         # let the reader report the real error
    sp = head.find(b" ")
    return sp < 0 or head[sp + 1:sp + 3] == b"\0B" or len(head) < sp + 3


if __name__ == "__main__":
    sys.exit(main())
