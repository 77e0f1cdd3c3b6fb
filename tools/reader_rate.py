"""Rate of the native ark batch reader alone (no GPU work): utterances/s and GB/s of 300 x 30 float matrices from a page-cached
ark, by number of payload-copy threads.  usage: reader_rate.py [n_utts] [dir]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tf_kaldi_speaker_amd import native_ark, kaldi_io
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
d = sys.argv[2] if len(sys.argv) > 2 else "/dev/shm"
path = os.path.join(d, "xv_reader_rate.ark")
m = np.random.RandomState(0).standard_normal((300, 30)).astype(np.float32)
with open(path, "wb") as f:
    for i in range(N):
        kaldi_io.write_mat(f, m, key="utt%07d" % i)
try:
    import torch
    pin = torch.cuda.is_available()
except Exception:
    pin = False
cap = (76800 + 65536) * 64
for threads in (1, 2, 4, 8):
    bufs = None
    if pin:
        import torch
        ts = [torch.empty(cap, dtype=torch.float32).pin_memory() for _ in range(3)]
        bufs = [t.numpy() for t in ts]
    best = 0.0
    for rep in range(3):
        rd = native_ark.ArkBatchReader("ark:" + path, batch_frames=76800, buffers=bufs, copy_threads=threads)
        t0 = time.perf_counter(); n = 0
        for b in rd:
            n += 1
        dt = time.perf_counter() - t0
        rd.close()
        best = max(best, N / dt)
    print("copy threads %d (%s buffers): %.0f utt/s = %.2f GB/s" % (threads, "pinned" if pin else "pageable", best, best * 36000 / 1e9))
os.unlink(path)
