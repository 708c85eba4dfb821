"""Input-inclusive training rate: TFRecord files -> libdmt_input.so (N threads) -> pageable host arrays -> PCIe -> train step,
with one prefetch thread (the reference's tf.data pipeline + feed).  bench.py's `value` keeps its batches resident in HBM; this is
the figure DESIGN.md quotes next to it.   usage: e2e_records_bench.py [steps] [parser threads]"""
import os, queue, sys, tempfile, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed import native, tfrecord
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.engine import DeviceBatch
from cikm2020_dmt_amd.train import Trainer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
nthreads = int(sys.argv[2]) if len(sys.argv) > 2 else 16
B, NFILES = 4096, 2
sp = S.e64_spec()
emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
feats = list(dict.fromkeys(e[3] for e in emb))
name_of = {e[3]: e[0] for e in reversed(emb)}
tmp = tempfile.mkdtemp()
files = []
t0 = time.perf_counter()
for fi in range(NFILES):
    inputs, mask, label = make_batch(sp, B, seed=900 + fi, lengths="full")
    rows = {f: inputs[f].rows() for f in feats}
    recs = []
    for b in range(B):
        ex = {"features": inputs["features"][b].astype(np.float32), "mask": mask[b].astype(np.float32), "label": np.array([label[b]], np.float32)}
        for f in feats:
            ex[f] = [("%d" % int(i)).encode() for i in rows[f][b]]
            ex[f + "Wts"] = np.ones(len(rows[f][b]), np.float32)
        recs.append(tfrecord.encode_example(ex))
    path = os.path.join(tmp, "part-r-%05d" % fi)
    tfrecord.write_records(path, recs)
    files.append(path)
print("wrote %d files x %d records (%.1f MB) in %.0f s" % (NFILES, B, sum(os.path.getsize(f) for f in files) / 1e6, time.perf_counter() - t0), flush=True)
vocabs = {}
for (name, nrows, _d, _f, _s) in emb:
    vocabs.setdefault(name, native.Vocab(["unknow"], nrows) if nrows > 23 else native.Vocab(["unknow"] + [str(i) for i in range(1, nrows)], nrows))
T = {f: max(int(inputs[f].dense_shape[1]), 1) for f in feats}
parser = native.BatchParser([(f, vocabs[name_of[f]], T[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)], n_threads=nthreads)
parser.pinned = True
dev = torch.device("cuda")
tr = Trainer(sp, device=dev, compute_dtype=torch.bfloat16, seed=1234, dropout=True)
q = queue.Queue(maxsize=3)
def producer():
    n = 0
    while n < steps + 6:
        for cols in parser.batches(files, B, verify_crc=True):
            q.put(DeviceBatch.from_columns(cols, sp, dev))
            n += 1
            if n >= steps + 6:
                break
    q.put(None)
t0 = time.perf_counter()
k = 0
for cols in parser.batches(files, B, verify_crc=True):
    bt = DeviceBatch.from_columns(cols, sp, dev)
    k += 1
torch.cuda.synchronize()
print("producer alone: %.2f ms per batch (parse + from_columns + upload)" % ((time.perf_counter() - t0) / k * 1e3), flush=True)
t0 = time.perf_counter()
for cols in parser.batches(files, B, verify_crc=True):
    pass
print("parse alone: %.2f ms per batch" % ((time.perf_counter() - t0) / k * 1e3), flush=True)
th = threading.Thread(target=producer, daemon=True)
th.start()
for _ in range(5):
    tr.train_step(q.get())
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    tr.train_step(q.get())
torch.cuda.synchronize()
dt = time.perf_counter() - t0
while q.get() is not None:      # let the producer finish before the interpreter tears down
    pass
th.join()
print("input-inclusive: %.1f samples/s, %.3f ms/step (%d parser threads, batches from TFRecord files, PCIe upload each step)" % (B * steps / dt, dt / steps * 1e3, nthreads))
