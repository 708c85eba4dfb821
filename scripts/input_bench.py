"""Input-stage throughput: libdmt_input.so vs the Python restatement on synthetic records of the model's schema
(full lengths 50/50/10, 615 dense floats; ~13 KB per record as in the reference's data).  CPU only."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed import native, tfrecord
from cikm2020_dmt_amd.data_feed.synthetic import make_batch

sp = S.default_spec()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
inputs, mask, label = make_batch(sp, B, seed=3, lengths="full")
emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
feats = list(dict.fromkeys(e[3] for e in emb))
name_of = {e[3]: e[0] for e in reversed(emb)}
rows = {f: inputs[f].rows() for f in feats}
recs = []
for b in range(B):
    ex = {"features": inputs["features"][b].astype(np.float32), "mask": mask[b].astype(np.float32), "label": np.array([label[b]], np.float32),
          "header": [b"x" * 120]}
    for f in feats:
        ex[f] = [("%d" % int(i)).encode() for i in rows[f][b]]       # decimal-string ids, all out-of-vocabulary -> every id is hashed
        ex[f + "Wts"] = np.ones(len(rows[f][b]), np.float32)
    recs.append(tfrecord.encode_example(ex))
path = os.path.join(tempfile.mkdtemp(), "part-r-00000")
tfrecord.write_records(path, recs)
nbytes = os.path.getsize(path)
vocabs = {}
for (name, nrows, _d, _f, _s) in emb:
    vocabs.setdefault(name, native.Vocab(["unknow"], nrows) if nrows > 23 else native.Vocab(["unknow"] + [str(i) for i in range(1, nrows)], nrows))
T = {f: max(int(inputs[f].dense_shape[1]), 1) for f in feats}
print('host cores', os.cpu_count())
for nt in (1, 4, 8, 16, 32):
    parser = native.BatchParser([(f, vocabs[name_of[f]], T[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)], n_threads=nt)
    t0 = time.perf_counter()
    n = 0
    for cols in parser.batches([path], 1024):
        n += cols["label"].shape[0]
    dt = time.perf_counter() - t0
    print("native %d thread(s): %8.0f records/s  %7.1f MB/s" % (nt, n / dt, nbytes / dt / 1e6))
t0 = time.perf_counter()
k = 0
for r in tfrecord.read_records(path, verify_crc=True):
    tfrecord.decode_example(r)
    k += 1
    if k == 200:
        break
dt = time.perf_counter() - t0
print("python (framing + crc + decode only, no lookup): %8.0f records/s" % (k / dt))
print("record size %.1f KB" % (nbytes / B / 1e3))
