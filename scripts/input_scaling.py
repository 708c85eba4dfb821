"""Parallel efficiency of the native input stage (libdmt_input.so): records / s of BatchParser.batches over synthetic TFRecord files of
the benchmark's shape, per thread count, against the one-thread rate.

    python scripts/input_scaling.py [files] [passes] [ring]      (ring: reused output buffers, 0 = a fresh one per batch)
"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cikm2020_dmt_amd import spec as S                                            # noqa: E402
from cikm2020_dmt_amd.data_feed import native                                     # noqa: E402
from cikm2020_dmt_amd.data_feed.synthetic import make_batch, write_records_file   # noqa: E402


def main(nf, passes, ring=3):
    sp = S.e64_spec()
    B = 4096
    tmp = tempfile.mkdtemp(prefix="dmt_records_")
    files = [os.path.join(tmp, "part-r-%05d" % i) for i in range(nf)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(nf, 8)) as pool:
        pool.map(write_records_file, [("e64", 0, B, 777000 + i, "zipf", f) for i, f in enumerate(files)])
    emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
    feats = list(dict.fromkeys(e[3] for e in emb))
    name_of = {e[3]: e[0] for e in reversed(emb)}
    vocabs = {}
    for (name, nrows, _d, _f, _s) in emb:
        vocabs.setdefault(name, native.Vocab(["unknow"], nrows) if nrows > 23 else native.Vocab(["unknow"] + [str(i) for i in range(1, nrows)], nrows))
    probe, _m, _l = make_batch(sp, 2, seed=1, lengths="full", law="zipf")
    T = {f: max(int(probe[f].dense_shape[1]), 1) for f in feats}
    base = None
    print("cores: %d; %d files x %d records (%.1f MB each)" % (os.cpu_count(), nf, B, os.path.getsize(files[0]) / 1e6))
    for nt in (1, 2, 4, 8, 16, 32, 64):
        if nt > (os.cpu_count() or 1):
            break
        parser = native.BatchParser([(f, vocabs[name_of[f]], T[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)],
                                    n_threads=nt)
        parser.ring = ring
        best = None
        for _rep in range(passes):
            t0 = time.perf_counter()
            n = 0
            for cols in parser.batches(files, B, verify_crc=True):
                n += B
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        rate = n / best
        if base is None:
            base = rate
        print("threads %3d: %9.0f records/s   %.2f ms per batch of %d   parallel efficiency %.0f %%" % (nt, rate, best / (n / B) * 1e3, B, 100.0 * rate / (base * nt)), flush=True)


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 4, int(sys.argv[2]) if len(sys.argv) > 2 else 3, int(sys.argv[3]) if len(sys.argv) > 3 else 3)
