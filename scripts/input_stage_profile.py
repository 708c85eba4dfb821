"""Where the input stage's time per batch goes: the native parse alone (per thread count), page-locked vs pageable output, the upload.
    python scripts/input_stage_profile.py
"""
import os
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cikm2020_dmt_amd import spec as S                                            # noqa: E402
from cikm2020_dmt_amd.data_feed import native                                     # noqa: E402
from cikm2020_dmt_amd.data_feed.synthetic import make_batch, write_records_file   # noqa: E402
from cikm2020_dmt_amd.engine import DeviceBatch                                   # noqa: E402


def main():
    sp = S.e64_spec()
    B, nf = 4096, 4
    tmp = tempfile.mkdtemp(prefix="dmt_records_")
    files = [os.path.join(tmp, "part-r-%05d" % i) for i in range(nf)]
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(nf) as pool:
        pool.map(write_records_file, [("e64", 0, B, 777000 + i, "zipf", f) for i, f in enumerate(files)])
    emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
    feats = list(dict.fromkeys(e[3] for e in emb))
    name_of = {e[3]: e[0] for e in reversed(emb)}
    vocabs = {}
    for (name, nrows, _d, _f, _s) in emb:
        vocabs.setdefault(name, native.Vocab(["unknow"], nrows) if nrows > 23 else native.Vocab(["unknow"] + [str(i) for i in range(1, nrows)], nrows))
    probe, _m, _l = make_batch(sp, 2, seed=1, lengths="full", law="zipf")
    T = {f: max(int(probe[f].dense_shape[1]), 1) for f in feats}
    dev = torch.device("cuda:0")
    for pinned in (False, True):
        for nt in (4, 8, 16, 32, 64, 128):
            parser = native.BatchParser([(f, vocabs[name_of[f]], T[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)],
                                        n_threads=nt)
            parser.pinned = pinned
            for rep in range(2):
                t0 = time.perf_counter()
                n = 0
                cols_keep = []
                for cols in parser.batches(files, B, verify_crc=True):
                    n += 1
                    cols_keep.append(cols)
                t1 = time.perf_counter()
            t2 = time.perf_counter()
            for cols in cols_keep:
                b = DeviceBatch.from_columns(cols, sp, dev)
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            print("pinned=%d threads=%3d: parse %.2f ms/batch, from_columns + upload %.2f ms/batch" % (pinned, nt, (t1 - t0) / n * 1e3, (t3 - t2) / n * 1e3), flush=True)


if __name__ == "__main__":
    main()
