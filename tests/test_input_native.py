"""libdmt_input.so (C++ input stage) against the Python restatement of the same pipeline (data_feed/tfrecord.py,
farmhash.py, index_tables.py -- the oracle side here) and against published known answers.  Bit-exact: integer / byte work."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from cikm2020_dmt_amd.data_feed import farmhash, native, tfrecord
from cikm2020_dmt_amd.data_feed.index_tables import LookupTables
from cikm2020_dmt_amd.sparse import SparseTensorValue

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_symbol_of_the_header():
    hdr = open(os.path.join(ROOT, "include", "dmt_input.h")).read()
    declared = set(re.findall(r"\b(dmt_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"dmt_feature_spec"}
    lib = native.load()
    for sym in sorted(declared):
        assert hasattr(lib, sym), sym
    assert declared == set(native.EXPORTED_SYMBOLS)


def test_crc32c_known_answers_and_python_equivalence():
    # RFC 3720 B.4 test vectors
    assert native.crc32c(b"\x00" * 32) == 0x8A9136AA
    assert native.crc32c(b"\xff" * 32) == 0x62A8AB43
    assert native.crc32c(bytes(range(32))) == 0x46DD794E
    assert native.crc32c(b"123456789") == 0xE3069283
    rng = np.random.default_rng(0)
    for n in list(range(0, 40)) + [255, 256, 1000, 4097]:
        b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert native.crc32c(b) == tfrecord.crc32c(b)
        assert native.masked_crc32c(b) == tfrecord.masked_crc32c(b)


def test_fingerprint64_equals_restatement_for_every_length_class():
    rng = np.random.default_rng(1)
    for n in list(range(0, 140)) + [191, 192, 193, 255, 256, 257, 1000]:
        for _ in range(3):
            b = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
            assert native.fingerprint64(b) == farmhash.fingerprint64(b), n
    # documented example: tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) -> [0, 2, 2]
    assert [native.fingerprint64(s) % 3 for s in (b"Hello", b"TensorFlow", b"2.x")] == [0, 2, 2]


def _conf_like():
    class Cf:
        embedding_list = [("Sku", 1000, 8, "item_sku", "i"), ("Sku", 1000, 8, "seq_sku", "u"), ("Cid", 7, 4, "seq_cid", "u"), ("Time", 4, 4, "seq_ts", "u")]
        embedding_list_bias = [("CidB", 7, 3, "item_cid", "i")]
    return Cf()


def _vocab_lists():
    return {"Sku": ["unknow"] + ["s%d" % i for i in range(300)], "Cid": ["unknow", "c1", "c2", "c3", "c2"], "Time": ["unknow", "1", "2", "3"],
            "CidB": ["unknow", "c1", "c2"]}


def test_vocab_lookup_equals_python_lookup_tables():
    cf, voc = _conf_like(), _vocab_lists()
    py = LookupTables(cf, id_tables=voc)
    sizes = {"Sku": 1000, "Cid": 7, "Time": 4, "CidB": 7}
    nat = {k: native.Vocab(v, sizes[k]) for k, v in voc.items()}
    probes = ["unknow", "s0", "s299", "s300", "zzz", "", "c2", "c9", "1", "7", "a" * 70, "éè"]
    for name in voc:
        exp = py.lookup_embedding(name, probes)
        got = [nat[name].lookup(s) for s in probes]
        assert got == [int(x) for x in exp], name
    assert nat["Time"].lookup("99") == 0                # no OOV buckets -> default 0
    assert nat["Cid"].lookup("c2") == 2                 # first occurrence of a duplicated key wins


def _make_examples(n, rng):
    voc = _vocab_lists()
    exs = []
    for i in range(n):
        L = int(rng.integers(1, 6))
        pick = lambda pool, k: [(rng.choice(pool) if rng.random() < 0.8 else "oov%d" % rng.integers(0, 50)).encode() for _ in range(k)]
        f = {
            "item_sku": pick(voc["Sku"], 1), "item_skuWts": np.ones(1, np.float32),
            "seq_sku": pick(voc["Sku"], L), "seq_skuWts": rng.random(L).astype(np.float32),
            "seq_cid": pick(voc["Cid"], L), "seq_cidWts": np.ones(L, np.float32),
            "seq_ts": pick(voc["Time"], L), "seq_tsWts": np.ones(L, np.float32),
            "item_cid": pick(voc["CidB"], 1), "item_cidWts": np.ones(1, np.float32),
            "features": rng.uniform(-1, 1, 6).astype(np.float32), "mask": np.eye(5, dtype=np.float32)[int(rng.integers(0, 5))],
            "label": np.array([float(rng.integers(0, 2))], np.float32), "header": [b"a\tb\tc"], "unused_int": np.array([3, -1], np.int64),
        }
        if i % 7 == 3:
            del f["seq_cidWts"]                         # optional weights
        exs.append(f)
    return exs


def test_tfrecord_and_batch_parse_equal_the_python_pipeline(tmp_path):
    rng = np.random.default_rng(2)
    exs = _make_examples(53, rng)
    path = str(tmp_path / "part-r-00000")
    assert tfrecord.write_records(path, [tfrecord.encode_example(e) for e in exs]) == 53
    recs_py = list(tfrecord.read_records(path, verify_crc=True))
    recs_nat = list(native.read_records(path, verify_crc=True))
    assert recs_py == recs_nat
    # python pipeline: decode -> SparseTensorValue -> transform_id2index -> to_padded
    cf, voc = _conf_like(), _vocab_lists()
    py_tables = LookupTables(cf, id_tables=voc)
    dec = [tfrecord.decode_example(r) for r in recs_py]
    T = 6
    feats = ["item_sku", "seq_sku", "seq_cid", "seq_ts", "item_cid"]
    emb_of = {"item_sku": "Sku", "seq_sku": "Sku", "seq_cid": "Cid", "seq_ts": "Time", "item_cid": "CidB"}
    sizes = {"Sku": 1000, "Cid": 7, "Time": 4, "CidB": 7}
    nat_voc = {k: native.Vocab(v, sizes[k]) for k, v in voc.items()}
    parser = native.BatchParser([(f, nat_voc[emb_of[f]], T) for f in feats], [("features", 6), ("mask", 5), ("label", 1)], n_threads=3)
    got = parser.parse(recs_nat)
    for f in feats:
        ids = SparseTensorValue.from_rows([d[f] for d in dec], dtype=object)
        idx = SparseTensorValue(ids.indices, py_tables.inf_transform(f, list(ids.values)), ids.dense_shape)
        exp_idx, exp_lens = idx.to_padded(T)
        assert np.array_equal(got[f], exp_idx.astype(np.int32)), f
        assert np.array_equal(got[f + "/lens"], exp_lens.astype(np.int32)), f
        w = SparseTensorValue.from_rows([d.get(f + "Wts", np.zeros(0, np.float32)) for d in dec], dtype=np.float32)
        exp_w, _ = w.to_padded(T)
        assert np.array_equal(got[f + "Wts"], exp_w.astype(np.float32)), f
    for k, n in (("features", 6), ("mask", 5), ("label", 1)):
        assert np.array_equal(got[k], np.stack([d[k] for d in dec]).astype(np.float32).reshape(-1, n))
    # batching helper: same rows, remainder kept
    bs = list(parser.batches([path], 20))
    assert [b["label"].shape[0] for b in bs] == [20, 20, 13]
    assert np.array_equal(np.concatenate([b["seq_sku"] for b in bs]), got["seq_sku"])


def test_worker_pool_serves_concurrent_callers_growing_thread_counts_and_a_forked_child(tmp_path):
    """The library's parked worker pool (one job at a time, callers serialise): several Python threads parsing at once with different
    thread counts (the pool grows between jobs), every result equal to the single-threaded parse; an error raised inside a worker comes
    back as the caller's error; a fork()ed child -- which has none of the parent's threads -- starts its own workers."""
    import threading
    rng = np.random.default_rng(7)
    exs = _make_examples(97, rng)
    recs = [tfrecord.encode_example(e) for e in exs]
    voc = _vocab_lists()
    sizes = {"Sku": 1000, "Cid": 7, "Time": 4, "CidB": 7}
    emb_of = {"item_sku": "Sku", "seq_sku": "Sku", "seq_cid": "Cid", "seq_ts": "Time", "item_cid": "CidB"}
    feats = ["item_sku", "seq_sku", "seq_cid", "seq_ts", "item_cid"]

    def make(nt, T=6):
        nat_voc = {k: native.Vocab(v, sizes[k]) for k, v in voc.items()}
        return native.BatchParser([(f, nat_voc[emb_of[f]], T) for f in feats], [("features", 6), ("mask", 5), ("label", 1)], n_threads=nt)
    ref = make(1).parse(recs)
    out, errs = {}, []

    def work(i, nt):
        try:
            p = make(nt)
            for _ in range(20):
                got = p.parse(recs)
                for k, v in ref.items():
                    if isinstance(v, np.ndarray) and not np.array_equal(got[k], v):
                        raise AssertionError("thread %d (%d workers): column %s differs" % (i, nt, k))
            out[i] = True
        except Exception as e:          # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=work, args=(i, nt)) for i, nt in enumerate((2, 5, 3, 8, 4, 16))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errs, errs
    assert len(out) == 6
    # a failure inside a pool worker reaches the caller (the over-long list is in the LAST record: not the caller's own chunk)
    with pytest.raises(native.InputError):
        make(4, T=1).parse(recs)
    assert np.array_equal(make(4).parse(recs)["seq_sku"], ref["seq_sku"])      # the pool is usable after the failure
    # fork: the child has the pool object but none of its threads
    pid = os.fork()
    if pid == 0:
        ok = 1
        try:
            got = make(4).parse(recs)
            ok = 0 if all(np.array_equal(got[k], v) for k, v in ref.items() if isinstance(v, np.ndarray)) else 2
        finally:
            os._exit(ok)
    _, status = os.waitpid(pid, 0)
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0, status


def test_reader_rejects_corruption_and_parser_rejects_overlong_lists(tmp_path):
    rng = np.random.default_rng(3)
    exs = _make_examples(4, rng)
    path = str(tmp_path / "rec")
    tfrecord.write_records(path, [tfrecord.encode_example(e) for e in exs])
    raw = bytearray(open(path, "rb").read())
    raw[20] ^= 0x40                                       # flip a payload bit of the first record
    bad = str(tmp_path / "bad")
    open(bad, "wb").write(bytes(raw))
    with pytest.raises(native.InputError):
        list(native.read_records(bad, verify_crc=True))
    assert len(list(native.read_records(bad, verify_crc=False))) == 4      # as TF with check disabled
    open(bad, "wb").write(bytes(raw[:-3]))                # truncated tail
    with pytest.raises(native.InputError):
        list(native.read_records(bad, verify_crc=False))
    voc = native.Vocab(["unknow", "a"], 10)
    parser = native.BatchParser([("seq_sku", voc, 2)], [("label", 1)], n_threads=1)
    long_ex = tfrecord.encode_example({"seq_sku": [b"a", b"b", b"c"], "label": np.array([1.0], np.float32)})
    with pytest.raises(native.InputError):
        parser.parse([long_ex])
    with pytest.raises(native.InputError):
        parser.parse([b"\x0a\xff\xff"])                    # malformed wire data


REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF + "/jd_recsys_demo"), reason="the reference's demo TFRecords only exist in the build container")
def test_native_parse_of_the_reference_demo_records_equals_the_committed_golden_fixture():
    """The 474 jd_recsys_demo examples through libdmt_input.so (reader + decode + vocabulary lookup with the reference's own
    conf/idtables vocabularies) must reproduce tests/golden/demo474.npz, which make_golden.py produced with the Python pipeline."""
    import glob
    import runpy
    from cikm2020_dmt_amd.conf.recsys_conf import Conf
    from tests import golden_util as GU
    conf = Conf(os.path.join(ROOT, "cikm2020_dmt_amd/conf/settings/"), "dmt_demo.conf")
    demo = GU.load_demo()
    emb = list(conf.embedding_list) + list(conf.embedding_list_bias)
    vocabs, feat_vocab, sizes = {}, {}, {}
    for e in emb:
        name, id_size, feat = e[0], int(e[1]), e[3]
        if name not in vocabs:
            fn = os.path.join(REF, "DMT_code/conf/idtables", name + ".py")
            keys = list(runpy.run_path(fn)["ID_TABLES"][name]) if os.path.exists(fn) else ["unknow"]
            vocabs[name] = native.Vocab(keys, id_size)
        feat_vocab.setdefault(feat, vocabs[name])
    feats = list(feat_vocab)
    T = {f: int(demo["l_" + f].max()) for f in feats}
    parser = native.BatchParser([(f, feat_vocab[f], T[f]) for f in feats], [("features", demo["features"].shape[1]), ("mask", 5), ("label", 1)])
    files = sorted(glob.glob(REF + "/jd_recsys_demo/*/test_ord/*/data/part-r-*"))
    recs = [r for fn in files for r in native.read_records(fn, verify_crc=True)]
    assert len(recs) == 474
    got = parser.parse(recs)
    assert np.array_equal(got["features"], demo["features"]) and np.array_equal(got["mask"], demo["mask"])
    assert np.array_equal(got["label"][:, 0], demo["label"])
    for f in feats:
        lens = demo["l_" + f].astype(np.int32)
        assert np.array_equal(got[f + "/lens"], lens), f
        valid = np.arange(T[f])[None, :] < lens[:, None]
        assert np.array_equal(got[f][valid], demo["v_" + f]), f
        assert not got[f][~valid].any()
        w = demo["w_" + f] if ("w_" + f) in demo else np.ones(int(lens.sum()), np.float32)
        assert np.array_equal(got[f + "Wts"][valid], w), f


def test_device_batch_from_native_columns_equals_from_inputs():
    """End of the input stage: records written from synthetic model inputs -> native parse -> DeviceBatch.from_columns must be the
    batch DeviceBatch.from_inputs builds from the same examples (CPU tensors; the device copy is plumbing)."""
    import torch
    from cikm2020_dmt_amd import spec as S
    from cikm2020_dmt_amd.data_feed.synthetic import make_batch
    from cikm2020_dmt_amd.engine import DeviceBatch
    from tests.util import small_specs
    _so, sp = small_specs()
    B = 9
    inputs, mask, label = make_batch(sp, B, seed=11, lengths="ragged", weights="random")
    emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
    feats = list(dict.fromkeys(e[3] for e in emb))
    # string ids: vocabulary position i <-> "k<i>" (every index of the synthetic batch is in-vocabulary)
    vocabs = {}
    for (name, rows, _d, _f, _s) in emb:
        vocabs.setdefault(name, native.Vocab(["k%d" % i for i in range(rows)], rows))
    recs = []
    rows_of = {f: inputs[f].rows() for f in feats}
    wrows_of = {f: inputs[f + "Wts"].rows() for f in feats}
    for b in range(B):
        ex = {"features": inputs["features"][b].astype(np.float32), "mask": mask[b].astype(np.float32), "label": np.array([label[b]], np.float32)}
        for f in feats:
            ex[f] = [("k%d" % int(i)).encode() for i in rows_of[f][b]]
            ex[f + "Wts"] = np.asarray(wrows_of[f][b], np.float32)
        recs.append(tfrecord.encode_example(ex))
    max_lens = {f: max(int(inputs[f].dense_shape[1]), 1) for f in feats}
    name_of = {e[3]: e[0] for e in reversed(emb)}
    parser = native.BatchParser([(f, vocabs[name_of[f]], max_lens[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)])
    got = DeviceBatch.from_columns(parser.parse(recs), sp, "cpu")
    ref = DeviceBatch.from_inputs(inputs, sp, "cpu", mask=mask, label=label, pad_to=max_lens)
    assert got.B == ref.B and torch.equal(got.dense, ref.dense) and torch.equal(got.mask, ref.mask) and torch.equal(got.label, ref.label)
    for f in feats:
        a, r = got.feats[f], ref.feats[f]
        assert a.T == r.T and torch.equal(a.idx, r.idx) and torch.equal(a.lens, r.lens), f
        assert (a.wts is None) == (r.wts is None), f
        if a.wts is not None:
            assert torch.equal(a.wts, r.wts), f
