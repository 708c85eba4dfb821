"""Host-side logic: TFRecord codec, vocabulary lookup + FarmHash, Conf parser against the reference's own parse
(golden), spec equality with the oracle, batches, variable inventory."""
import json
import os

import numpy as np
import pytest
import torch

from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.conf.recsys_conf import Conf
from cikm2020_dmt_amd.data_feed import tfrecord
from cikm2020_dmt_amd.data_feed.farmhash import fingerprint64, to_hash_bucket_fast
from cikm2020_dmt_amd.data_feed.index_tables import LookupTables
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.engine import DeviceBatch, _GatherPlan
from cikm2020_dmt_amd.sparse import SparseTensorValue
from cikm2020_dmt_amd.variables import VariableStore
from oracle import dmt_oracle as O
from tests import golden_util as GU
from tests.util import small_specs

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tfrecord_roundtrip_and_crc(tmp_path):
    assert tfrecord.crc32c(b"123456789") == 0xE3069283          # CRC-32C check value
    ex = {"label": np.array([1.0], np.float32), "mask": np.arange(5, dtype=np.float32), "header": [b"a\tb"],
          "clk_seq_sku_7d_50": [b"123", b"unknow"], "clk_seq_sku_7d_50Wts": np.array([1.0, 2.0], np.float32),
          "cnt": np.array([3, -1, 1 << 40], np.int64), "empty": []}
    p = str(tmp_path / "x.tfrecord")
    assert tfrecord.write_records(p, [tfrecord.encode_example(ex)] * 3) == 3
    recs = list(tfrecord.read_records(p, verify_crc=True))
    assert len(recs) == 3
    back = tfrecord.decode_example(recs[1])
    assert back["header"] == [b"a\tb"] and back["clk_seq_sku_7d_50"] == [b"123", b"unknow"]
    assert np.array_equal(back["mask"], ex["mask"]) and np.array_equal(back["cnt"], ex["cnt"])
    raw = bytearray(open(p, "rb").read()); raw[20] ^= 0xFF
    open(p, "wb").write(bytes(raw))
    with pytest.raises(IOError):
        list(tfrecord.read_records(p, verify_crc=True))


def test_farmhash_known_answers():
    # tf.strings.to_hash_bucket_fast(["Hello", "TensorFlow", "2.x"], 3) -> [0, 2, 2]   (TensorFlow API docs)
    assert [to_hash_bucket_fast(s, 3) for s in (b"Hello", b"TensorFlow", b"2.x")] == [0, 2, 2]
    assert fingerprint64(b"") == 0x9AE16A3B2F90404F           # HashLen0to16 of the empty string is k2
    # TensorFlow's own unit test of the op the reference calls (string_to_hash_bucket_op_test.py, testStringToHashBucketsFast: the
    # Fingerprint64 values are written out in its comments; buckets [9, 2, 2, 5] of 10)
    for t, v in ((b"a", 12917804110809363939), (b"b", 11795596070477164822), (b"c", 11430444447143000872), (b"d", 4470636696479570465)):
        assert fingerprint64(t) == v
    assert [to_hash_bucket_fast(t, 10) for t in (b"a", b"b", b"c", b"d")] == [9, 2, 2, 5]
    # BigQuery FARM_FINGERPRINT documentation example (signed 64-bit view of Fingerprint64): lengths 8, 11, 5
    s64 = lambda u: u - (1 << 64) if u >= (1 << 63) else u
    assert [s64(fingerprint64(t)) for t in (b"1footrue", b"2applefalse", b"3true")] == [-1541654101129638711, 2794438866806483259, -4880158226897771312]
    # README values of the `farmhash` Rust crate (hash64) and of pyfarmhash: lengths 11, 3
    assert fingerprint64(b"hello world") == 6381520714923946011 and fingerprint64(b"abc") == 2640714258260161385
    # (Every published vector available offline is <= 16 bytes: the 17-32 / 33-64 / > 64-byte branches are pinned only by the
    #  native-vs-Python cross-check of tests/test_input_native.py -- two implementations by the same author -- and are not reached
    #  by the demo data's ids, decimal strings of <= 12 characters.)
    for n in (1, 3, 4, 7, 8, 16, 17, 32, 33, 64, 65, 200):    # every length class runs and stays within 64 bits
        assert 0 <= fingerprint64(b"x" * n) < (1 << 64)


def test_lookup_tables_against_golden_and_semantics():
    g = json.load(open(os.path.join(GU.GOLDEN, "lookup_golden.json")))
    conf = Conf(os.path.join(ROOT, "cikm2020_dmt_amd/conf/settings/"), "dmt_demo.conf")
    # Sku has no vocabulary upstream: every id hashes into [1, 5e6)
    t = LookupTables(conf, id_tables={})
    for raw, idx in zip(g["item_fea_sku"]["raw"], g["item_fea_sku"]["idx"]):
        assert list(t.inf_transform("item_fea_sku", raw)) == idx
        assert all(1 <= i < 5000000 for i in idx)
    # Time tables have 23 == id_size entries -> no OOV buckets -> default 0
    tt = LookupTables(conf, id_tables={"TimeCart": ["unknow"] + [str(i) for i in range(22)]})
    assert list(tt.inf_transform("cart_seq_ts_12m_10", ["unknow", "3", "6648465"])) == [0, 4, 0]
    # in-vocabulary ids map to their position
    tb = LookupTables(conf, id_tables={"Brand": ["unknow", "184144", "7"]})
    out = tb.inf_transform("item_brand", ["7", "unknow", "184144", "999"])
    assert list(out[:3]) == [2, 0, 1] and 3 <= out[3] < 190000


def test_conf_parser_matches_the_reference_parse():
    g = json.load(open(os.path.join(GU.GOLDEN, "conf_golden.json")))
    conf = Conf(os.path.join(ROOT, "cikm2020_dmt_amd/conf/settings/"), "dmt.conf")
    assert [list(e) for e in conf.embedding_list] == g["embedding_list"]
    assert [list(e) for e in conf.embedding_list_bias] == g["embedding_list_bias"]
    assert [[list(p) for p in grp] for grp in conf.attention_embed_pairs] == g["attention_embed_pairs"]
    assert conf.attention_embed_seq_ts == g["attention_embed_seq_ts"]
    assert conf.weight_ctr == g["weight_ctr"] and conf.weight_ecvr == g["weight_ecvr"]
    sp = conf.to_spec()
    assert sp["hidden_units_bottom"] == g["hidden_units_bottom"] and sp["hidden_units_task"] == g["hidden_units_task"]
    assert sp["d_model"] == int(g["model"]["transformer_d_model"]) and sp["d_ff"] == int(g["model"]["transformer_d_ff"])
    assert conf.zero_pad == g["model"]["zero_pad"] == "true"


def test_conf_transformer_options_the_engine_takes_and_refuses():
    """position_learn (dmt.conf) and position_sin_cos reach the engine's spec; the options that would need other graphs are refused by
    name instead of being ignored."""
    conf = Conf(os.path.join(ROOT, "cikm2020_dmt_amd/conf/settings/"), "dmt.conf")
    assert conf.to_spec()["position_encoding_method"] == "position_learn"
    conf.position_encoding_method = "position_sin_cos"
    conf.is_decoder_add_pos_emb = True
    conf.is_trans_out_concat_item = True
    sp = conf.to_spec()
    assert sp["position_encoding_method"] == "position_sin_cos" and sp["is_decoder_add_pos_emb"] is True and sp["is_trans_out_concat_item"] is True
    assert S.mmoe_input_width(sp) == S.mmoe_input_width(dict(sp, is_trans_out_concat_item=False)) + 3 * sp["d_model"]
    conf.is_trans_out_by_mlp = True
    sp2 = conf.to_spec()
    assert sp2["is_trans_out_by_mlp"] is True and S.mmoe_input_width(sp2) == S.mmoe_input_width(dict(sp, is_trans_out_concat_item=False))
    conf.is_trans_out_by_mlp = False
    st = VariableStore(S.scaled_spec(sp, {"Sku": 500, "Brand": 50, "Shopid": 50, "Cid3": 20}), "cpu", torch.float32, seed=0)
    assert not any("position_learn" in k for k in st.state_dict())
    conf.is_trans_input_by_mlp = True
    assert conf.to_spec()["is_trans_input_by_mlp"] is True
    conf.num_blocks_encode, conf.num_blocks_decode = 2, 3
    sp3 = conf.to_spec()
    assert (sp3["num_blocks_encode"], sp3["num_blocks_decode"]) == (2, 3)
    for attr, val in (("position_encoding_method", "time_add"), ("position_encoding_method", "time_concat"), ("num_blocks_encode", 0)):
        c2 = Conf(os.path.join(ROOT, "cikm2020_dmt_amd/conf/settings/"), "dmt.conf")
        setattr(c2, attr, val)
        with pytest.raises(NotImplementedError):
            c2.to_spec()


def test_product_spec_equals_oracle_spec():
    for suffix in ("12m_50", "12m_10"):
        so, sp = O.default_spec(suffix), S.default_spec(suffix)
        for k in so:
            a, b = so[k], sp[k]
            if k in ("embedding_list", "embedding_list_bias"):
                a, b = [tuple(x) for x in a], [tuple(x) for x in b]
            if k == "attention_embed_pairs":
                a, b = [[tuple(p) for p in g] for g in a], [[tuple(p) for p in g] for g in b]
            assert a == b, k
    e = S.e64_spec()
    assert e["d_model"] == 320 and e["d_ff"] == 1280 and S.mmoe_input_width(e) == 615 + 5 * 64 + 3 * 6 * 64 + 3 * 320


def test_variable_inventory_matches_appendix_b():
    so, sp = small_specs()
    st = VariableStore(sp, "cpu", torch.float32, seed=0)
    sd = st.state_dict()
    shp = O.param_shapes(so)
    assert set(sd) == set(shp)
    for k, v in shp.items():
        assert tuple(sd[k].shape) == tuple(v), k
    assert st.P >= 3418515 - 166883052 * 0        # dense parameter count of SURVEY.md Appendix B fits the arena
    P = O.init_params(so, seed=3)
    st.load_state(P)
    back = st.state_dict()
    assert max(np.abs(back[k] - P[k].astype(np.float32)).max() for k in P) == 0.0
    # packed leaves really are contiguous views
    w = st.leaf["mmoe_layers/l0_cat_weights"]
    assert w.shape == (1199, 4 * 512 + 8) and w.is_contiguous()
    plan = _GatherPlan(sp, st)
    assert (plan.K, plan.interest_off, plan.bias_off, plan.ldz) == (1199, 959, 1200, 1224)


def test_sparse_and_device_batch_layout():
    sp_ = SparseTensorValue.from_rows([[5, 6, 7], [], [9]], np.int64)
    assert sp_.dense_shape == (3, 3) and list(sp_.lengths()) == [3, 0, 1]
    dense, lens = sp_.to_padded(4)
    assert dense.tolist() == [[5, 6, 7, 0], [0, 0, 0, 0], [9, 0, 0, 0]]
    _so, sp = small_specs()
    inputs, mask, label = make_batch(sp, 9, seed=1, lengths="ragged", weights="random")
    b = DeviceBatch.from_inputs(inputs, sp, "cpu", mask, label)
    col = b.feats["clk_seq_sku_7d_50"]
    assert col.idx.dtype == torch.int32 and col.idx.shape == (9, col.T) and col.wts is not None
    ref_lens = inputs["clk_seq_sku_7d_50"].lengths()
    assert col.lens.tolist() == ref_lens.tolist()
    for r in range(9):                                   # left aligned, zero padded
        assert (col.idx[r, ref_lens[r]:] == 0).all()
    inputs2, _, _ = make_batch(sp, 4, seed=1, lengths="full", weights="ones")
    b2 = DeviceBatch.from_inputs(inputs2, sp, "cpu")
    assert b2.feats["ord_seq_sku_12m_50"].wts is None and b2.feats["ord_seq_sku_12m_50"].T == 50


def test_demo_fixture_shape_facts():
    demo = GU.load_demo()
    assert demo["features"].shape == (474, 615) and demo["mask"].shape == (474, 5)
    cls = demo["mask"].argmax(1)
    assert np.bincount(cls, minlength=5).tolist() == [434, 1, 21, 4, 14]       # SURVEY.md §8c label distribution
    assert demo["l_clk_seq_sku_7d_50"].max() == 50 and demo["l_ord_seq_sku_12m_10"].max() == 10
    assert demo["l_clk_seq_sku_7d_50"].min() >= 1


def test_deferred_weight_gradient_queue_semantics():
    """ops.begin_deferred_wgrads / run_deferred_wgrads: closures run once, in collection order, in one or several batches (the
    data-parallel step launches half of them beside the all_to_all and the rest beside the all-gather)."""
    from cikm2020_dmt_amd import ops
    st, other = ops.StepState(), ops.StepState()
    ops.activate(st)
    assert ops.run_deferred_wgrads() == 0 and ops.deferred_wgrads_pending() == 0
    seen = []
    ops.begin_deferred_wgrads()
    for i in range(5):
        st.deferred.append(lambda i=i: seen.append(i))
    ops.activate(other)                         # another engine's state: sees nothing of the first one's collection
    assert ops.deferred_wgrads_pending() == 0 and ops.run_deferred_wgrads() == 0 and seen == []
    ops.activate(st)
    assert ops.deferred_wgrads_pending() == 5
    assert ops.run_deferred_wgrads(upto=3) == 3 and seen == [0, 1, 2] and ops.deferred_wgrads_pending() == 2
    assert ops.run_deferred_wgrads() == 2 and seen == [0, 1, 2, 3, 4]
    assert st.deferred is None and ops.run_deferred_wgrads() == 0      # collection is off again: backward launches at once
    ops.begin_deferred_wgrads()
    assert ops.run_deferred_wgrads(upto=4) == 0 and st.deferred is None
    assert ops.StepState(64).min_rows() == 64 and ops.StepState().min_rows() == ops.WGRAD320_MIN_ROWS
    ops.activate(None)


def test_index_group_is_the_default_group_without_rccl():
    """parallel.index_group(): a second communicator exists only for RCCL; without a process group (or with gloo) the index plane
    uses the default one (None)."""
    from cikm2020_dmt_amd import parallel
    assert parallel.index_group() is None


def test_tfadam_refuses_an_increasing_learning_rate_schedule():
    """The exact lazy-row replay assumes a non-increasing piecewise schedule (cikm2020_dmt_amd/optim.py); a warm-up schedule is rejected
    before any tensor is touched."""
    from cikm2020_dmt_amd.optim import TFAdam
    with pytest.raises(ValueError, match="non-increasing"):
        TFAdam(object(), learning_rate=(1e-4, 1e-3), step_boundary=(100,))


def test_tf_checkpoint_bundle_container_round_trip_and_format_invariants(tmp_path):
    """cikm2020_dmt_amd/tf_bundle.py: the TensorFlow checkpoint V2 files (LevelDB-format index table + data shard) written and read
    without TensorFlow.  Checked here: round trip of names / shapes / dtypes / values over several table blocks; the published format
    constants (footer magic, 48-byte footer, 5-byte block trailers with masked CRC-32C, header entry under the empty key, entries in
    bytewise key order with prefix compression restarting every 16 keys); corruption of either file is detected.  (Parity with files
    written by TensorFlow itself is unpinned: neither TensorFlow nor a TF-written checkpoint is available.)"""
    import struct
    from cikm2020_dmt_amd import tf_bundle as TB
    # RFC 3720 known answer through the masking rule of crc32c.h: Mask(crc) = ((crc >> 15) | (crc << 17)) + 0xa282ead8
    crc = 0x8A9136AA                                            # CRC-32C of 32 zero bytes (RFC 3720 B.4)
    assert TB.masked_crc32c(bytes(32)) == ((((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF)
    rng = np.random.default_rng(0)
    tensors = {"DnnModel/embedding_trans/Sku/embedding": rng.standard_normal((3000, 32)).astype(np.float32),
               "DnnModel/click/click-output/biases": np.array([0.25], np.float32), "global_step": np.array(7, np.int64),
               "DnnModel/scalar_like": np.zeros((0, 4), np.float32)}
    for i in range(5000):                                       # many small variables: several 256 KB index blocks, shared key prefixes
        tensors["DnnModel/mmoe_layers/expert-%04d/weights" % i] = rng.standard_normal((3,)).astype(np.float32)
    prefix = str(tmp_path / "model.ckpt-7")
    TB.write_bundle(prefix, tensors)
    assert sorted(os.listdir(tmp_path)) == ["model.ckpt-7.data-00000-of-00001", "model.ckpt-7.index"]
    back = TB.read_bundle(prefix)
    assert set(back) == set(tensors)
    for k, v in tensors.items():
        assert back[k].dtype == v.dtype and back[k].shape == v.shape and np.array_equal(back[k], v), k
    raw = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", raw[-8:])[0] == 0xDB4775248B80FB57 and len(raw) > 48
    items = TB.read_table(prefix + ".index")
    keys = [k for k, _v in items]
    assert keys[0] == b"" and keys == sorted(keys) and len(keys) == len(tensors) + 1
    hdr = TB._parse_pb(items[0][1])
    assert hdr[1] == [1] and TB._parse_pb(hdr[3][0])[1] == [1]                     # num_shards = 1, version.producer = 1
    e = TB._parse_pb(dict(items)[b"DnnModel/embedding_trans/Sku/embedding"])
    assert e[1] == [1] and e[5] == [3000 * 32 * 4]                                  # DT_FLOAT, byte size
    assert [TB._parse_pb(d)[1][0] for d in TB._parse_pb(e[2][0])[2]] == [3000, 32]
    assert os.path.getsize(prefix + ".data-00000-of-00001") == sum(v.nbytes for v in tensors.values())
    # corruption is detected: a flipped byte in the data shard (tensor checksum) and in the index (block checksum)
    with open(prefix + ".data-00000-of-00001", "r+b") as f:
        f.seek(100); b = f.read(1); f.seek(100); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match="checksum"):
        TB.read_bundle(prefix)
    TB.write_bundle(prefix, tensors)
    with open(prefix + ".index", "r+b") as f:
        f.seek(200); b = f.read(1); f.seek(200); f.write(bytes([b[0] ^ 1]))
    with pytest.raises(ValueError, match="checksum"):
        TB.read_bundle(prefix)
    with pytest.raises(ValueError):
        TB.write_table(str(tmp_path / "t"), [(b"b", b""), (b"a", b"")])


def test_bench_gpus_n_without_a_launcher_starts_n_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment re-executes itself under torch.distributed.run with two ranks
    (here, without a GPU, both then refuse to run: the message must come from the ranks, and a one-rank run must never happen), and a
    launcher whose WORLD_SIZE disagrees with --gpus is refused -- also for --gpus N under WORLD_SIZE=1."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "LOCAL_WORLD_SIZE")}
    if torch.cuda.is_available():
        env["CUDA_VISIBLE_DEVICES"] = env["HIP_VISIBLE_DEVICES"] = ""
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "local_rank: 0" in out.stderr or "local_rank: 1" in out.stderr, out.stderr[-2000:]        # torchrun's failure report: ranks existed
    assert out.stderr.count("bench.py needs an MI355X") >= 1
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and "WORLD_SIZE=1 but --gpus 8" in out.stderr
