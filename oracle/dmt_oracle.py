"""CPU ORACLE (test infrastructure, NOT product code) -- numpy restatement of the DMT hot path.

PARITY UNPINNED: the reference (guyulongcs/CIKM2020_DMT) is TensorFlow-1.12 graph code; TensorFlow is
not installable here and the reference ships no tests / golden vectors, so
this file restates the reference's algorithm (and, where the arithmetic lives inside the un-vendored
`tensorflow==1.12` dependency -- README.md:10-14 -- TF's published op semantics) and is cross-checked
only against a second, independent restatement (oracle/dmt_oracle_torch.py) and finite differences.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.

All citations are relative to /root/reference/DMT_code/.  Everything is computed in the dtype of the
parameters handed in (tests use float64).  Loops are deliberately literal.

Input batch (the int-index boundary, i.e. AFTER data_feed/index_tables.py:37-45 transform_id2index):
    inputs['features']            float [B, feature_dimension]
    inputs[f], inputs[f+'Wts']    objects with .indices [nnz,2] int64, .values [nnz], .dense_shape (B,T)
Parameters: dict  TF-variable-name (without the 'DnnModel/' prefix) -> ndarray.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np

PADDING_NUM = float(-2 ** 32 + 1)  # model/net/TransformerModel_util.py:81


class SparseTensorValue:
    """Minimal stand-in for tf.SparseTensorValue (row-major ordered indices)."""

    def __init__(self, indices, values, dense_shape):
        self.indices = np.asarray(indices, dtype=np.int64).reshape(-1, 2)
        self.values = np.asarray(values)
        self.dense_shape = tuple(int(x) for x in dense_shape)


def sparse_from_lists(rows: List[List], dtype) -> SparseTensorValue:
    """What tf.parse_example's VarLenFeature batching yields: dense_shape = (B, max len)."""
    idx, val = [], []
    T = 0
    for b, r in enumerate(rows):
        T = max(T, len(r))
        for t, v in enumerate(r):
            idx.append((b, t))
            val.append(v)
    return SparseTensorValue(np.array(idx, dtype=np.int64).reshape(-1, 2), np.array(val, dtype=dtype),
                             (len(rows), T))


def sparse_to_dense(sp, default=0):
    """tf.sparse.to_dense (mmoe_transformer_unbias.py:146,156)."""
    out = np.full(sp.dense_shape, default, dtype=np.asarray(sp.values).dtype)
    ind = np.asarray(sp.indices)
    if len(ind):
        out[ind[:, 0], ind[:, 1]] = np.asarray(sp.values)
    return out


# --------------------------------------------------------------------------------------------- spec
def default_spec(ord_len_name: str = "12m_50") -> dict:
    """Hyper-parameters of conf/settings/dmt.conf:15-129 in plain-dict form.

    ord_len_name='12m_10' gives the feature names the demo TFRecords actually carry.
    """
    o = ord_len_name
    emb = ("Sku:5000000:32:item_fea_sku:i#Cid2:500:8:item_c2:i#Cid3:12000:8:item_c3:i#Brand:190000:16:item_brand:i#"
           "Shopid:230000:16:item_shop:i#Sku:5000000:32:clk_seq_sku_7d_50:u#TimeClick:23:8:clk_seq_ts_7d_50:u#"
           "Cid2:500:8:clk_seq_c2_7d_50:u#Cid3:12000:8:clk_seq_c3_7d_50:u#Brand:190000:16:clk_seq_brand_7d_50:u#"
           "Shopid:230000:16:clk_seq_shop_7d_50:u#Sku:5000000:32:ord_seq_sku_%s:u#TimeOrder:23:8:ord_seq_ts_%s:u#"
           "Cid2:500:8:ord_seq_c2_%s:u#Cid3:12000:8:ord_seq_c3_%s:u#Brand:190000:16:ord_seq_brand_%s:u#"
           "Shopid:230000:16:ord_seq_shop_%s:u#Sku:5000000:32:cart_seq_sku_12m_10:u#TimeCart:23:8:cart_seq_ts_12m_10:u#"
           "Cid2:500:8:cart_seq_c2_12m_10:u#Cid3:12000:8:cart_seq_c3_12m_10:u#Brand:190000:16:cart_seq_brand_12m_10:u#"
           "Shopid:230000:16:cart_seq_shop_12m_10:u") % (o, o, o, o, o, o)
    att = ("clk_seq_sku_7d_50:item_fea_sku#clk_seq_c2_7d_50:item_c2#clk_seq_c3_7d_50:item_c3#"
           "clk_seq_brand_7d_50:item_brand#clk_seq_shop_7d_50:item_shop|ord_seq_sku_%s:item_fea_sku#"
           "ord_seq_c2_%s:item_c2#ord_seq_c3_%s:item_c3#ord_seq_brand_%s:item_brand#ord_seq_shop_%s:item_shop|"
           "cart_seq_sku_12m_10:item_fea_sku#cart_seq_c2_12m_10:item_c2#cart_seq_c3_12m_10:item_c3#"
           "cart_seq_brand_12m_10:item_brand#cart_seq_shop_12m_10:item_shop") % (o, o, o, o, o)
    ts = "clk_seq_ts_7d_50|ord_seq_ts_%s|cart_seq_ts_12m_10" % o
    bias = "Cid2:500:5:item_c2:i#Cid3:12000:5:item_c3:i#Cid2:500:5:near_expo_seq_c2:u#Cid3:12000:5:near_expo_seq_c3:u"

    def parse_emb(s):  # conf/recsys_conf.py:274-284 get_emb
        out = []
        for e in s.split("#"):
            f = e.split(":")
            out.append((f[0], int(f[1]), int(f[2]), f[3], f[4]))
        return out

    return dict(
        embedding_list=parse_emb(emb),
        embedding_list_bias=parse_emb(bias),
        attention_embed_pairs=[[tuple(p.split(":")) for p in grp.split("#")] for grp in att.split("|")],
        attention_embed_seq_ts=[x.strip() for x in ts.split("|")],
        feature_dimension=615, d_model=80, d_ff=320, num_heads=4, maxlen_k=50,
        num_blocks_encode=1, num_blocks_decode=1,
        hidden_units_bottom=[512, 256, 128], hidden_units_task=[32], num_experts=4, num_tasks=2,
        hidden_units_bias=[32, 16], output_units=1,
        weight_ctr=[1.0, 15.0, 15.0, 15.0, 15.0], weight_ecvr=[1.0, 1.0, 1.0, 400.0, 400.0],
        loss_weight=[1.0, 1.0], tie_ffn=True,
    )


def scaled_spec(spec: dict, rows_scale: dict) -> dict:
    """Same model with smaller vocabularies (tests).  rows_scale: table name -> rows."""
    s = dict(spec)
    s["embedding_list"] = [(n, rows_scale.get(n, r), d, f, side) for (n, r, d, f, side) in spec["embedding_list"]]
    s["embedding_list_bias"] = [(n, rows_scale.get(n, r), d, f, side) for (n, r, d, f, side) in spec["embedding_list_bias"]]
    return s


def table_shapes(spec) -> Dict[str, tuple]:
    out = {}
    for (name, rows, dim, _f, _s) in spec["embedding_list"]:
        out["embedding_trans/%s/embedding" % name] = (rows, dim)
    for (name, rows, dim, _f, _s) in spec["embedding_list_bias"]:
        out["%s/embedding" % name] = (rows, dim)
    return out


def mmoe_input_width(spec) -> int:
    w = spec["feature_dimension"] + sum(d for (_n, _r, d, _f, _s) in spec["embedding_list"])
    per_seq = spec["d_model"] * (2 if (spec.get("is_trans_out_concat_item") and not spec.get("is_trans_out_by_mlp")) else 1)      # :212-219
    return w + len(spec["attention_embed_pairs"]) * per_seq


def trans_prefix(i: int) -> str:
    # mmoe_transformer_unbias.py:193 ('trans_'+stag), TransformerModel.py:52 (name), :89/:136 (name again)
    return ("embedding_trans/trans_sequence_%d/encode_decode_sequence_%d/encode_decode_sequence_%d/" % (i, i, i))


def param_shapes(spec) -> Dict[str, tuple]:
    """SURVEY.md Appendix B variable inventory (names without the 'DnnModel/' prefix)."""
    d, dff = spec["d_model"], spec["d_ff"]
    shp = dict(table_shapes(spec))
    for i in range(len(spec["attention_embed_pairs"])):
        p = trans_prefix(i)
        if spec.get("position_encoding_method", "position_learn") == "position_learn":
            shp[p + "positional_encoding_k_position_learn/embedding_position_learn"] = (spec["maxlen_k"], d)
        # encoder block j (self-attention + ff) and decoder block j (vanilla_attention + ff) open the same scope 'num_blocks_j' under
        # AUTO_REUSE (TransformerModel.py:104-121, 154-169): one feed-forward per j (tie_ffn); dmt.conf has one block of each
        ne, nd = int(spec.get("num_blocks_encode", 1)), int(spec.get("num_blocks_decode", 1))
        for j in range(max(ne, nd)):
            blk = p + "num_blocks_%d/" % j
            for att in (("self-attention",) if j < ne else ()) + (("vanilla_attention",) if j < nd else ()):
                for dn in ("dense", "dense_1", "dense_2"):
                    shp[blk + "%s/%s/kernel" % (att, dn)] = (d, d)
                    shp[blk + "%s/%s/bias" % (att, dn)] = (d,)
                shp[blk + "%s/ln/beta" % att] = (d,)
                shp[blk + "%s/ln/gamma" % att] = (d,)
            if j < ne or spec.get("tie_ffn", True):
                ffj = blk + "positionwise_feedforward/"
                shp[ffj + "dense/kernel"] = (d, dff)
                shp[ffj + "dense/bias"] = (dff,)
                shp[ffj + "dense_1/kernel"] = (dff, d)
                shp[ffj + "dense_1/bias"] = (d,)
                shp[ffj + "ln/beta"] = (d,)
                shp[ffj + "ln/gamma"] = (d,)
        if spec.get("is_trans_input_by_mlp"):
            # tf.layers.dense(seq_emb / tar_sku_emb, d_model, name='dense_trans_seq_' / 'dense_trans_sku_' + stag) inside 'trans_' + stag: :196-198
            for nm in ("seq", "sku"):
                tp = "embedding_trans/trans_sequence_%d/dense_trans_%s_sequence_%d/" % (i, nm, i)
                shp[tp + "kernel"] = (d, d)
                shp[tp + "bias"] = (d,)
        if spec.get("is_trans_out_concat_item") and spec.get("is_trans_out_by_mlp"):
            # tf.layers.dense(final_state, d_model, name='dense_trans_concat_' + stag) inside variable_scope('trans_' + stag): :193, :216-217
            tp = "embedding_trans/trans_sequence_%d/dense_trans_concat_sequence_%d/" % (i, i)
            shp[tp + "kernel"] = (2 * d, d)
            shp[tp + "bias"] = (d,)
        if not spec.get("tie_ffn", True):  # untied variant keeps a second copy for the decoder
            for j in range(nd):
                ffd = p + "num_blocks_%d/positionwise_feedforward_dec/" % j
                for k, shape in (("dense/kernel", (d, dff)), ("dense/bias", (dff,)), ("dense_1/kernel", (dff, d)), ("dense_1/bias", (d,)),
                                 ("ln/beta", (d,)), ("ln/gamma", (d,))):
                    shp[ffd + k] = shape
    k_in = mmoe_input_width(spec)
    for e in range(spec["num_experts"]):
        prev = k_in
        for li, size in enumerate(spec["hidden_units_bottom"]):
            shp["mmoe_layers/expert-%d/expert-layer-%d/weights" % (e, li)] = (prev, size)
            shp["mmoe_layers/expert-%d/expert-layer-%d/biases" % (e, li)] = (size,)
            prev = size
    for t in range(spec["num_tasks"]):
        shp["mmoe_layers/gates-%d/gates-layer-0/weights" % t] = (k_in, spec["num_experts"])
        shp["mmoe_layers/gates-%d/gates-layer-0/biases" % t] = (spec["num_experts"],)
    for name in ("click", "order")[: spec["num_tasks"]]:
        prev = spec["hidden_units_bottom"][-1]
        for li, size in enumerate(spec["hidden_units_task"]):
            shp["%s/%s-fc-%d/weights" % (name, name, li)] = (prev, size)
            shp["%s/%s-fc-%d/biases" % (name, name, li)] = (size,)
            prev = size
        shp["%s/%s-output/weights" % (name, name)] = (prev, 1)
        shp["%s/%s-output/biases" % (name, name)] = (1,)
    prev = sum(d_ for (_n, _r, d_, _f, _s) in spec["embedding_list_bias"])
    for li, size in enumerate(list(spec["hidden_units_bias"]) + [spec["output_units"]]):
        shp["layer_bias%d/kernel" % li] = (prev, size)
        shp["layer_bias%d/bias" % li] = (size,)
        prev = size
    return shp


def init_params(spec, seed=0, dtype=np.float64) -> Dict[str, np.ndarray]:
    """Initialisers of the reference, by DISTRIBUTION (numpy RNG, not TF's stream):
    xavier/glorot-uniform (base.py:86, tf.layers.dense default), truncated normal sigma=0.1 cut at 2 sigma
    (base.py:32), constant 0.1 biases (base.py:36; bias_init=0.1 call sites), zeros / ones (LN, tf.layers bias)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in sorted(param_shapes(spec).items()):
        if name.endswith("/embedding") or name.endswith("embedding_position_learn") or name.endswith("/kernel"):
            fan_in, fan_out = shape[0], shape[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            v = rng.uniform(-lim, lim, size=shape)
        elif name.endswith("/weights"):
            v = rng.normal(0.0, 0.1, size=shape)
            bad = np.abs(v) > 0.2
            while bad.any():
                v[bad] = rng.normal(0.0, 0.1, size=int(bad.sum()))
                bad = np.abs(v) > 0.2
        elif name.endswith("/biases"):
            v = np.full(shape, 0.1)
        elif name.endswith("/gamma"):
            v = np.ones(shape)
        else:  # tf.layers bias, LN beta
            v = np.zeros(shape)
        out[name] = v.astype(dtype)
    return out


# --------------------------------------------------------------------------------------------- dropout
def _mix32(h):
    h = np.asarray(h, dtype=np.uint64) & 0xFFFFFFFF
    h ^= h >> 16; h = (h * 0x85EBCA6B) & 0xFFFFFFFF; h ^= h >> 13; h = (h * 0xC2B2AE35) & 0xFFFFFFFF; h ^= h >> 16
    return h


def site_seed(step_seed: int, stream: int) -> int:
    """32-bit seed of one dropout site (stream) at one step; streams: 10*i+0 encoder input, 10*i+1 decoder input,
    10*i+2 (+1000*j) self-attention weights, 10*i+3 (+1000*j) cross-attention weights of block j of sequence i; 100+l bias-tower layer l."""
    return int(_mix32((step_seed * 0x9E3779B1 + stream * 0x85EBCA6B + 1) & 0xFFFFFFFF))


def dropout_mask(seed32: int, n: int, keep_prob: float) -> np.ndarray:
    """keep(i) = (mix32(i ^ seed) >> 8) < keep_prob * 2**24 -- the counter-based mask of libdmt_hip (dmt_common.h).
    The reference uses tf.layers.dropout's stateful RNG (TransformerModel.py:101,151; TransformerModel_util.py:51;
    mmoe_transformer_unbias.py:274-278); only the DISTRIBUTION (Bernoulli(keep), scaled by 1/keep) is reference behaviour."""
    idx = np.arange(n, dtype=np.uint64)
    thr = np.uint64(int(np.float32(keep_prob) * np.float32(16777216.0)))
    return (_mix32(idx ^ np.uint64(seed32)) >> 8) < thr


def dropout(x, rate, step_seed, stream):
    if step_seed is None or not rate:
        return x
    keep = 1.0 - rate
    m = dropout_mask(site_seed(step_seed, stream), x.size, keep).reshape(x.shape)
    return np.where(m, x / keep, 0.0)


# --------------------------------------------------------------------------------------------- ops
def ln(x, gamma, beta, epsilon=1e-8):
    """TransformerModel_util.py:58-78: biased variance over the last dim, eps INSIDE the sqrt."""
    mean = x.mean(axis=-1, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=-1, keepdims=True)
    normalized = (x - mean) / ((var + epsilon) ** 0.5)
    return gamma * normalized + beta


def softmax(x):
    m = x.max(axis=-1, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=-1, keepdims=True)


def sequence_mask(lengths, maxlen):
    return (np.arange(maxlen)[None, :] < np.asarray(lengths)[:, None])


def scaled_dot_product_attention(Q, K, V, query_masks, key_masks, dropout_rate=0.0, step_seed=None, stream=0):
    """TransformerModel_util.py:11-56 with causality False (dropout only when step_seed is given).
    Q [hN,Tq,dk]  K,V [hN,Tk,dk]  masks [N,T*] bool."""
    d_k = Q.shape[-1]
    outputs = np.matmul(Q, np.transpose(K, (0, 2, 1)))
    outputs = outputs / (d_k ** 0.5)
    h = Q.shape[0] // key_masks.shape[0]
    km = np.tile(key_masks, (h, 1))[:, None, :]                       # mask(type='key'), :83-90
    outputs = np.where(np.broadcast_to(km, outputs.shape), outputs, PADDING_NUM)
    outputs = softmax(outputs)
    qm = np.tile(query_masks, (h, 1))[:, :, None]                     # mask(type='query'), :91-97 (AFTER softmax)
    outputs = np.where(np.broadcast_to(qm, outputs.shape), outputs, PADDING_NUM)
    if step_seed is not None and dropout_rate:
        N = key_masks.shape[0]
        Tq, Tk = outputs.shape[1], outputs.shape[2]
        keep = 1.0 - dropout_rate
        m = dropout_mask(site_seed(step_seed, stream), N * h * Tq * Tk, keep).reshape(N, h, Tq, Tk)   # idx = ((b*H+h)*Tq+q)*Tk+k
        m = np.transpose(m, (1, 0, 2, 3)).reshape(h * N, Tq, Tk)                                      # head-major packing
        outputs = np.where(m, outputs / keep, 0.0)
    return np.matmul(outputs, V)


def multihead_attention(queries, keys, values, queries_length, keys_length, num_heads, P, scope, dropout_rate=0.0,
                        step_seed=None, stream=0):
    """TransformerModel_util.py:160-209.  NOTE: no output projection."""
    query_masks = sequence_mask(queries_length, queries.shape[1])
    key_masks = sequence_mask(keys_length, keys.shape[1])
    Q = queries @ P[scope + "dense/kernel"] + P[scope + "dense/bias"]
    K = keys @ P[scope + "dense_1/kernel"] + P[scope + "dense_1/bias"]
    V = values @ P[scope + "dense_2/kernel"] + P[scope + "dense_2/bias"]
    Q_ = np.concatenate(np.split(Q, num_heads, axis=2), axis=0)
    K_ = np.concatenate(np.split(K, num_heads, axis=2), axis=0)
    V_ = np.concatenate(np.split(V, num_heads, axis=2), axis=0)
    out = scaled_dot_product_attention(Q_, K_, V_, query_masks, key_masks, dropout_rate, step_seed, stream)
    out = np.concatenate(np.split(out, num_heads, axis=0), axis=2)
    out = out + queries
    return ln(out, P[scope + "ln/gamma"], P[scope + "ln/beta"])


def ff(inputs, P, scope):
    """TransformerModel_util.py:212-235."""
    h = np.maximum(inputs @ P[scope + "dense/kernel"] + P[scope + "dense/bias"], 0.0)
    o = h @ P[scope + "dense_1/kernel"] + P[scope + "dense_1/bias"]
    o = o + inputs
    return ln(o, P[scope + "ln/gamma"], P[scope + "ln/beta"])


def position_table(P, prefix, spec):
    """TransformerModel.py:60-69 position_encode: the learned [maxlen_k, d] variable (position_learn, dmt.conf:50) or the sinusoid of
    TransformerModel_util.py:238-279 (position_sin_cos): PE[pos, i] = pos / 10000^((i - i % 2) / E), sin on even columns, cos on odd ones,
    a float32 constant there (tf.convert_to_tensor(position_enc, tf.float32))."""
    if spec.get("position_encoding_method", "position_learn") == "position_learn":
        return P[prefix + "positional_encoding_k_position_learn/embedding_position_learn"]
    E, maxlen = spec["d_model"], spec["maxlen_k"]
    enc = np.array([[pos / np.power(10000, (i - i % 2) / E) for i in range(E)] for pos in range(maxlen)])
    enc[:, 0::2] = np.sin(enc[:, 0::2])
    enc[:, 1::2] = np.cos(enc[:, 1::2])
    return enc.astype(np.float32).astype(np.float64)


def encode(seq_emb, seqlens, P, prefix, spec, step_seed=None, seq_index=0):
    """TransformerModel.py:84-123 with position_learn (dmt.conf:50); dropout (rate spec['dropout_rate']) when step_seed given."""
    T = seq_emb.shape[1]
    rate = spec.get("dropout_rate", 0.0) if step_seed is not None else 0.0
    enc = seq_emb * (spec["d_model"] ** 0.5)
    pos = position_table(P, prefix, spec)
    enc = enc + pos[np.arange(T)][None, :, :]                          # positional_encoding_learn, util:281-316 / positional_encoding :238-279
    enc = dropout(enc, rate, step_seed, 10 * seq_index + 0)            # TransformerModel.py:101
    for i in range(spec["num_blocks_encode"]):
        blk = prefix + "num_blocks_%d/" % i
        enc = multihead_attention(enc, enc, enc, seqlens, seqlens, spec["num_heads"], P, blk + "self-attention/", rate, step_seed,
                                  10 * seq_index + 2 + 1000 * i)      # one independent tf.layers.dropout mask per block
        enc = ff(enc, P, blk + "positionwise_feedforward/")
    return enc


def decode(query_emb, query_length, key_emb, key_length, P, prefix, spec, step_seed=None, seq_index=0):
    """TransformerModel.py:125-171 (dmt.conf: is_decoder_add_pos_emb = false)."""
    rate = spec.get("dropout_rate", 0.0) if step_seed is not None else 0.0
    dec = query_emb * (spec["d_model"] ** 0.5)
    if spec.get("is_decoder_add_pos_emb"):                             # TransformerModel.py:148-149: sinusoid rows 0 .. Tq-1 (maxlen_q >= Tq)
        E, Tq = spec["d_model"], query_emb.shape[1]
        pe = np.array([[pos / np.power(10000, (i - i % 2) / E) for i in range(E)] for pos in range(Tq)])
        pe[:, 0::2] = np.sin(pe[:, 0::2])
        pe[:, 1::2] = np.cos(pe[:, 1::2])
        dec = dec + pe.astype(np.float32).astype(np.float64)[None, :, :]
    dec = dropout(dec, rate, step_seed, 10 * seq_index + 1)            # TransformerModel.py:151
    for i in range(spec["num_blocks_decode"]):
        blk = prefix + "num_blocks_%d/" % i
        dec = multihead_attention(dec, key_emb, key_emb, query_length, key_length, spec["num_heads"], P,
                                  blk + "vanilla_attention/", rate, step_seed, 10 * seq_index + 3 + 1000 * i)
        ffs = "positionwise_feedforward/" if spec.get("tie_ffn", True) else "positionwise_feedforward_dec/"
        dec = ff(dec, P, blk + ffs)                                    # same scope => tied weights
    return dec


def embedding_zero_pad(E):
    return np.concatenate([np.zeros((1, E.shape[1]), dtype=E.dtype), E], axis=0)   # base.py:87-89


def generate_data(inputs, P, spec):
    """mmoe_transformer_unbias.py:130-186."""
    seq_data = []
    emb_by_feat = {f: (n, r, d) for (n, r, d, f, _s) in spec["embedding_list"]}
    for index, pairs in enumerate(spec["attention_embed_pairs"]):
        seq_features, tar_features = [], []
        mask = lens = None
        for (user_feature, item_feature) in pairs:
            sp = inputs[user_feature]
            mask = sparse_to_dense(SparseTensorValue(sp.indices, np.ones(len(sp.values), np.int32), sp.dense_shape))
            lens = mask.sum(axis=1)
            for (name, _rows, _dim, feat, _side) in spec["embedding_list"]:
                if feat == user_feature:
                    E0 = embedding_zero_pad(P["embedding_trans/%s/embedding" % name])
                    seq_features.append(E0[sparse_to_dense(inputs[feat])])
                elif feat == item_feature:
                    E0 = embedding_zero_pad(P["embedding_trans/%s/embedding" % name])
                    tar_features.append(E0[np.asarray(inputs[feat].values)])
        seq_ts_emb = None
        if spec["attention_embed_seq_ts"]:
            ts_feature = spec["attention_embed_seq_ts"][index]
            name = emb_by_feat[ts_feature][0]
            E0 = embedding_zero_pad(P["embedding_trans/%s/embedding" % name])
            ts_input = sparse_to_dense(inputs[ts_feature]).astype(np.float32)
            with np.errstate(divide="ignore"):
                lg = np.log(ts_input) / np.float32(np.log(2.0))
            # tf.cast(-inf, int32) == INT32_MIN; +1; clip to [0, 23]   (:172-173)
            ti = np.where(np.isfinite(lg), np.trunc(np.where(np.isfinite(lg), lg, 0)), -2147483648.0).astype(np.int64) + 1
            ti = np.clip(ti, 0, 23)
            seq_ts_emb = E0[ti]
        seq_data.append([mask, lens, np.concatenate(seq_features, -1), np.concatenate(tar_features, -1), seq_ts_emb])
    return seq_data


def trans_core(seq_data, P, spec, step_seed=None):
    """mmoe_transformer_unbias.py:189-223; is_trans_out_concat_item (:212-215, dmt.conf: false): final_state = [user_stat, tar_sku_emb] with the
    RAW target embedding (the decoder scales its own copy); is_trans_out_by_mlp (:216-217): a dense layer folds the pair back to d_model."""
    states = []
    for i, (mask, lens, seq_emb, tar, _ts) in enumerate(seq_data):
        prefix = trans_prefix(i)
        if spec.get("is_trans_input_by_mlp"):          # :196-198, before the Transformer's own scale / position / dropout prep
            tp = "embedding_trans/trans_sequence_%d/" % i
            seq_emb = seq_emb @ P[tp + "dense_trans_seq_sequence_%d/kernel" % i] + P[tp + "dense_trans_seq_sequence_%d/bias" % i]
            tar = tar @ P[tp + "dense_trans_sku_sequence_%d/kernel" % i] + P[tp + "dense_trans_sku_sequence_%d/bias" % i]
        seq_q = tar[:, None, :]
        q_lens = np.ones(seq_q.shape[0], dtype=np.int64)
        memory = encode(seq_emb, lens, P, prefix, spec, step_seed, i)
        dec = decode(seq_q, q_lens, memory, lens, P, prefix, spec, step_seed, i)
        final = dec[:, 0, :]
        if spec.get("is_trans_out_concat_item"):
            final = np.concatenate([final, tar], -1)
            if spec.get("is_trans_out_by_mlp"):
                tp = "embedding_trans/trans_sequence_%d/dense_trans_concat_sequence_%d/" % (i, i)
                final = final @ P[tp + "kernel"] + P[tp + "bias"]
        states.append(final)
    return np.concatenate(states, -1)


def embedding_lookup_sparse_mean(E, sp_ids, sp_w, B):
    """tf.nn.embedding_lookup_sparse(combiner='mean') (base.py:116): sum_t w*E[id] / sum_t w per example."""
    out = np.zeros((B, E.shape[1]), dtype=E.dtype)
    wsum = np.zeros((B,), dtype=E.dtype)
    ids = np.asarray(sp_ids.values)
    w = np.ones(len(ids), dtype=E.dtype) if sp_w is None else np.asarray(sp_w.values).astype(E.dtype)
    for n in range(len(ids)):
        b = int(sp_ids.indices[n, 0])
        out[b] += w[n] * E[int(ids[n])]
        wsum[b] += w[n]
    nz = wsum != 0
    out[nz] = out[nz] / wsum[nz][:, None]
    return out


def embedding_combiner(inputs, P, spec, emb_list=None, prefix="embedding_trans/", with_dense=True):
    """base.py:93-134 (sim_embed empty) and mmoe_transformer_unbias.py:235-257 (bias variant)."""
    emb_list = spec["embedding_list"] if emb_list is None else emb_list
    B = inputs["features"].shape[0]
    feats = [np.asarray(inputs["features"])] if with_dense else []
    for (name, _rows, _dim, feat, _side) in emb_list:
        E = P[prefix + "%s/embedding" % name]
        feats.append(embedding_lookup_sparse_mean(E, inputs[feat], inputs.get(feat + "Wts"), B))
    dt = P[prefix + "%s/embedding" % emb_list[0][0]].dtype
    return np.concatenate([f.astype(dt) for f in feats], axis=1)


def embedding_trans(inputs, P, spec, step_seed=None):
    """mmoe_transformer_unbias.py:226-233."""
    seq_data = generate_data(inputs, P, spec)
    interest = trans_core(seq_data, P, spec, step_seed)
    features = embedding_combiner(inputs, P, spec)
    return np.concatenate([features, interest], -1)


def dense_layer(x, W, b, activation):
    """base.py:39-68 with is_bn=false, is_dropout=false."""
    y = x @ W + b
    if activation == "relu":
        return np.maximum(y, 0.0)
    if activation == "softmax":
        return softmax(y)
    return y


def expert_gate(features, P, spec):
    """mmoe_transformer_unbias.py:63-105."""
    experts = []
    for e in range(spec["num_experts"]):
        y = features
        for li in range(len(spec["hidden_units_bottom"])):
            s = "mmoe_layers/expert-%d/expert-layer-%d/" % (e, li)
            y = dense_layer(y, P[s + "weights"], P[s + "biases"], "relu")
        experts.append(y)
    gates = []
    for t in range(spec["num_tasks"]):
        s = "mmoe_layers/gates-%d/gates-layer-0/" % t
        gates.append(dense_layer(features, P[s + "weights"], P[s + "biases"], "softmax"))
    ex = np.stack(experts, axis=-1)                                    # [B, units, E]
    outs = [(ex * g[:, None, :]).sum(axis=2) for g in gates]
    return outs, gates


def build_tower(x, P, spec, name):
    """mmoe_transformer_unbias.py:107-126."""
    y = x
    for li in range(len(spec["hidden_units_task"])):
        s = "%s/%s-fc-%d/" % (name, name, li)
        y = dense_layer(y, P[s + "weights"], P[s + "biases"], "relu")
    s = "%s/%s-output/" % (name, name)
    return dense_layer(y, P[s + "weights"], P[s + "biases"], "identity")


def embedding_mlp_bias(inputs, P, spec, step_seed=None):
    """mmoe_transformer_unbias.py:259-289 (dropout after each hidden layer when step_seed is given)."""
    y = embedding_combiner(inputs, P, spec, emb_list=spec["embedding_list_bias"], prefix="", with_dense=False)
    n = len(spec["hidden_units_bias"])
    for li in range(n):
        y = np.maximum(y @ P["layer_bias%d/kernel" % li] + P["layer_bias%d/bias" % li], 0.0)
        y = dropout(y, spec.get("dropout_rate_bias", [0.0] * n)[li] if step_seed is not None else 0.0, step_seed, 100 + li)
    return y @ P["layer_bias%d/kernel" % n] + P["layer_bias%d/bias" % n]


def inference(inputs, P, spec, is_predict=False, step_seed=None):
    """mmoe_transformer_unbias.py:293-316 -> ((click_logit, order_logit), y_bias).  step_seed != None == is_train with dropout."""
    features = embedding_trans(inputs, P, spec, step_seed)
    mmoe_layers, _g = expert_gate(features, P, spec)
    logits = tuple(build_tower(m, P, spec, nm) for m, nm in zip(mmoe_layers, ("click", "order")))
    if is_predict:
        return logits
    return logits, embedding_mlp_bias(inputs, P, spec, step_seed)


# --------------------------------------------------------------------------------------------- loss
def serving_normalise(raw, mean, std):
    """Online normalisation of the raw dense features in the exported graph, float32 as the TF constants are
    (saved_model/export_model.py:86-96; constants from saved_model/preprocess.py:17-40, computed there in float64)."""
    m, s = np.asarray(mean, dtype=np.float64), np.asarray(std, dtype=np.float64)
    eps64 = np.full_like(s, 1e-7)
    div1 = (m * s) / (np.square(s + eps64) * 3)                 # preprocess.py:30-31
    div2 = (m * s) / (s + eps64)                                # :32
    const_vec = ((div1 + div2) - m).astype(np.float32)          # :33-34, fed to tf.constant(float32) at export_model.py:87
    std32 = np.asarray(std, dtype=np.float32)
    eps32 = np.full(std32.shape, 0.0000001, dtype=np.float32)
    x = np.clip(np.asarray(raw, dtype=np.float32), 0.0, np.finfo(np.float32).max)            # :91
    y = (x * std32) / (np.square(std32 + eps32) * np.float32(3.0)) - const_vec                 # :92-93
    return np.clip(y, np.float32(-0.99), np.float32(0.99))                                     # :95


def serving_scores(click_logit, order_logit, export_weight):
    """Scores = (w0 * sigmoid(click) + w1 * sigmoid(order)) / sum(w)  (saved_model/export_model.py:106-114)."""
    pc, po = sigmoid(np.reshape(click_logit, -1)), sigmoid(np.reshape(order_logit, -1))
    return (export_weight[0] * pc + export_weight[1] * po) / float(sum(export_weight))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def cal_cross_entropy(p, labels, eps=1e-7):
    """model/inference_mlp.py:162-168 -> keras.backend.sparse_categorical_crossentropy(from_logits=False):
    clip([1-p, p], eps, 1-eps) -> log -> sparse softmax CE with logits = log(clipped)."""
    p = np.asarray(p).reshape(-1, 1)
    q = np.concatenate([1.0 - p, p], axis=-1)
    q = np.clip(q, eps, 1.0 - eps)
    logits = np.log(q)
    lse = np.log(np.exp(logits).sum(axis=-1))
    y = np.asarray(labels).reshape(-1).astype(np.int64)
    return lse - logits[np.arange(len(y)), y]


def loss_multi_task_unbias(logits, mask, spec, loss_unbias_method="two_head_add", loss_ctr_rel_method="ctr_rel"):
    """model/inference_mlp.py:173-223."""
    (click_logit, order_logit), y_bias = logits
    if loss_unbias_method == "two_head_multiply":
        p_ctr = sigmoid(click_logit) * sigmoid(y_bias)
        p_cvr = sigmoid(order_logit) * sigmoid(y_bias)
    else:
        p_ctr = sigmoid(click_logit + y_bias)
        p_cvr = sigmoid(order_logit + y_bias)
    p_rel_ctr, p_rel_cvr = sigmoid(click_logit), sigmoid(order_logit)
    mask = np.asarray(mask)
    labels_clk = mask[:, 1:5].sum(axis=-1)
    labels_ord = mask[:, 3] + mask[:, 4]
    x_clk = cal_cross_entropy(p_ctr, labels_clk)
    x_ord = cal_cross_entropy(p_cvr, labels_ord)
    if loss_ctr_rel_method == "ctr_rel":
        x_clk = x_clk + cal_cross_entropy(p_rel_ctr, labels_clk)
        x_ord = x_ord + cal_cross_entropy(p_rel_cvr, labels_ord)
    w_ctr = np.asarray(spec["weight_ctr"], dtype=mask.dtype)
    w_ecvr = np.asarray(spec["weight_ecvr"], dtype=mask.dtype)
    loss_clk = ((mask * w_ctr).T * x_clk).mean(axis=1).sum()
    loss_ord = ((mask * w_ecvr).T * x_ord).mean(axis=1).sum()
    return spec["loss_weight"][0] * loss_clk + spec["loss_weight"][1] * loss_ord


def loss_multi_task(logits, mask, spec):
    """model/inference_mlp.py:228-258 (sigmoid_cross_entropy_with_logits, no clipping)."""
    click_logit, order_logit = logits
    mask = np.asarray(mask)
    labels_clk = mask[:, 1:5].sum(axis=-1)
    labels_ord = mask[:, 3] + mask[:, 4]

    def sce(x, z):
        x = x.reshape(-1)
        return np.maximum(x, 0) - x * z + np.log1p(np.exp(-np.abs(x)))

    loss_clk = ((mask * np.asarray(spec["weight_ctr"])).T * sce(click_logit, labels_clk)).mean(axis=1).sum()
    loss_ord = ((mask * np.asarray(spec["weight_ecvr"])).T * sce(order_logit, labels_ord)).mean(axis=1).sum()
    return spec["loss_weight"][0] * loss_clk + spec["loss_weight"][1] * loss_ord


def cal_ctr_cvr_unbias(y_rel, y_bias, loss_unbias_method="two_head_add"):
    """run_dnn.py:90-101."""
    click_logit, order_logit = y_rel
    if loss_unbias_method == "two_head_multiply":
        return sigmoid(click_logit) * sigmoid(y_bias), sigmoid(order_logit) * sigmoid(y_bias)
    return sigmoid(click_logit + y_bias), sigmoid(order_logit + y_bias)


# --------------------------------------------------------------------------------------------- optimiser
class TFAdam:
    """tf.train.AdamOptimizer(lr) (model/inference_mlp.py:264-273) as applied DENSELY to every variable by
    run_dnn.py:203-207.  Arithmetic of tensorflow==1.12 core/kernels/training_ops.cc ApplyAdam (un-vendored):
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)        (beta powers kept as running products)
        m += (g - m) * (1 - beta1);  v += (g*g - v) * (1 - beta2);  var -= lr_t * m / (sqrt(v) + eps)
    """

    def __init__(self, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float64):
        self.dt = dtype
        self.lr, self.b1, self.b2, self.eps = dtype(lr), dtype(beta1), dtype(beta2), dtype(eps)
        self.b1p, self.b2p = dtype(beta1), dtype(beta2)   # beta1_power / beta2_power before step 1
        self.m, self.v = {}, {}

    def lr_t(self):
        return self.lr * np.sqrt(self.dt(1) - self.b2p) / (self.dt(1) - self.b1p)

    def apply(self, P: Dict[str, np.ndarray], G: Dict[str, np.ndarray]):
        a = self.lr_t()
        for k, g in G.items():
            if k not in self.m:
                self.m[k] = np.zeros_like(P[k])
                self.v[k] = np.zeros_like(P[k])
            m, v = self.m[k], self.v[k]
            m += (g - m) * (self.dt(1) - self.b1)
            v += (g * g - v) * (self.dt(1) - self.b2)
            P[k] -= (m * a) / (np.sqrt(v) + self.eps)
        self.b1p = self.b1p * self.b1
        self.b2p = self.b2p * self.b2


def l2_norm(inputs, P: Dict[str, np.ndarray], spec: dict, l2_emb_lambda: float, batch_size: float):
    """l2_norm (model/net/mmoe_transformer_unbias.py:42-60), added to the tower loss when wnd_wd > 1e-5 (run_dnn.py:174-175):
    sum over the embedding_list entries of tf.nn.l2_loss(tf.gather(E, tf.unique(ids of the entry's feature))) = sum(row^2) / 2 over
    the DISTINCT ids, times l2_emb_lambda / batch_size (tf.losses.get_regularization_losses() is empty: no layer registers one).
    Returns (value, {table tf_name: d value / d E}): lambda / batch_size * E[id] per entry on each of its distinct ids -- a table that
    two entries read gets both terms."""
    val, grads = 0.0, {}
    for (name, _rows, _dim, feat, _side) in spec["embedding_list"]:
        tf_name = "embedding_trans/%s/embedding" % name
        E = P[tf_name].astype(np.float64)
        ids = np.unique(np.asarray(inputs[feat].values).astype(np.int64))
        val += 0.5 * float((E[ids] ** 2).sum())
        g = grads.setdefault(tf_name, np.zeros_like(E))
        g[ids] += E[ids]
    k = float(l2_emb_lambda) / float(batch_size)
    return val * k, {n: g * k for n, g in grads.items()}


class TFOptimizer:
    """The other optimizers get_optimizer returns (model/inference_mlp.py:264-280), built with the learning rate only, so every other
    hyper-parameter is TF 1.12's constructor default; applied DENSELY to every variable (run_dnn.py:203-207 on the densified
    gradients of run_dnn.py:45-80).  Arithmetic of tensorflow==1.12 core/kernels/training_ops.cc (un-vendored):
      sgd       ApplyGradientDescent   var -= lr * g
      adagrad   ApplyAdagrad           accum (init 0.1) += g*g;  var -= lr * g / sqrt(accum)
      adadelta  ApplyAdadelta          rho 0.95, epsilon 1e-8:  accum = rho accum + (1-rho) g*g;
                                       update = sqrt(accum_update + eps) / sqrt(accum + eps) * g;  var -= lr * update;
                                       accum_update = rho accum_update + (1-rho) update^2
      rmsprop   ApplyRMSProp           decay 0.9, momentum 0.0, epsilon 1e-10, rms slot init 1.0:
                                       ms += (g*g - ms) (1-decay);  mom = momentum mom + lr g / sqrt(ms + eps);  var -= mom
      ftrl      ApplyFtrl              learning_rate_power -0.5, initial_accumulator_value 0.1, l1 = l2 = 0:
                                       new = accum + g*g;  linear += g - (sqrt(new) - sqrt(accum)) / lr * var;
                                       var = (sign(linear) l1 - linear) / (sqrt(new) / lr + 2 l2) if |linear| > l1 else 0;  accum = new
                                       (so an element whose gradient has been zero at every step so far -- an embedding row no batch
                                        has read -- becomes ZERO at the first step: that is what the reference's dense update does)
    """
    KINDS = ("sgd", "adagrad", "adadelta", "rmsprop", "ftrl")

    def __init__(self, kind, lr, dtype=np.float64):
        assert kind in self.KINDS, kind
        self.kind, self.dt, self.lr = kind, dtype, dtype(lr)
        self.s0, self.s1 = {}, {}

    def apply(self, P: Dict[str, np.ndarray], G: Dict[str, np.ndarray]):
        dt, lr, kind = self.dt, self.lr, self.kind
        for k, g in G.items():
            g = g.astype(dt)
            if k not in self.s0:
                init0 = {"adagrad": 0.1, "ftrl": 0.1, "rmsprop": 1.0}.get(kind, 0.0)
                self.s0[k] = np.full_like(P[k], init0, dtype=dt)
                self.s1[k] = np.zeros_like(P[k], dtype=dt)
            a, b = self.s0[k], self.s1[k]
            if kind == "sgd":
                P[k] -= (lr * g).astype(P[k].dtype)
            elif kind == "adagrad":
                a += g * g
                P[k] -= (lr * g / np.sqrt(a)).astype(P[k].dtype)
            elif kind == "adadelta":
                rho, eps = dt(0.95), dt(1e-8)
                a[...] = rho * a + (dt(1) - rho) * g * g
                upd = np.sqrt(b + eps) / np.sqrt(a + eps) * g
                P[k] -= (lr * upd).astype(P[k].dtype)
                b[...] = rho * b + (dt(1) - rho) * upd * upd
            elif kind == "rmsprop":
                decay, eps = dt(0.9), dt(1e-10)
                a += (g * g - a) * (dt(1) - decay)
                b[...] = lr * g / np.sqrt(a + eps)                 # momentum 0.0
                P[k] -= b.astype(P[k].dtype)
            else:
                l1, l2 = dt(0.0), dt(0.0)
                new = a + g * g
                b += g - (np.sqrt(new) - np.sqrt(a)) / lr * P[k].astype(dt)
                quad = np.sqrt(new) / lr + dt(2) * l2
                P[k][...] = np.where(np.abs(b) > l1, (np.sign(b) * l1 - b) / quad, dt(0)).astype(P[k].dtype)
                a[...] = new


# --------------------------------------------------------------------------------------------- metrics
def tf_metrics_auc(labels, predictions, num_thresholds=200):
    """tf.metrics.auc(curve='ROC', summation_method='trapezoidal') as used at run_dnn.py:228-241
    (tensorflow==1.12 python/ops/metrics_impl.py, un-vendored)."""
    kepsilon = 1e-7
    thresholds = [(i + 1) * 1.0 / (num_thresholds - 1) for i in range(num_thresholds - 2)]
    thresholds = np.array([0.0 - kepsilon] + thresholds + [1.0 + kepsilon])
    labels = np.asarray(labels).reshape(-1) > 0.5
    pred = np.asarray(predictions).reshape(-1).astype(np.float32)
    pos = pred[None, :] > thresholds[:, None].astype(np.float32)
    tp = (pos & labels[None, :]).sum(axis=1).astype(np.float64)
    fp = (pos & ~labels[None, :]).sum(axis=1).astype(np.float64)
    fn = (~pos & labels[None, :]).sum(axis=1).astype(np.float64)
    tn = (~pos & ~labels[None, :]).sum(axis=1).astype(np.float64)
    epsilon = 1.0e-6
    rec = (tp + epsilon) / (tp + fn + epsilon)
    fp_rate = fp / (fp + tn + epsilon)
    x, y = fp_rate, rec
    return float(((x[:-1] - x[1:]) * (y[:-1] + y[1:]) / 2.0).sum())


def exact_auc(labels, predictions):
    """Rank (Mann-Whitney) AUC with average ranks for ties."""
    y = np.asarray(labels).reshape(-1) > 0.5
    s = np.asarray(predictions).reshape(-1).astype(np.float64)
    n_pos, n_neg = int(y.sum()), int((~y).sum())
    if n_pos == 0 or n_neg == 0:
        return float("nan")
    order = np.argsort(s, kind="mergesort")
    ranks = np.empty(len(s), dtype=np.float64)
    ss = s[order]
    i = 0
    while i < len(ss):
        j = i
        while j + 1 < len(ss) and ss[j + 1] == ss[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    return float((ranks[y].sum() - n_pos * (n_pos + 1) / 2.0) / (n_pos * n_neg))
