"""Functional Transformer ops with the reference's signatures (/root/reference/DMT_code/model/net/TransformerModel_util.py).

    scaled_dot_product_attention :11-56     ln :58-78     mask :80-108     multihead_attention :160-209
    ff :212-235     positional_encoding_learn :281-316
Variables are looked up by scoped name in the active Runtime's store (the reference creates them under the same
names with tf.get_variable / tf.layers.dense).  Dropout (training=True and dropout_rate > 0) uses the library's counter mask with
the engine's step seed (engine.dropout_step_seed, set by Inference.inference(is_train=True)); the dropout site is derived from the
variable scope (sequence i: self-attention weights stream 10 i + 2, vanilla attention 10 i + 3).
Tensors are device tensors; every op is a libdmt_hip.so kernel.
"""
from __future__ import annotations

import torch

from ... import ops
from .. import runtime as R

PADDING_NUM = float(-2 ** 32 + 1)


def _store():
    return R.get_default().store


def _leaf(name):
    st = _store()
    full = R.current_scope() + name
    if full not in st.leaf:
        raise KeyError("variable '%s' is not part of the DMT variable inventory (SURVEY.md Appendix B)" % full)
    return st.leaf[full], st.weight.get(full)


def _seq_index() -> int:
    """i of the enclosing 'trans_sequence_i' scope (0 outside one)."""
    import re
    m = re.search(r"trans_sequence_(\d+)", R.current_scope())
    return int(m.group(1)) if m else 0


def _block_index() -> int:
    """j of the enclosing `num_blocks_j` scope (TransformerModel.py:104,154): every block draws its own attention-dropout mask."""
    import re
    m = re.search(r"num_blocks_(\d+)", R.current_scope())
    return int(m.group(1)) if m else 0


def _attn_dropout(rate, training, offset):
    """(site seed, keep probability) of the attention-weight dropout, or (0, 1.0) when it is off."""
    eng = R.get_default().engine
    if not training or not rate:
        return 0, 1.0
    if eng.dropout_step_seed is None:
        raise RuntimeError("training=True with dropout_rate > 0 needs engine.dropout_step_seed (Inference.inference(is_train=True) sets it)")
    return ops.site_seed(eng.dropout_step_seed, 10 * _seq_index() + offset + 1000 * _block_index()), 1.0 - float(rate)


def ln(inputs, epsilon=1e-8, scope="ln"):
    with R.variable_scope(scope):
        gamma, _ = _leaf("gamma")
        beta, _ = _leaf("beta")
    return ops.layer_norm(inputs, gamma, beta, epsilon)


def mask(inputs, query_masks=None, key_masks=None, type=None):
    """Stand-alone masking of a score tensor is fused into the attention kernel; this helper exists for API parity
    and applies the same rule with torch indexing on the host-visible tensor (not on the hot path)."""
    if type in ("k", "key", "keys"):
        h = inputs.shape[0] // key_masks.shape[0]
        km = key_masks.bool().repeat(h, 1)[:, None, :].expand_as(inputs)
        return torch.where(km, inputs, torch.full_like(inputs, PADDING_NUM))
    if type in ("q", "query", "queries"):
        h = inputs.shape[0] // query_masks.shape[0]
        qm = query_masks.bool().repeat(h, 1)[:, :, None].expand_as(inputs)
        return torch.where(qm, inputs, torch.full_like(inputs, PADDING_NUM))
    if type in ("f", "future", "right"):
        Tq, Tk = inputs.shape[1], inputs.shape[2]
        tril = torch.ones((Tq, Tk), dtype=torch.bool, device=inputs.device).tril()
        return torch.where(tril[None], inputs, torch.full_like(inputs, PADDING_NUM))
    raise ValueError("mask type %r (the reference prints 'Check if you entered type correctly!')" % (type,))


def scaled_dot_product_attention(Q, K, V, query_masks, key_masks, causality=False, dropout_rate=0., training=True,
                                 scope="scaled_dot_product_attention"):
    """Q,K,V: [h*N, T, d_k] head-major packing of the reference; masks [N, T] bool / 0-1.  Returns [h*N, T_q, d_k]."""
    seed, keep = _attn_dropout(dropout_rate, training, 2)
    N = key_masks.shape[0]
    h = Q.shape[0] // N
    dk = Q.shape[-1]

    def unpack(x):  # [h*N, T, dk] -> [N, T, h*dk]
        return torch.cat(torch.split(x, N, dim=0), dim=2).contiguous()

    q, k, v = unpack(Q), unpack(K), unpack(V)
    q_lens = query_masks.to(torch.int32).sum(1).to(torch.int32).contiguous()
    k_lens = key_masks.to(torch.int32).sum(1).to(torch.int32).contiguous()
    kv = torch.cat([k, v], dim=-1)
    out = ops.AttnFn.apply(q, kv, None, q_lens, k_lens, h, h * dk, False, seed, keep, None, bool(causality))
    return torch.cat(torch.split(out, dk, dim=2), dim=0)


def multihead_attention(queries, keys, values, queries_length, keys_length, num_heads=8, dropout_rate=0, training=True,
                        causality=False, scope="multihead_attention"):
    """causality=True (future blinding) and values != keys are kept for signature parity (TransformerModel_util.py:160-209); DMT's own
    graph uses neither (TransformerModel.py:117, 165: causality=False, values = keys).  Both take the same kernels: causality through the
    unfused score / softmax / value launches (dmt_softmax_fwd's `causal`), separate values through a second projection launch."""
    d = queries.shape[-1]
    with R.variable_scope(scope):
        wl, w = _leaf("qkv_kernel")
        bl, _ = _leaf("qkv_bias")
        gamma, _ = _leaf("ln/gamma")
        beta, _ = _leaf("ln/beta")
    eng = R.get_default().engine
    ql = queries_length.to(torch.int32).contiguous() if queries_length is not None else None
    kl = keys_length.to(torch.int32).contiguous()
    causal = bool(causality)
    if queries is keys and values is keys:
        seed, keep = _attn_dropout(dropout_rate, training, 2)
        qkv = ops.linear(queries, wl, bl, w)
        s = ops.AttnFn.apply(qkv, None, queries, ql, kl, num_heads, d, True, seed, keep, None, causal)
    else:
        seed, keep = _attn_dropout(dropout_rate, training, 3)
        q = ops.linear(queries, wl[:, :d], bl[:d], eng._wslice(w, 0, d))
        if values is keys:
            kv = ops.linear(keys, wl[:, d:], bl[d:], eng._wslice(w, d, 3 * d))
        else:
            # K = keys W_k + b_k (dense_1), V = values W_v + b_v (dense_2): two projections, packed side by side for the attention core
            kk = ops.linear(keys, wl[:, d:2 * d], bl[d:2 * d], eng._wslice(w, d, 2 * d))
            vv = ops.linear(values, wl[:, 2 * d:], bl[2 * d:], eng._wslice(w, 2 * d, 3 * d))
            kv = torch.cat([kk, vv], dim=-1)
        s = ops.AttnFn.apply(q, kv, queries, ql, kl, num_heads, d, False, seed, keep, None, causal)
    return ops.layer_norm(s, gamma, beta, 1e-8)


def ff(inputs, num_units, scope="positionwise_feedforward"):
    with R.variable_scope(scope):
        w1l, w1 = _leaf("dense/kernel")
        b1l, _ = _leaf("dense/bias")
        w2l, w2 = _leaf("dense_1/kernel")
        b2l, _ = _leaf("dense_1/bias")
        gamma, _ = _leaf("ln/gamma")
        beta, _ = _leaf("ln/beta")
    if list(num_units) != [w1.f32.shape[1], w2.f32.shape[1]]:
        raise ValueError("num_units %s does not match the stored FFN %s" % (list(num_units), [w1.f32.shape[1], w2.f32.shape[1]]))
    s = ops.FFNFn.apply(inputs, w1l, b1l, w2l, b2l, w1, w2)
    return ops.layer_norm(s, gamma, beta, 1e-8)


def positional_encoding_learn(inputs, maxlen, masking=False, scope="positional_encoding_learn"):
    """Returns P[0:T] broadcast to inputs' shape (lookup by range(T))."""
    with R.variable_scope(scope):
        pos, _ = _leaf("embedding_position_learn")
    zeros = torch.zeros_like(inputs)
    out = ops.ScaleAddPosFn.apply(zeros, pos, 0.0)
    if masking:          # :311-312: positions whose INPUT element is 0 keep the input (i.e. 0)
        out = torch.where(inputs == 0, inputs, out)
    return out


_SINCOS = {}


def positional_encoding(inputs, maxlen, masking=False, scope="positional_encoding"):
    """Sinusoidal positions (TransformerModel_util.py:238-279): PE[pos, i] = pos / 10000^((i - i % 2) / E), sin on the even columns,
    cos on the odd ones, computed in float64 on the host as the reference does (numpy) and kept as a float32 constant per
    (maxlen, E, device); returns PE[0:T] broadcast to the inputs' shape.  No variable is created (the reference's is a constant too)."""
    import numpy as np
    E, T = int(inputs.shape[-1]), int(inputs.shape[1])
    if T > maxlen:
        raise ValueError("positional_encoding: sequence length %d exceeds maxlen %d" % (T, maxlen))
    key = (int(maxlen), E, str(inputs.device))
    tab = _SINCOS.get(key)
    if tab is None:
        i = np.arange(E)
        enc = np.arange(maxlen, dtype=np.float64)[:, None] / np.power(10000.0, (i - i % 2) / float(E))[None, :]
        enc[:, 0::2] = np.sin(enc[:, 0::2])
        enc[:, 1::2] = np.cos(enc[:, 1::2])
        tab = torch.tensor(enc.astype(np.float32), device=inputs.device)
        _SINCOS[key] = tab
    out = ops.ScaleAddPosFn.apply(torch.zeros_like(inputs), tab, 0.0).float()
    if masking:          # :271-272
        out = torch.where(inputs == 0, inputs.float(), out)
    return out
