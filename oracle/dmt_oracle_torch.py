"""CPU ORACLE #2 (test infrastructure, NOT product code) -- independent torch restatement with autograd.

PARITY UNPINNED (see oracle/dmt_oracle.py header): cross-checks the numpy restatement (forward, fp64)
and supplies gradients (autograd) for the backward parity tests; timed as `cpu_baseline` (kind "port")
by bench.py.  Written independently of dmt_oracle.py: vectorised over padded [B,T] index tensors instead
of per-entry loops; the zero-pad table `[0;E]` (base.py:87-89) is realised as an index shift.

Citations relative to /root/reference/DMT_code/.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.

`storage="bf16"` (forward / loss_and_grads): the SAME function with every tensor the HIP engine keeps in bf16 rounded to bf16 at
the point where the engine stores it -- activations and their gradients (straight-through: the value is rounded on the way
forward, its gradient on the way back), the bf16 weight copies the GEMMs read (values only: parameter gradients stay fp32 there
and fp64 here) -- while every sum is still accumulated in float64.  What then remains between this oracle and the HIP path is
the accumulation order and the fp32 accumulators, so the parity bound of the benchmarked bf16 mode can be ~10x tighter than
against the unrounded function (tests/test_gpu_parity_bf16.py).  The unrounded mode stays the accuracy statement.
"""
from __future__ import annotations

import math
from typing import Dict

import numpy as np
import torch

PADDING_NUM = float(-2 ** 32 + 1)


def _bf(x):
    return x.to(torch.bfloat16).to(x.dtype)


class _RoundFn(torch.autograd.Function):
    """Storage rounding: value -> bf16 on the way forward (fwd), gradient -> bf16 on the way back (bwd)."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return _bf(x) if fwd else x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return (_bf(g) if ctx.bwd else g), None, None


class Storage:
    """Where the HIP engine rounds (cikm2020_dmt_amd, compute_dtype bf16).  R: a stored activation (and its stored gradient);
    Rw: a bf16 weight copy (gradient untouched); Rf: value only (an MFMA operand packed from fp32 registers); Rg: gradient only."""

    def __init__(self, mode=None):
        if mode not in (None, "bf16"):
            raise ValueError("storage must be None or 'bf16'")
        self.on = mode == "bf16"

    def R(self, x):
        return _RoundFn.apply(x, True, True) if self.on else x

    def Rw(self, x):
        return _RoundFn.apply(x, True, False) if self.on else x

    Rf = Rw

    def Rg(self, x):
        return _RoundFn.apply(x, False, True) if self.on else x


EXACT = Storage(None)


class _LNSavedRoundedFn(torch.autograd.Function):
    """LayerNorm whose forward runs on the exact pre-norm sum (dmt_chain2 normalises its fp32 accumulators) while the backward reads
    that sum back as the bf16 tensor the forward stored, with the forward's fp32 row statistics."""

    @staticmethod
    def forward(ctx, s, g, b, eps):
        mu = s.mean(-1, keepdim=True)
        rstd = torch.rsqrt((s - mu).pow(2).mean(-1, keepdim=True) + eps)
        ctx.save_for_backward(_bf(s), g, mu, rstd)
        return g * ((s - mu) * rstd) + b

    @staticmethod
    def backward(ctx, dy):
        sr, g, mu, rstd = ctx.saved_tensors
        xh = (sr - mu) * rstd
        dg = (dy * xh).reshape(-1, xh.shape[-1]).sum(0)
        db = dy.reshape(-1, xh.shape[-1]).sum(0)
        dxh = dy * g
        ds = rstd * (dxh - dxh.mean(-1, keepdim=True) - xh * (dxh * xh).mean(-1, keepdim=True))
        return ds, dg, db, None


def _padded(sp, dtype=torch.long):
    B, T = int(sp.dense_shape[0]), int(sp.dense_shape[1])
    ind = torch.as_tensor(np.asarray(sp.indices), dtype=torch.long).reshape(-1, 2)
    val = torch.as_tensor(np.asarray(sp.values)).to(dtype)
    dense = torch.zeros((B, T), dtype=dtype)
    valid = torch.zeros((B, T), dtype=torch.bool)
    if ind.numel():
        dense[ind[:, 0], ind[:, 1]] = val
        valid[ind[:, 0], ind[:, 1]] = True
    return dense, valid


def _drop(x, rate, step_seed, stream, mask_shape=None, perm=None):
    """x * mask / keep with the numpy oracle's counter-based mask (oracle/dmt_oracle.py:dropout_mask)."""
    if step_seed is None or not rate:
        return x
    from oracle.dmt_oracle import dropout_mask, site_seed
    keep = 1.0 - rate
    m = dropout_mask(site_seed(step_seed, stream), x.numel(), keep).reshape(tuple(x.shape))
    return torch.where(torch.as_tensor(m), x / keep, torch.zeros_like(x))


def _trans_prefix(i):
    return "embedding_trans/trans_sequence_%d/encode_decode_sequence_%d/encode_decode_sequence_%d/" % (i, i, i)


def _ln(x, g, b, eps=1e-8):
    mu = x.mean(-1, keepdim=True)
    var = (x - mu).pow(2).mean(-1, keepdim=True)
    return g * (x - mu) * torch.rsqrt(var + eps) + b          # TransformerModel_util.py:72-76


def _mha(q_in, kv_in, q_len, k_len, H, P, s, rate=0.0, step_seed=None, stream=0, st=EXACT):
    """TransformerModel_util.py:160-209 / :11-56.  [B,Tq,d],[B,Tk,d] -> [B,Tq,d]."""
    B, Tq, d = q_in.shape
    Tk = kv_in.shape[1]
    dh = d // H
    Q = st.R(q_in @ st.Rw(P[s + "dense/kernel"]) + P[s + "dense/bias"]).view(B, Tq, H, dh).transpose(1, 2)
    K = st.R(kv_in @ st.Rw(P[s + "dense_1/kernel"]) + P[s + "dense_1/bias"]).view(B, Tk, H, dh).transpose(1, 2)
    V = st.R(kv_in @ st.Rw(P[s + "dense_2/kernel"]) + P[s + "dense_2/bias"]).view(B, Tk, H, dh).transpose(1, 2)
    S = st.Rg((Q @ K.transpose(-1, -2)) / (dh ** 0.5))                            # [B,H,Tq,Tk]; dS is an MFMA operand (bf16)
    kmask = (torch.arange(Tk)[None, :] < k_len[:, None])[:, None, None, :]
    S = torch.where(kmask, S, torch.full_like(S, PADDING_NUM))
    A = torch.softmax(S, dim=-1)
    qmask = (torch.arange(Tq)[None, :] < q_len[:, None])[:, None, :, None]
    A = torch.where(qmask, A, torch.full_like(A, PADDING_NUM))                     # post-softmax query mask (F13)
    A = st.Rf(_drop(A, rate, step_seed, stream))                                   # [B,H,Tq,Tk] order == kernel index order
    O = (A @ V).transpose(1, 2).reshape(B, Tq, d)
    return st.R(_ln(st.R(O + q_in), P[s + "ln/gamma"], P[s + "ln/beta"]))


def _mha_q1mem(y, mem, k_len, H, P, s, rate=0.0, step_seed=None, stream=0, st=EXACT):
    """The SAME function as _mha(y, mem, ones, k_len) for one query per example, in the association the HIP engine computes it in
    (dmt_q1mem.hip: score_k = (Q_h Wk_h^T) . mem_k / sqrt(dh) -- the term Q_h . bk_h is constant over the keys and drops out of the
    softmax --, out_h = (sum_k P_k mem_k) Wv_h + (sum_k P_k) bv_h), so that its bf16 storage points can be restated."""
    B, _one, d = y.shape
    T = mem.shape[1]
    dh = d // H
    Q = st.R(y[:, 0] @ st.Rw(P[s + "dense/kernel"]) + P[s + "dense/bias"]).view(B, H, dh)
    Wk = st.Rw(P[s + "dense_1/kernel"]).view(d, H, dh)
    Wv = st.Rw(P[s + "dense_2/kernel"]).view(d, H, dh)
    bv = st.Rw(P[s + "dense_2/bias"]).view(H, dh)
    qp = st.R(torch.einsum("bhe,dhe->bhd", Q, Wk))                                  # [B,H,d]
    S = st.Rg(torch.einsum("bhd,btd->bht", qp, mem) / (dh ** 0.5))
    kmask = (torch.arange(T)[None, :] < k_len[:, None])[:, None, :]
    S = torch.where(kmask, S, torch.full_like(S, PADDING_NUM))
    A = torch.softmax(S, dim=-1)
    A = st.Rf(_drop(A.unsqueeze(2), rate, step_seed, stream).squeeze(2))            # index order [B,H,1,T] == [B,H,T]
    ctx = st.R(torch.einsum("bht,btd->bhd", A, mem))
    sp = st.R(A.sum(-1))
    O = (torch.einsum("bhd,dhe->bhe", ctx, Wv) + sp[..., None] * bv[None]).reshape(B, 1, d)
    return st.R(_ln(st.R(O + y), P[s + "ln/gamma"], P[s + "ln/beta"]))


def _ff(x, P, s, st=EXACT):
    h = st.R(torch.relu(x @ st.Rw(P[s + "dense/kernel"]) + P[s + "dense/bias"]))
    pre = h @ st.Rw(P[s + "dense_1/kernel"]) + P[s + "dense_1/bias"] + x
    if st.on:
        return st.R(_LNSavedRoundedFn.apply(pre, P[s + "ln/gamma"], P[s + "ln/beta"], 1e-8))
    return _ln(pre, P[s + "ln/gamma"], P[s + "ln/beta"])


def _lookup_zero_pad(E, idx):
    """[0;E][idx]  ==  E[idx-1] for idx>0, zeros for idx==0."""
    rows = E[(idx - 1).clamp_min(0)]
    return rows * (idx > 0).unsqueeze(-1).to(E.dtype)


def _pool_mean(E, idx, valid, w):
    rows = E[idx] * (w * valid.to(E.dtype)).unsqueeze(-1)
    ws = (w * valid.to(E.dtype)).sum(1, keepdim=True)
    return rows.sum(1) / torch.where(ws == 0, torch.ones_like(ws), ws)


def forward(P: Dict[str, torch.Tensor], inputs, spec, is_predict=False, return_intermediates=False, step_seed=None, storage=None):
    """mmoe_transformer_unbias.py:293-316.  storage="bf16": see the module header."""
    st = Storage(storage)
    any_p = next(iter(P.values()))
    dt = any_p.dtype
    feats_dense = st.R(torch.as_tensor(np.asarray(inputs["features"])).to(dt))
    B = feats_dense.shape[0]
    d, H = spec["d_model"], spec["num_heads"]
    table_of = {f: n for (n, _r, _d, f, _s) in spec["embedding_list"]}
    inter = {}

    # ---- generate_data + trans_core (:130-223)
    states = []
    for i, pairs in enumerate(spec["attention_embed_pairs"]):
        pre = _trans_prefix(i)
        seq_parts, tar_parts = [], []
        lens = None
        for (uf, itf) in pairs:
            idx, valid = _padded(inputs[uf])
            lens = valid.sum(1)
            seq_parts.append(_lookup_zero_pad(P["embedding_trans/%s/embedding" % table_of[uf]], idx))
            tidx = torch.as_tensor(np.asarray(inputs[itf].values), dtype=torch.long)
            tar_parts.append(_lookup_zero_pad(P["embedding_trans/%s/embedding" % table_of[itf]], tidx))
        seq_emb = torch.cat(seq_parts, -1)
        tar = torch.cat(tar_parts, -1)
        if spec.get("is_trans_input_by_mlp"):          # mmoe_transformer_unbias.py:196-198
            tp = "embedding_trans/trans_sequence_%d/" % i
            seq_emb = st.R(seq_emb @ P[tp + "dense_trans_seq_sequence_%d/kernel" % i] + P[tp + "dense_trans_seq_sequence_%d/bias" % i])
            tar = st.R(tar @ P[tp + "dense_trans_sku_sequence_%d/kernel" % i] + P[tp + "dense_trans_sku_sequence_%d/bias" % i])
        T = seq_emb.shape[1]
        rate = spec.get("dropout_rate", 0.0) if step_seed is not None else 0.0
        if spec.get("position_encoding_method", "position_learn") == "position_learn":
            pos_tab = P[pre + "positional_encoding_k_position_learn/embedding_position_learn"]
        else:
            # position_sin_cos (TransformerModel_util.py:238-279), vectorised: angle[pos, i] = pos / 10000^((i - i % 2) / E)
            ii = torch.arange(d, dtype=torch.float64)
            ang = torch.arange(spec["maxlen_k"], dtype=torch.float64)[:, None] / torch.pow(torch.tensor(10000.0, dtype=torch.float64), (ii - ii % 2) / d)[None, :]
            pos_tab = torch.where((torch.arange(d) % 2 == 0)[None, :], torch.sin(ang), torch.cos(ang)).float().to(seq_emb.dtype)
        x = seq_emb * (d ** 0.5) + pos_tab[:T][None]
        x = st.R(_drop(x, rate, step_seed, 10 * i + 0))
        for j in range(int(spec.get("num_blocks_encode", 1))):          # TransformerModel.py:104-121
            blk = pre + "num_blocks_%d/" % j
            x = _mha(x, x, lens, lens, H, P, blk + "self-attention/", rate, step_seed, 10 * i + 2 + 1000 * j, st)
            x = _ff(x, P, blk + "positionwise_feedforward/", st)
        mem = x
        y = tar * (d ** 0.5)
        if spec.get("is_decoder_add_pos_emb"):      # one-step query: sinusoid row 0 = (0, 1, 0, 1, ...)
            y = y + (torch.arange(d) % 2).to(y.dtype)
        y = st.R(y)[:, None, :]
        if rate and step_seed is not None:
            y = st.R(_drop(y, rate, step_seed, 10 * i + 1))
        ffs = "positionwise_feedforward/" if spec.get("tie_ffn", True) else "positionwise_feedforward_dec/"
        for j in range(int(spec.get("num_blocks_decode", 1))):          # TransformerModel.py:154-169
            blk = pre + "num_blocks_%d/" % j
            if st.on:
                y = _mha_q1mem(y, mem, lens, H, P, blk + "vanilla_attention/", rate, step_seed, 10 * i + 3 + 1000 * j, st)
            else:
                y = _mha(y, mem, torch.ones(B, dtype=torch.long), lens, H, P, blk + "vanilla_attention/", rate, step_seed, 10 * i + 3 + 1000 * j)
            y = _ff(y, P, blk + ffs, st)
        fin = y[:, 0, :]
        if spec.get("is_trans_out_concat_item"):
            fin = torch.cat([fin, tar], -1)
            if spec.get("is_trans_out_by_mlp"):
                tp = "embedding_trans/trans_sequence_%d/dense_trans_concat_sequence_%d/" % (i, i)
                fin = st.R(fin @ P[tp + "kernel"] + P[tp + "bias"])
        states.append(fin)
        inter["seq_emb_%d" % i] = seq_emb
        inter["tar_emb"] = tar
        inter["memory_%d" % i] = mem
    interest = torch.cat(states, -1)

    # ---- embedding_combiner (base.py:93-134)
    parts = [feats_dense]
    for (name, _r, _dim, feat, _s) in spec["embedding_list"]:
        idx, valid = _padded(inputs[feat])
        wsp = inputs.get(feat + "Wts")
        w = _padded(wsp, dtype=dt)[0] if wsp is not None else torch.ones(idx.shape, dtype=dt)
        parts.append(st.R(_pool_mean(P["embedding_trans/%s/embedding" % name], idx, valid, w)))
    z = torch.cat(parts + [interest], -1)
    inter["mmoe_input"] = z

    # ---- expert_gate (:63-105)
    experts = []
    for e in range(spec["num_experts"]):
        h = z
        for li in range(len(spec["hidden_units_bottom"])):
            s = "mmoe_layers/expert-%d/expert-layer-%d/" % (e, li)
            h = st.R(torch.relu(h @ st.Rw(P[s + "weights"]) + P[s + "biases"]))
        experts.append(h)
    ex = torch.stack(experts, -1)
    logits = []
    for t, nm in enumerate(("click", "order")[: spec["num_tasks"]]):
        s = "mmoe_layers/gates-%d/gates-layer-0/" % t
        g = torch.softmax(st.R(z @ st.Rw(P[s + "weights"]) + P[s + "biases"]), -1)
        inter["gate_%d" % t] = g
        m = st.R((ex * g[:, None, :]).sum(-1))
        h = m
        for li in range(len(spec["hidden_units_task"])):
            s2 = "%s/%s-fc-%d/" % (nm, nm, li)
            h = st.R(torch.relu(h @ st.Rw(P[s2 + "weights"]) + P[s2 + "biases"]))
        s2 = "%s/%s-output/" % (nm, nm)
        logits.append(st.Rg(h @ st.Rw(P[s2 + "weights"]) + P[s2 + "biases"]))     # fp32 logit; its gradient enters the layers as bf16
    logits = tuple(logits)
    if is_predict:
        return (logits, inter) if return_intermediates else logits

    # ---- bias tower (:235-289)
    bparts = []
    for (name, _r, _dim, feat, _s) in spec["embedding_list_bias"]:
        idx, valid = _padded(inputs[feat])
        wsp = inputs.get(feat + "Wts")
        w = _padded(wsp, dtype=dt)[0] if wsp is not None else torch.ones(idx.shape, dtype=dt)
        bparts.append(st.R(_pool_mean(P["%s/embedding" % name], idx, valid, w)))
    yb = torch.cat(bparts, -1)
    n = len(spec["hidden_units_bias"])
    for li in range(n):
        yb = st.R(torch.relu(yb @ st.Rw(P["layer_bias%d/kernel" % li]) + P["layer_bias%d/bias" % li]))
        rb = spec.get("dropout_rate_bias", [0.0] * n)[li] if step_seed is not None else 0.0
        if rb:
            yb = st.R(_drop(yb, rb, step_seed, 100 + li))
    yb = st.Rg(yb @ st.Rw(P["layer_bias%d/kernel" % n]) + P["layer_bias%d/bias" % n])
    out = (logits, yb)
    return (out, inter) if return_intermediates else out


def _xent(p, y, eps=1e-7):
    """inference_mlp.py:162-168 via keras sparse_categorical_crossentropy (clip, log, softmax-CE)."""
    q = torch.stack([1.0 - p.reshape(-1), p.reshape(-1)], -1).clamp(eps, 1.0 - eps)
    lg = torch.log(q)
    return torch.logsumexp(lg, -1) - lg.gather(1, y.reshape(-1, 1).long()).squeeze(1)


def loss_unbias(out, mask, spec, loss_unbias_method="two_head_add", loss_ctr_rel_method="ctr_rel"):
    """inference_mlp.py:173-223."""
    (c, o), yb = out
    dt = c.dtype
    mask = torch.as_tensor(np.asarray(mask)).to(dt)
    if loss_unbias_method == "two_head_multiply":
        p_ctr, p_cvr = torch.sigmoid(c) * torch.sigmoid(yb), torch.sigmoid(o) * torch.sigmoid(yb)
    else:
        p_ctr, p_cvr = torch.sigmoid(c + yb), torch.sigmoid(o + yb)
    y_clk = mask[:, 1:5].sum(-1)
    y_ord = mask[:, 3] + mask[:, 4]
    x_clk, x_ord = _xent(p_ctr, y_clk), _xent(p_cvr, y_ord)
    if loss_ctr_rel_method == "ctr_rel":
        x_clk = x_clk + _xent(torch.sigmoid(c), y_clk)
        x_ord = x_ord + _xent(torch.sigmoid(o), y_ord)
    wc = (mask * torch.tensor(spec["weight_ctr"], dtype=dt)).sum(-1)
    wo = (mask * torch.tensor(spec["weight_ecvr"], dtype=dt)).sum(-1)
    Bn = mask.shape[0]
    return spec["loss_weight"][0] * (wc * x_clk).sum() / Bn + spec["loss_weight"][1] * (wo * x_ord).sum() / Bn


def to_torch(P_np: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad) for k, v in P_np.items()}


def loss_and_grads(P_np, inputs, mask, spec, dtype=torch.float64, step_seed=None, storage=None):
    P = to_torch(P_np, dtype)
    out = forward(P, inputs, spec, step_seed=step_seed, storage=storage)
    loss = loss_unbias(out, mask, spec)
    loss.backward()
    grads = {k: (v.grad.numpy() if v.grad is not None else np.zeros(v.shape)) for k, v in P.items()}
    (c, o), yb = out
    return float(loss.detach()), (c.detach().numpy(), o.detach().numpy(), yb.detach().numpy()), grads


class TorchTrainer:
    """CPU stand-in for the reference train step (run_dnn.py:148-207): forward, loss, dense Adam on every
    variable with tf.train.AdamOptimizer arithmetic.  Used as bench.py's cpu_baseline (kind 'port')."""

    def __init__(self, P_np, spec, lr=1e-3, dtype=torch.float32, beta1=0.9, beta2=0.999, eps=1e-8):
        self.spec = spec
        self.P = to_torch(P_np, dtype)
        self.m = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in self.P.items()}
        self.lr, self.b1, self.b2, self.eps = lr, beta1, beta2, eps
        self.b1p, self.b2p = beta1, beta2

    def step(self, inputs, mask):
        for p in self.P.values():
            p.grad = None
        out = forward(self.P, inputs, self.spec)
        loss = loss_unbias(out, mask, self.spec)
        loss.backward()
        a = self.lr * math.sqrt(1.0 - self.b2p) / (1.0 - self.b1p)
        with torch.no_grad():
            for k, p in self.P.items():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                m, v = self.m[k], self.v[k]
                m.add_((g - m) * (1.0 - self.b1))
                v.add_((g * g - v) * (1.0 - self.b2))
                p.sub_((m * a) / (v.sqrt() + self.eps))
        self.b1p *= self.b1
        self.b2p *= self.b2
        return float(loss), out
