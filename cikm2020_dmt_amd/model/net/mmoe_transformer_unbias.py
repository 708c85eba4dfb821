"""DMT with the Bias Deep Neural Network: the shipped default model
(/root/reference/DMT_code/model/net/mmoe_transformer_unbias.py:18-316), same method names and returns."""
from __future__ import annotations

import ctypes as C

import torch

from ... import _lib as L
from ... import ops
from ...spec import trans_prefix
from .. import runtime as R
from .base import base
from .TransformerModel import TransformerModel


class _HP(dict):
    __getattr__ = dict.get


class mmoe_transformer_unbias(base):
    def __init__(self, wnd_conf):
        base.__init__(self, wnd_conf)
        sp = self.rt.spec
        self.output_units = sp["output_units"]
        self.hidden_units_bottom = sp["hidden_units_bottom"]
        self.hidden_units_task = sp["hidden_units_task"]
        self.num_experts = sp["num_experts"]
        self.hidden_units_bias = sp["hidden_units_bias"]
        self.seq_data = None
        self.interest_state = None

    # ---- mmoe_transformer_unbias.py:63-105
    def expert_gate(self, features, units, dropout_keep_prob_list, num_experts=4, num_tasks=3, is_train=True):
        sp = self.rt.spec
        if list(units) != list(sp["hidden_units_bottom"]) or num_experts != sp["num_experts"] or num_tasks != sp["num_tasks"]:
            raise ValueError("expert_gate arguments differ from the configured MMoE")
        eng = self.rt.engine                 # (the variables live under the reference's scope 'mmoe_layers', see variables.py)
        z = features
        if features.shape[1] != eng.plan.K:
            raise ValueError("features must be [B, %d]" % eng.plan.K)
        return eng.expert_gate(z)

    # ---- :107-126
    def build_tower(self, task_layer, units, dropout_keep_prob_list, name, is_train=True):
        return self.rt.engine.build_tower(task_layer, name)

    # ---- :130-186
    def generate_data(self, inputs):
        """-> list of [mask, lens, seq_emb, tar_sku_emb, seq_ts_emb] with the RAW (unscaled) embeddings.
        seq_ts_emb is None: it is dead under position_learn (TransformerModel.py:61-82)."""
        eng, sp = self.rt.engine, self.rt.spec
        batch = self.rt.as_batch(inputs)
        X, tar = eng.gather_raw(batch)
        out = []
        for i, pairs in enumerate(sp["attention_embed_pairs"]):
            col = batch.feats[pairs[-1][0]]
            T = X[i].shape[1]
            mask = (torch.arange(T, device=col.lens.device)[None, :] < col.lens[:, None]).to(torch.int32)
            out.append([mask, col.lens, X[i], tar, None])
        return out

    # ---- :189-223
    def trans_core(self, seq_data, is_train=True):
        sp = self.rt.spec
        hp = _HP(d_model=sp["d_model"], d_ff=sp["d_ff"], num_heads=sp["num_heads"], maxlen_k=sp["maxlen_k"],
                 num_blocks_encode=1, num_blocks_decode=1, position_encoding_method="position_learn",
                 dropout_rate=sp.get("dropout_rate", 0.0))
        states = []
        # the reference calls trans_core inside variable_scope('embedding_trans') (:227); open it when called directly
        import contextlib
        inside = R.current_scope().startswith("embedding_trans")
        for i, (seq_mask, seq_lens, seq_emb, tar_sku_emb, seq_ts_emb) in enumerate(seq_data):
            stag = "sequence_" + str(i)
            outer = contextlib.nullcontext() if inside else R.variable_scope("embedding_trans")
            with outer, R.variable_scope("trans_" + stag):
                m = TransformerModel(hp)
                seq_q = tar_sku_emb.unsqueeze(1)
                q_lens = torch.ones(seq_q.shape[0], dtype=torch.int32, device=seq_q.device)
                user_stat = m.encode_decode((seq_q, q_lens, seq_emb, seq_lens, seq_ts_emb), name="encode_decode_" + stag,
                                              training=bool(is_train) and self.rt.engine.dropout_step_seed is not None)
            states.append(user_stat)
        return torch.cat(states, -1)

    # ---- :226-233  (fast path: gather + Transformers fused through the engine)
    def embedding_trans(self, inputs, is_train=True):
        eng = self.rt.engine
        z = eng.embedding_trans(self.rt.as_batch(inputs))
        self._z = z
        return z[:, : eng.plan.K]

    # ---- :235-289
    def embedding_combiner_bias(self, inputs, is_train=True, combiner_type="mean"):
        eng = self.rt.engine
        z = getattr(self, "_z", None)
        if z is None:
            _X, _tar, z = eng.gather(self.rt.as_batch(inputs))
        return z[:, eng.plan.bias_off: eng.plan.bias_off + eng.plan.bias_width]

    def embedding_mlp_bias(self, inputs, is_train=True):
        eng = self.rt.engine
        z = getattr(self, "_z", None)
        if z is None:
            _X, _tar, z = eng.gather(self.rt.as_batch(inputs))
        return eng.embedding_mlp_bias(z)

    # ---- :293-316
    def inference(self, inputs, is_train=True, is_predict=False):
        """is_train decides the dropout of the forward pass through engine.dropout_step_seed, which Inference.inference sets
        (None for is_train=False); a caller that uses this class directly sets it the same way."""
        batch = self.rt.as_batch(inputs)
        eng = self.rt.engine
        if not is_train or is_predict:
            eng.dropout_step_seed = None
        out = eng.inference(batch, is_predict=is_predict)
        return out

    def l2_norm(self, inputs):
        """mmoe_transformer_unbias.py:42-60: sum over the embedding_list entries of l2_loss(E[unique ids of the feature]) times
        l2_emb_lambda / batch_size (tf.losses.get_regularization_losses() is empty: no layer registers a regularizer).  Reached only
        when wnd_wd > 1e-5 (run_dnn.py:174-175; dmt.conf has 0.0).  Value and gradient on the GPU (engine.L2NormFn): the gradient --
        lambda / batch_size times the row, per entry, on every distinct row of the batch -- joins the sparse embedding-gradient rows of
        the same backward pass (DMTEngine.sparse)."""
        eng = self.rt.engine
        batch = self.rt.as_batch(inputs)
        m = self.wnd_conf["model"] if self.wnd_conf is not None else {}
        lam = float(m.get("l2_emb_lambda", 0.01)) if isinstance(m, dict) else 0.01
        bs = float(m.get("batch_size", batch.B)) if isinstance(m, dict) else float(batch.B)
        return eng.l2_norm(batch, lam / bs)
