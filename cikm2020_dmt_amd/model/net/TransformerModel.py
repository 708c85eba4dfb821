"""TransformerModel with the reference's interface (/root/reference/DMT_code/model/net/TransformerModel.py:29-171):
encoder = self-attention over the behaviour sequence, decoder = the target item as the single query."""
from __future__ import annotations

import torch

from ... import ops
from .. import runtime as R
from .TransformerModel_util import ff, multihead_attention, positional_encoding, _leaf, _seq_index


class TransformerModel():
    def __init__(self, hp):
        self.hp = hp

    def _d_model(self):
        return self.hp.d_model if hasattr(self.hp, "d_model") else self.hp["d_model"]

    def _get(self, key, default=None):
        return getattr(self.hp, key) if hasattr(self.hp, key) else (self.hp.get(key, default) if isinstance(self.hp, dict) else default)

    def encode_decode(self, input, name="encode_decode", training=True):
        with R.variable_scope(name):
            (seq_q, seq_q_lens, seq_k, seq_k_lens, seq_k_ts) = input
            state_encode, _ = self.encode((seq_k, seq_k_lens, seq_k_ts), name, training=training)
            state_decode = self.decode((seq_q, seq_q_lens, state_encode, seq_k_lens), name, training=training)
            return state_decode.squeeze(1)

    def position_encode(self, seq_k, seq_k_ts, seq_max_len, scale):
        method = self._get("position_encoding_method", "position_learn")
        if method == "position_sin_cos":
            # TransformerModel.py:61-64: seq_k += positional_encoding(seq_k, maxlen) -- the sinusoid table is a constant, no variable
            pe = positional_encoding(seq_k, seq_max_len, masking=False, scope="positional_encoding_k_position_sin_cos")
            return ops.ScaleAddPosFn.apply(seq_k, None, scale) + pe.to(seq_k.dtype)
        if method in ("time_add", "time_concat") and self._get("is_use_seq_ts", False) and seq_k_ts is not None:
            # (:70-78: a dense layer over the time-stamp embedding, `dense_trans_seq_time_*`: variables the DMT inventory -- SURVEY.md
            #  Appendix B, dmt.conf -- does not hold.  The reference takes these branches ONLY with is_use_seq_ts set and a time-stamp
            #  tensor present; otherwise the scaled embedding passes through below, as it does there)
            raise NotImplementedError("position_encoding_method=%s with is_use_seq_ts needs the dense_trans_seq_%s variables, which dmt.conf's model does not create" % (method, method))
        if method != "position_learn":
            return ops.ScaleAddPosFn.apply(seq_k, None, scale)       # (:59-82: no branch matches, the scaled embedding passes through)
        with R.variable_scope("positional_encoding_k_position_learn"):
            pos, _ = _leaf("embedding_position_learn")
        return ops.ScaleAddPosFn.apply(seq_k, pos, scale)

    def encode(self, xs, name="encoder", training=True):
        rate = float(self._get("dropout_rate", 0.0) or 0.0) if training else 0.0
        eng = R.get_default().engine
        with R.variable_scope(name):
            seq_emb, seqlens, seq_k_ts = xs
            enc = self.position_encode(seq_emb, seq_k_ts, self._get("maxlen_k"), float(self._d_model()) ** 0.5)
            enc = ops.dropout(enc, rate, eng.dropout_step_seed if rate else None, 10 * _seq_index() + 0)   # TransformerModel.py:101
            for i in range(self._get("num_blocks_encode", 1)):
                with R.variable_scope("num_blocks_{}".format(i)):
                    enc = multihead_attention(queries=enc, keys=enc, values=enc, queries_length=seqlens, keys_length=seqlens,
                                              num_heads=self._get("num_heads"), dropout_rate=rate, training=training, causality=False,
                                              scope="self-attention")
                    enc = ff(enc, num_units=[self._get("d_ff"), self._d_model()])
        return enc, seqlens

    def decode(self, ys, name="decoder", training=True):
        with R.variable_scope(name):
            query_emb, query_length, key_emb, key_length = ys
            rate = float(self._get("dropout_rate", 0.0) or 0.0) if training else 0.0
            eng = R.get_default().engine
            dec = ops.ScaleAddPosFn.apply(query_emb, None, float(self._d_model()) ** 0.5)
            if self._get("is_decoder_add_pos_emb", False):       # TransformerModel.py:148-150: sinusoid positions of the query
                dec = dec + positional_encoding(dec, self._get("maxlen_q", dec.shape[1]), masking=False, scope="positional_encoding").to(dec.dtype)
            dec = ops.dropout(dec, rate, eng.dropout_step_seed if rate else None, 10 * _seq_index() + 1)     # TransformerModel.py:151
            for i in range(self._get("num_blocks_decode", 1)):
                with R.variable_scope("num_blocks_{}".format(i)):
                    dec = multihead_attention(queries=dec, keys=key_emb, values=key_emb, queries_length=query_length,
                                              keys_length=key_length, num_heads=self._get("num_heads"), dropout_rate=rate,
                                              training=training, causality=False, scope="vanilla_attention")
                    tied = R.get_default().spec.get("tie_ffn", True)
                    dec = ff(dec, num_units=[self._get("d_ff"), self._d_model()],
                             scope="positionwise_feedforward" if tied else "positionwise_feedforward_dec")
        return dec
