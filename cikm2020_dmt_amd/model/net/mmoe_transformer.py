"""DMT without the bias tower (/root/reference/DMT_code/model/net/mmoe_transformer.py:14-249).  Same network as
mmoe_transformer_unbias minus embedding_mlp_bias; `inference(inputs, is_train)` returns (click_logit, order_logit)."""
from __future__ import annotations

from .mmoe_transformer_unbias import mmoe_transformer_unbias


class mmoe_transformer(mmoe_transformer_unbias):
    def inference(self, inputs, is_train=True, is_predict=False):
        return self.rt.engine.inference(self.rt.as_batch(inputs), is_predict=True)
