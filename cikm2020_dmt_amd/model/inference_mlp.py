"""Model facade with the reference's interface (/root/reference/DMT_code/model/inference_mlp.py:16-280):
Inference(wnd_conf).inference / loss_multi_task[_unbias] / get_optimizer, as called by run_dnn.train()
(run_dnn.py:129-181)."""
from __future__ import annotations

import torch

from .. import ops
from ..optim import TFAdam
from . import runtime as R


class Inference(object):
    def __init__(self, wnd_conf, device="cuda", compute_dtype=torch.float32, seed: int = 0, spec: dict = None):
        super(Inference, self).__init__()
        self.wnd_conf = wnd_conf
        self.model_type = wnd_conf["model"]["model_type"] if wnd_conf is not None else "mmoe_transformer_unbias"
        if self.model_type not in ("mmoe_transformer", "mmoe_transformer_unbias"):
            print("Unknown model, exit now")
            raise SystemExit(1)                      # inference_mlp.py:66-68
        spec = spec if spec is not None else wnd_conf.to_spec()
        self.rt = R.Runtime(spec, device=device, compute_dtype=compute_dtype, seed=seed)
        R.set_default(self.rt)
        from .net.mmoe_transformer import mmoe_transformer
        from .net.mmoe_transformer_unbias import mmoe_transformer_unbias
        self.model = (mmoe_transformer_unbias if self.model_type == "mmoe_transformer_unbias" else mmoe_transformer)(wnd_conf)

    def inference(self, inputs, is_train=True, is_predict=False):
        return self.model.inference(inputs, is_train, is_predict)

    def online_inference(self, inputs, is_train=False):
        raise NotImplementedError("serving-side input re-packing (inference_mlp.py:73-143) is outside the train hot path")

    def _mask(self, mask):
        dev = self.rt.store.device
        return mask if torch.is_tensor(mask) else torch.as_tensor(mask, dtype=torch.float32).to(dev)

    def loss_multi_task_unbias(self, logits, labels, mask, is_train=True, loss_unbias_method="two_head_add", loss_ctr_rel_method="ctr"):
        loss, _pc, _pv = self.rt.engine.loss_unbias(logits, self._mask(mask), loss_unbias_method, loss_ctr_rel_method)
        return loss

    def logit_loss_unbias(self, logits, labels, mask, is_train, loss_unbias_method, loss_ctr_rel_method):
        return self.loss_multi_task_unbias(logits, labels, mask, is_train, loss_unbias_method, loss_ctr_rel_method)

    def loss_multi_task(self, logits, labels, mask, is_train=True):
        (c, o) = logits
        eng = self.rt.engine
        loss, _pc, _pv = ops.LossUnbiasFn.apply(c, o, torch.zeros_like(c), self._mask(mask), eng.w_ctr, eng.w_ecvr,
                                                eng.spec["loss_weight"], 2, 0)
        return loss

    def logit_loss(self, logits, labels, mask, is_train=True):
        return self.loss_multi_task(logits, labels, mask, is_train)

    def l2_norm(self, inputs):
        return self.model.l2_norm(inputs)

    def get_optimizer(self, optimizer, learning_rate):
        print("Use the optimizer: {}".format(optimizer))
        if optimizer == "adam":
            return TFAdam(self.rt.store, learning_rate)
        print("Unknow optimizer, exit now")
        raise SystemExit(1)                          # inference_mlp.py:278-280 (only adam is configured / implemented)
