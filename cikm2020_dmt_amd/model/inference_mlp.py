"""Model facade with the reference's interface (/root/reference/DMT_code/model/inference_mlp.py:16-280):
Inference(wnd_conf).inference / loss_multi_task[_unbias] / get_optimizer, as called by run_dnn.train()
(run_dnn.py:129-181)."""
from __future__ import annotations

import torch

from .. import ops
from ..optim import TFSlotOptimizer, make_optimizer
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
        # is_train=True runs the Transformer dropout 0.1 / bias-tower dropout 0.5 the reference always has in train()
        # (TransformerModel.py:101,151; mmoe_transformer_unbias.py:274-278).  The mask is counter-based: seed = dropout_seed + number
        # of training forward passes so far, or whatever set_step_seed() pinned for the next call.
        self.dropout_seed = int(seed) + 1
        self._train_calls = 0
        self._pinned_seed = None
        self._scorer = None

    def set_step_seed(self, step_seed):
        """Pin the dropout step seed of the NEXT inference(is_train=True) call (None: back to the running counter)."""
        self._pinned_seed = None if step_seed is None else int(step_seed)

    def inference(self, inputs, is_train=True, is_predict=False):
        eng = self.rt.engine
        if is_train and not is_predict:
            if self._pinned_seed is not None:
                eng.dropout_step_seed, self._pinned_seed = self._pinned_seed, None
            else:
                eng.dropout_step_seed = self.dropout_seed + self._train_calls
            self._train_calls += 1
        else:
            eng.dropout_step_seed = None
        return self.model.inference(inputs, is_train, is_predict)

    def online_inference(self, inputs, is_train=False):
        """inference_mlp.py:122-143 -> online_build_sparsetensor :73-113: the request carries ONE user's id lists (1-D, for every
        embedding_list entry of side 'u', plus '<f>Wts') and per-candidate item features for BatchSize candidates; the reference tiles
        the user lists across the batch and runs the predict graph.  Here serving.CandidateScorer does the same request but encodes
        the user's sequences once.  Returns (click_logit, order_logit), each [BatchSize, 1]."""
        import numpy as np
        from ..serving import CandidateScorer
        if is_train:
            raise NotImplementedError("online_inference is the serving path (is_train=False in export_model.py)")
        if self._scorer is None:
            self._scorer = CandidateScorer(self.rt.engine)
        spec = self.rt.spec
        B = int(inputs["BatchSize"])
        user, item = {}, {}
        for (_n, _r, _d, f, side) in list(spec["embedding_list"]) + list(spec["embedding_list_bias"]):
            if f in user or f in item:
                continue
            v, w = inputs[f], inputs.get(f + "Wts")
            if side == "u" and not hasattr(v, "to_padded") and np.asarray(v).ndim == 1:
                user[f] = (np.asarray(v), None if w is None else np.asarray(w))
            else:
                if hasattr(v, "to_padded"):
                    T = max(int(v.dense_shape[1]), 1)
                    idx, lens = v.to_padded(T)
                    wp = w.to_padded(T)[0] if (w is not None and hasattr(w, "to_padded")) else None
                else:
                    idx = np.asarray(v).reshape(B, -1)
                    lens = np.full((B,), idx.shape[1], dtype=np.int32)
                    wp = None if w is None else np.asarray(w).reshape(B, -1)
                item[f] = (idx, lens, wp)
        batch = self._scorer.tile_request(user, item, np.asarray(inputs["features"], dtype=np.float32))
        self.rt.engine.dropout_step_seed = None
        return self._scorer.logits(batch)

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
        if optimizer == "adam" or optimizer in TFSlotOptimizer.KINDS:
            return make_optimizer(optimizer, self.rt.store, learning_rate)
        print("Unknow optimizer, exit now")
        raise SystemExit(1)                          # inference_mlp.py:278-280
