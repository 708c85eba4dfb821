"""`base` with the reference's method names (/root/reference/DMT_code/model/net/base.py:12-195): variable access,
dense_layer, embedding tables and the mean-pooling combiner -- each backed by libdmt_hip.so kernels."""
from __future__ import annotations

import torch

from ... import ops
from .. import runtime as R


class base(object):
    def __init__(self, wnd_conf):
        self.wnd_conf = wnd_conf
        m = wnd_conf["model"] if hasattr(wnd_conf, "__getitem__") else {}
        self.is_bn = m.get("is_bn", False) if isinstance(m, dict) else False
        self.is_dropout = m.get("is_dropout", False) if isinstance(m, dict) else False
        if self.is_bn or self.is_dropout:
            raise NotImplementedError("is_bn / is_dropout are false in dmt.conf and not implemented")
        self.rt = R.get_default()

    def weight_bias(self, input_size, layer_size, bias_init):
        """base.py:28-37: the ('weights', 'biases') pair of the current variable scope."""
        st = self.rt.store
        scope = R.current_scope()
        W, b = st.leaf[scope + "weights"], st.leaf[scope + "biases"]
        if tuple(W.shape) != (input_size, layer_size):
            raise ValueError("%sweights has shape %s, expected %s" % (scope, tuple(W.shape), (input_size, layer_size)))
        return W, b

    def dense_layer(self, layer_name, inputs, input_size, layer_size, activation, bias_init=0.1, keep_prob=1.0, is_train=True):
        """base.py:39-68: activation(inputs @ W + b); activation in {'relu', 'identity', 'softmax'} (or the torch fns)."""
        act = getattr(activation, "__name__", activation)
        with R.variable_scope(layer_name):
            st = self.rt.store
            full = R.current_scope()
            W, b = self.weight_bias(input_size, layer_size, bias_init)
            y = ops.linear(inputs, W, b, st.weight[full + "weights"], relu=(act == "relu"))
        if act in ("relu", "identity"):
            return y
        if act == "softmax":
            gates = ops.MixFn.apply(torch.zeros((y.shape[0], y.shape[1]), dtype=y.dtype, device=y.device), y, y.shape[1], 1, 1)[1]
            return gates[0]
        raise NotImplementedError("activation %r" % (act,))

    def embedding(self, id_name, id_size, emb_dim, reuse=None, zero_pad=False):
        """base.py:81-91: the table variable `<id_name>/embedding` (a view of the HBM table arena).  zero_pad is a
        property of the LOOKUP here (index i reads row i-1, index 0 reads zeros), not a concatenated copy."""
        st = self.rt.store
        scope = R.current_scope()
        name = scope + "%s/embedding" % id_name
        if name not in st.table:
            raise KeyError("embedding table %s not in the variable inventory" % name)
        t = st.table[name]
        if tuple(t.shape) != (id_size, emb_dim):
            raise ValueError("table %s has shape %s, expected %s" % (name, tuple(t.shape), (id_size, emb_dim)))
        return t

    def embedding_combiner(self, inputs, is_train=True, combiner_type="mean"):
        """base.py:93-134 (sim_embed empty): [dense features | mean-pooled embedding of every emb entry] -> [B, 959]."""
        if combiner_type != "mean":
            raise NotImplementedError("combiner %s" % combiner_type)
        eng = self.rt.engine
        batch = self.rt.as_batch(inputs)
        _X, _tar, zbuf = eng.gather(batch)
        return zbuf[:, : eng.plan.interest_off]
