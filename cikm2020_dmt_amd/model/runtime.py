"""Runtime context shared by the reference-shaped classes: the TF 'default graph + variable store' stand-in.

The reference builds a TF graph whose variables live under `DnnModel/...`; here one `Runtime` owns the HBM arenas
(VariableStore), the kernel engine and the optimizer, and tracks the variable scope the functional ops
(TransformerModel_util.*) are called under.
"""
from __future__ import annotations

import contextlib
from typing import List, Optional

import torch

from ..engine import DeviceBatch, DMTEngine
from ..variables import VariableStore

_default: Optional["Runtime"] = None
_scope: List[str] = []


class Runtime:
    def __init__(self, spec: dict, device="cuda", compute_dtype=torch.float32, seed: int = 0, init: bool = True):
        self.spec = spec
        self.store = VariableStore(spec, device, compute_dtype, seed=seed, init=init)
        self.engine = DMTEngine(spec, self.store)

    def as_batch(self, inputs, mask=None, label=None) -> DeviceBatch:
        if isinstance(inputs, DeviceBatch):
            return inputs
        return DeviceBatch.from_inputs(inputs, self.spec, self.store.device, mask=mask, label=label)


def set_default(rt: Runtime):
    global _default
    _default = rt


def get_default() -> Runtime:
    if _default is None:
        raise RuntimeError("no Runtime: construct model.inference_mlp.Inference (or model.runtime.Runtime) first")
    return _default


@contextlib.contextmanager
def variable_scope(name: str):
    """tf.variable_scope(name, reuse=tf.AUTO_REUSE): names nest with '/'."""
    _scope.append(name)
    try:
        yield "/".join(_scope)
    finally:
        _scope.pop()


def current_scope() -> str:
    return "/".join(_scope) + ("/" if _scope else "")
