"""Checkpoints with the reference's semantics (run_dnn.py:258-261, 296-304, 379-388, 119-122).

  * what is saved: the TRAINABLE variables only (tf.train.Saver(var_list = trainable_variables(), max_to_keep=0)), under their
    graph names `DnnModel/<name>` -- no Adam slots, no beta powers, no global_step variable;
  * where: `<model_path>/model.ckpt-<step>` + an empty marker `<model_path>/step-<step>.model.DONE` written after it;
  * resume: the step comes from the checkpoint NAME (`model.ckpt-N`), the optimizer restarts its slots (TFAdam.reset_slots).

The container is an .npz (name -> fp32 array): TensorFlow's tensor-bundle files cannot be produced or read here (TF is absent);
the names and shapes are the exchange surface -- a TF checkpoint dumped name by name loads with `restore_arrays`.
The lazy table optimizer is flushed first, so the saved embedding rows are exactly what a dense Adam sweep would hold.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Optional

import numpy as np

PREFIX = "DnnModel/"


def checkpoint_name(step: int) -> str:
    return "model.ckpt-%d" % int(step)


def step_of(ckpt_name: str) -> int:
    """run_dnn.py:119-122: step = int(ckpt_name.split('-')[1]) unless the name contains 'current'."""
    base = os.path.basename(ckpt_name)
    if base.endswith(".npz"):
        base = base[:-4]
    if "current" in base:
        return 0
    return int(base.split("-")[1])


def save(trainer, model_path: str, step: Optional[int] = None) -> str:
    """saver.save(sess, model_path + 'model.ckpt', global_step=step); create_file(model_path, 'step-%d.model.DONE' % step)."""
    step = trainer.opt.global_step if step is None else int(step)
    os.makedirs(model_path, exist_ok=True)
    trainer.opt.flush_tables()
    state = trainer.store.state_dict()
    path = os.path.join(model_path, checkpoint_name(step) + ".npz")
    tmp = path + ".tmp.npz"
    np.savez(tmp, **{PREFIX + k: np.asarray(v, dtype=np.float32) for k, v in state.items()})
    os.replace(tmp, path)
    open(os.path.join(model_path, "step-%d.model.DONE" % step), "w").close()
    return path


def latest(model_path: str) -> Optional[str]:
    """The newest finished checkpoint (largest N with a step-N.model.DONE marker), or None."""
    best = -1
    if os.path.isdir(model_path):
        for fn in os.listdir(model_path):
            m = re.match(r"^step-(\d+)\.model\.DONE$", fn)
            if m and os.path.exists(os.path.join(model_path, checkpoint_name(int(m.group(1))) + ".npz")):
                best = max(best, int(m.group(1)))
    return checkpoint_name(best) if best >= 0 else None


def restore_arrays(trainer, arrays: Dict[str, np.ndarray], step: int = 0):
    """Load variables given under their graph names (with or without the 'DnnModel/' scope) and restart the optimizer at `step`."""
    state = {}
    for k, v in arrays.items():
        state[k[len(PREFIX):] if k.startswith(PREFIX) else k] = np.asarray(v)
    have = trainer.store.state_dict()
    missing = sorted(set(have) - set(state))
    if missing:
        raise KeyError("checkpoint lacks %d variable(s), e.g. %s" % (len(missing), missing[:3]))
    for k in have:
        if tuple(state[k].shape) != tuple(have[k].shape):
            raise ValueError("variable %s: checkpoint shape %s, model shape %s" % (k, state[k].shape, have[k].shape))
    trainer.store.load_state({k: state[k] for k in have})
    trainer.opt.reset_slots(step)


def restore(trainer, model_path: str, ckpt_name: Optional[str] = None) -> int:
    """saver.restore(sess, model_path + ckpt_name) (run_dnn.py:300-304); returns the step the run resumes at."""
    ckpt_name = ckpt_name or latest(model_path)
    if ckpt_name is None:
        raise FileNotFoundError("no finished checkpoint under %s" % model_path)
    fn = os.path.join(model_path, ckpt_name if ckpt_name.endswith(".npz") else ckpt_name + ".npz")
    with np.load(fn) as z:
        restore_arrays(trainer, {k: z[k] for k in z.files}, step_of(ckpt_name))
    return trainer.opt.global_step
