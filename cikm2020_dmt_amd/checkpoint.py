"""Checkpoints with the reference's semantics (run_dnn.py:258-261, 296-304, 379-388, 119-122).

  * what is saved: the TRAINABLE variables only (tf.train.Saver(var_list = trainable_variables(), max_to_keep=0)), under their
    graph names `DnnModel/<name>` -- no Adam slots, no beta powers, no global_step variable;
  * where: `<model_path>/model.ckpt-<step>` + an empty marker `<model_path>/step-<step>.model.DONE` written after it;
  * resume: the step comes from the checkpoint NAME (`model.ckpt-N`), the optimizer restarts its slots (TFAdam.reset_slots).

Two containers.  The default is an .npz (name -> fp32 array).  `container="tf"` writes TensorFlow's own checkpoint V2 files
(`model.ckpt-N.index` + `model.ckpt-N.data-00000-of-00001` + the `checkpoint` state file) with the format restated in
`tf_bundle.py` -- what `saver.restore(sess, model_path + ckpt_name)` of the reference reads -- and `restore` reads either.  TensorFlow is
absent here and the reference ships no checkpoint, so the TF container is pinned by its own invariants only (block checksums, footer
magic, round trips: tests/test_host.py); the names and shapes are the exchange surface either way (`restore_arrays`).
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


def _dist():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist, dist.get_rank(), dist.get_world_size()
    return None, 0, 1


def _write_npz(path: str, arrays: Dict[str, np.ndarray], tag: str):
    tmp = "%s.tmp-%s.npz" % (path, tag)            # (one temporary per writer: two ranks never truncate each other's file)
    np.savez(tmp, **arrays)
    os.replace(tmp, path)


def shard_name(step: int, rank: int, world: int) -> str:
    return "%s.shard-%05d-of-%05d" % (checkpoint_name(step), rank, world)


def save(trainer, model_path: str, step: Optional[int] = None, container: str = "npz") -> str:
    """saver.save(sess, model_path + 'model.ckpt', global_step=step); create_file(model_path, 'step-%d.model.DONE' % step).
    More than one rank: every rank calls it (it ends in a barrier).  Replicated tables: rank 0 writes the one file (every replica
    holds the same values).  Row-sharded tables: NO table is gathered -- rank r writes the rows it owns to
    `model.ckpt-N.shard-r-of-W.npz` (local row l = global row l * W + r), rank 0 writes the dense variables and the shard count to
    `model.ckpt-N.npz`; the DONE marker appears after all shard files exist."""
    if container not in ("npz", "tf"):
        raise ValueError("container must be 'npz' or 'tf'")
    step = trainer.opt.global_step if step is None else int(step)
    dist, rank, world = _dist()
    os.makedirs(model_path, exist_ok=True)
    trainer.opt.flush_tables()
    store = trainer.store
    path = os.path.join(model_path, checkpoint_name(step) + ".npz")
    sharded = store.shard is not None and store.shard[1] > 1
    if container == "tf":
        if sharded:
            raise NotImplementedError("the TensorFlow container holds whole variables: save row-sharded tables as .npz shards")
        from . import tf_bundle
        prefix = os.path.join(model_path, checkpoint_name(step))
        if rank == 0:
            arrays = {PREFIX + k: v for k, v in store.dense_state_dict().items()}
            for name in store.tables:
                arrays[PREFIX + name] = store.table[name].detach().float().cpu().numpy()[: store.tables[name].shape[0]]
            tf_bundle.write_bundle(prefix, arrays)
            done = sorted(int(m.group(1)) for m in (re.match(r"^model\.ckpt-(\d+)\.index$", fn) for fn in os.listdir(model_path)) if m)
            tf_bundle.write_checkpoint_state(model_path, checkpoint_name(step), [checkpoint_name(n) for n in done])
        if dist is not None and world > 1:
            dist.barrier()
        if rank == 0:
            open(os.path.join(model_path, "step-%d.model.DONE" % step), "w").close()
        if dist is not None and world > 1:
            dist.barrier()
        return prefix + ".index"
    if sharded:
        r, W = store.shard
        mine = {PREFIX + name: store.table[name].detach().float().cpu().numpy() for name in store.tables}
        _write_npz(os.path.join(model_path, shard_name(step, r, W) + ".npz"), mine, "r%d" % r)
    if rank == 0:
        arrays = {PREFIX + k: v for k, v in store.dense_state_dict().items()}
        if sharded:
            arrays["__table_shards__"] = np.array([store.shard[1]], dtype=np.int64)
        else:
            for name in store.tables:
                arrays[PREFIX + name] = store.table[name].detach().float().cpu().numpy()[: store.tables[name].shape[0]]
        _write_npz(path, arrays, "r0")
    if dist is not None and world > 1:
        dist.barrier()
    if rank == 0:
        open(os.path.join(model_path, "step-%d.model.DONE" % step), "w").close()
    if dist is not None and world > 1:
        dist.barrier()
    return path


def latest(model_path: str) -> Optional[str]:
    """The newest finished checkpoint (largest N with a step-N.model.DONE marker), or None."""
    best = -1
    if os.path.isdir(model_path):
        for fn in os.listdir(model_path):
            m = re.match(r"^step-(\d+)\.model\.DONE$", fn)
            if m and any(os.path.exists(os.path.join(model_path, checkpoint_name(int(m.group(1))) + ext)) for ext in (".npz", ".index")):
                best = max(best, int(m.group(1)))
    return checkpoint_name(best) if best >= 0 else None


def _expected_shapes(store) -> Dict[str, tuple]:
    """Variable name -> shape, from the store's declarations (no tensor is touched: with row-sharded tables a state_dict() would
    gather every whole table onto every rank)."""
    shapes = {name: tuple(v.shape) for name, v in store.views.items()}
    shapes.update({name: tuple(info.shape) for name, info in store.tables.items()})
    return shapes


def restore_arrays(trainer, arrays, step: int = 0, table_loader=None):
    """Load variables given under their graph names (with or without the 'DnnModel/' scope) and restart the optimizer at `step`.
    arrays: a mapping name -> array (an open .npz works: members are read one at a time).  table_loader(name) -> full [rows, dim]
    array for tables the mapping does not hold (sharded checkpoints)."""
    names = {(k[len(PREFIX):] if k.startswith(PREFIX) else k): k for k in arrays.keys() if not k.startswith("__")}
    want = _expected_shapes(trainer.store)
    missing = sorted(n for n in want if n not in names and not (table_loader is not None and n in trainer.store.tables))
    if missing:
        raise KeyError("checkpoint lacks %d variable(s), e.g. %s" % (len(missing), missing[:3]))
    for n, shp in want.items():          # one variable at a time: the host never holds more than one table
        a = np.asarray(arrays[names[n]]) if n in names else np.asarray(table_loader(n))
        if tuple(a.shape) != shp:
            raise ValueError("variable %s: checkpoint shape %s, model shape %s" % (n, a.shape, shp))
        trainer.store.load_state({n: a}, refresh=False)
    trainer.store.refresh_shadows()
    trainer.opt.reset_slots(step)


def restore(trainer, model_path: str, ckpt_name: Optional[str] = None) -> int:
    """saver.restore(sess, model_path + ckpt_name) (run_dnn.py:300-304); returns the step the run resumes at.  A checkpoint written
    with row-sharded tables restores into any layout / rank count: same shard count -> every rank reads its own shard file; otherwise
    each table is re-assembled on the host from the shard files (one table at a time) and re-sliced by load_state."""
    ckpt_name = ckpt_name or latest(model_path)
    if ckpt_name is None:
        raise FileNotFoundError("no finished checkpoint under %s" % model_path)
    base = ckpt_name[:-4] if ckpt_name.endswith(".npz") else ckpt_name
    step = step_of(base)
    store = trainer.store
    if not os.path.exists(os.path.join(model_path, base + ".npz")) and os.path.exists(os.path.join(model_path, base + ".index")):
        from . import tf_bundle
        restore_arrays(trainer, tf_bundle.read_bundle(os.path.join(model_path, base)), step)       # a TensorFlow checkpoint V2 bundle
        return trainer.opt.global_step
    with np.load(os.path.join(model_path, base + ".npz")) as z:
        if "__table_shards__" not in z.files:
            restore_arrays(trainer, z, step)
            return trainer.opt.global_step
        Wf = int(z["__table_shards__"][0])
        files = [os.path.join(model_path, "%s.shard-%05d-of-%05d.npz" % (base, r, Wf)) for r in range(Wf)]
        if store.shard is not None and store.shard[1] == Wf:
            # the same sharding: this rank's rows, as they lie
            dense = {k: z[k] for k in z.files if not k.startswith("__")}
            want = _expected_shapes(store)
            have = {(k[len(PREFIX):] if k.startswith(PREFIX) else k) for k in dense}
            missing = sorted(n for n in store.views if n not in have)
            if missing:      # (a truncated / mismatched dense file must not leave freshly initialised weights in place)
                raise KeyError("checkpoint %s lacks %d dense variables, e.g. %s" % (base, len(missing), missing[:3]))
            with np.load(files[store.shard[0]]) as zs:
                lacking = sorted(name for name in store.tables if PREFIX + name not in zs.files)
                if lacking:
                    raise KeyError("shard file %s lacks tables %s" % (files[store.shard[0]], lacking[:3]))
            for k, a in dense.items():
                n = k[len(PREFIX):] if k.startswith(PREFIX) else k
                if n in store.views:
                    if tuple(a.shape) != want[n]:
                        raise ValueError("variable %s: checkpoint shape %s, model shape %s" % (n, a.shape, want[n]))
                    store.load_state({n: a}, refresh=False)
            with np.load(files[store.shard[0]]) as zs:
                for name in store.tables:
                    loc = zs[PREFIX + name]
                    if tuple(loc.shape) != tuple(store.table[name].shape):
                        raise ValueError("table %s: shard shape %s, this rank holds %s" % (name, loc.shape, tuple(store.table[name].shape)))
                    store.load_local_rows(name, loc)
            store.refresh_shadows()
            trainer.opt.reset_slots(step)
            return trainer.opt.global_step

        def assemble(name):
            rows, dim = store.tables[name].shape
            full = np.empty((rows, dim), dtype=np.float32)
            for r in range(Wf):
                with np.load(files[r]) as zs:
                    part = zs[PREFIX + name]
                    n_r = len(range(r, rows, Wf))
                    full[r::Wf] = part[:n_r]
            return full

        restore_arrays(trainer, z, step, table_loader=assemble)
    return trainer.opt.global_step
