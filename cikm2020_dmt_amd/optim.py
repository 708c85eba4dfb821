"""tf.train.AdamOptimizer semantics on the flat arenas (dense sweep for the 3.4 M dense parameters, exact lazy
row updates for the embedding tables).

Reference: model/inference_mlp.py:264-273 (get_optimizer -> tf.train.AdamOptimizer(lr)), applied to the averaged
tower gradients by run_dnn.py:203-207; learning rate = tf.train.piecewise_constant(global_step, step_boundary,
learning_rate) (run_dnn.py:125-126, dmt.conf:79-80).
The reference densifies the IndexedSlices embedding gradients (run_dnn.py:45-80) so TF sweeps all 167 M table
parameters every step; `dmt_adam_sparse_rows` reproduces that arithmetic bit-for-bit but only touches rows when
they are next read (zero-gradient steps are replayed on the way in), see DESIGN.md §Adam.  Exactness: p is bit-identical to the dense
sweep for any gap; m and v are bit-identical for gaps up to ~150 + 64 steps and agree to ~1e-5 relative beyond (closed-form tail).
The learning-rate schedule must be non-increasing (checked in __init__).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from . import ops


class TFAdam:
    def __init__(self, store, learning_rate=(0.001, 0.0001), step_boundary=(300000000,), beta1=0.9, beta2=0.999, epsilon=1e-8,
                 max_steps=1 << 20):
        self.store = store
        self.lrs = list(learning_rate) if isinstance(learning_rate, (list, tuple)) else [float(learning_rate)]
        self.bounds = list(step_boundary)[: len(self.lrs) - 1]
        # The lazy rows' replay stops replaying p at the first zero-gradient step that no longer changes it and treats that as final
        # (csrc/dmt_optim.hip:catch_up).  True while lr_t never grows by more than the bias correction does -- i.e. for a non-increasing
        # piecewise schedule, which is what the reference runs (dmt.conf:79-80: 0.001 -> 0.0001).  A warm-up schedule could move p again
        # after it stopped, and the lazy rows would then differ from the dense sweep: refuse it instead of being silently inexact.
        if any(b > a for a, b in zip(self.lrs[:-1], self.lrs[1:])):
            raise ValueError("TFAdam: learning_rate %s increases; the exact lazy-row replay needs a non-increasing piecewise-constant "
                             "schedule (the reference's is: dmt.conf learning_rate = 0.001, 0.0001)" % (self.lrs,))
        self.b1, self.b2, self.eps = float(beta1), float(beta2), float(epsilon)
        dev = store.device
        self.state = torch.zeros(4, dtype=torch.float32, device=dev)
        self.state[0], self.state[1] = self.b1, self.b2
        self.lr_hist = torch.zeros(max_steps, dtype=torch.float32, device=dev)
        self.max_steps = max_steps
        self.global_step = 0
        self._step_base = 0          # global_step at which the device-side step counter / lr history last restarted
        self._begun = False          # begin() of the step in flight has run (it may run early: Trainer.train_step)
        self._applied = False        # a parameter update of the step in flight has been enqueued (no rollback past this point)
        self._early_event = None     # the last early catch-up on the index lane (reads state / lr_hist / stamp)
        self._broken = None          # why no further step may begin (a step failed between its parameter update and end())
        self.stamp = None            # int32 per local table row: "the optimizer step with this local number updates the row itself"
        self.tm = store.fill_table_map(L.TableMap())

    def current_lr(self) -> float:
        # tf.train.piecewise_constant: values[i] while step <= boundaries[i]
        for b, lr in zip(self.bounds, self.lrs):
            if self.global_step <= b:
                return lr
        return self.lrs[-1]

    def begin(self):
        # the lr history holds one entry per step since the last restart of the device-side counter (reset_slots / rebase), NOT per
        # global step: a run resumed from model.ckpt-1500000 starts it at 0.  When it fills up, pending lazy rows are flushed (so no
        # row needs an older entry) and the history restarts; the Adam state (m, v, beta powers) is untouched.
        if self._begun:              # (idempotent within a step: the Trainer calls it at the step's start, the optimizer phase again)
            return
        if self._broken is not None:
            raise RuntimeError(self._broken)
        self._applied = False
        if self.global_step - self._step_base + 1 >= self.max_steps:
            self.rebase()
        L.call("dmt_adam_begin_step", ops.p(self.state), ops.p(self.lr_hist), self.max_steps, float(self.current_lr()), self.b1,
               self.b2, ops.stream_ptr())
        self._begun = True

    def end(self):
        L.call("dmt_adam_end_step", ops.p(self.state), self.b1, self.b2, ops.stream_ptr())
        self.global_step += 1
        self._begun = False

    def abort_step(self):
        """Undo begin() for a step that will not be applied (forward / backward / a collective raised after Trainer._open_step had
        begun it early): the device-side step counter goes back by one (the next begin() rewrites the same lr-history entry with the
        same lr_t: global_step has not moved), the rows stamped for the abandoned step are un-stamped.  Rows that catch_up_early already
        advanced THROUGH the abandoned step stay exact: the retried step has the same number and step size, and what they received is
        the zero-gradient update the dense sweep gives them at that step."""
        if not self._begun:
            return
        if self._applied:
            # the dense and / or sparse update of this step has already been enqueued: rolling the counter back would make the retry
            # apply the same step number a second time on top of it.  Nothing exact is left to do from here.
            self._begun = False
            self._broken = ("TFAdam: a step failed after its parameter update had been enqueued (step %d); the optimizer state cannot be "
                            "rolled back -- restore from a checkpoint" % (self.global_step + 1))
            return
        if self._early_event is not None:
            # an early catch-up on the index lane reads state / lr_hist / stamp: order the rollback behind it
            ops.cur_stream(self.store.device).wait_event(self._early_event)
            self._early_event = None
        self.state.view(torch.int32)[3] -= 1
        self._begun = False
        if self.stamp is not None:
            self.stamp.zero_()

    def step_in_flight(self) -> int:
        """Local number (index into the lr history) of the optimizer step between begin() and end()."""
        assert self._begun
        return self.global_step - self._step_base + 1

    def stamp_rows(self, uniq, n_uniq, cap):
        """Mark the rows the step in flight updates itself (call after begin()): catch_up_early leaves them alone."""
        if self.stamp is None:
            self.stamp = torch.zeros_like(self.store.last_step)
        L.call("dmt_rows_stamp", C.byref(self.tm), ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(self.stamp), self.step_in_flight(), ops.stream_ptr())

    def catch_up_early(self, uniq, n_uniq, cap, to_step: int):
        """EARLY catch-up of a later batch's rows while step `to_step` is in flight (its lr_t is in the history since begin()): rows that
        step does not update (not stamped with its number) receive their pending zero-gradient updates THROUGH that step now -- the
        same arithmetic the dense sweep applies to them at that step -- so nothing is left to replay in front of the next gather."""
        s = self.store
        L.call("dmt_adam_catchup_rows_to", C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(self.state), ops.p(self.lr_hist), self.b1, self.b2, self.eps, int(to_step),
               ops.p(self.stamp) if self.stamp is not None else None, int(to_step), ops.stream_ptr())
        if s.device.type == "cuda":
            self._early_event = torch.cuda.Event()
            self._early_event.record(ops.cur_stream(s.device))

    def apply_dense(self, grad_scale: float = 1.0):
        s = self.store
        self._applied = True
        L.call("dmt_adam_dense", s.P, ops.p(s.params), ops.p(s.adam_m), ops.p(s.adam_v), ops.p(s.grads), float(grad_scale),
               ops.p(self.state), self.b1, self.b2, self.eps, None, ops.stream_ptr())

    def apply_sparse(self, sparse, grad_scale: float = 1.0):
        s = self.store
        uniq, n_uniq, grad_rows, cap = sparse
        if int(cap) == 0:
            return
        self._applied = True
        if grad_rows.dtype == torch.bfloat16:      # reduced rows straight off the data-parallel wire
            L.call("dmt_adam_sparse_rows_bf16", C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
                   ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(grad_rows), int(grad_rows.shape[1]), float(grad_scale),
                   ops.p(self.state), ops.p(self.lr_hist), self.b1, self.b2, self.eps, ops.stream_ptr())
            return
        L.call("dmt_adam_sparse_rows", C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(grad_rows), int(grad_rows.shape[1]), float(grad_scale),
               ops.p(self.state), ops.p(self.lr_hist), self.b1, self.b2, self.eps, ops.stream_ptr())

    def step(self, sparse=None, grad_scale: float = 1.0):
        """One optimizer step over store.grads (dense) and `sparse` = (uniq_keys, n_uniq, grad_rows, cap)."""
        self.begin()
        self.apply_dense(grad_scale)
        if sparse is not None:
            self.apply_sparse(sparse, grad_scale)
        self.end()
        self.store.refresh_shadows()

    def catch_up(self, uniq, n_uniq, cap):
        """Rows about to be gathered: replay their pending zero-gradient steps (through the last completed step)."""
        s = self.store
        L.call("dmt_adam_catchup_rows", C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(self.state), ops.p(self.lr_hist), self.b1, self.b2, self.eps,
               ops.stream_ptr())

    def reset_slots(self, global_step: int = 0):
        """What restoring a reference checkpoint leaves behind: tf.train.Saver(var_list=trainable_variables()) (run_dnn.py:258-261)
        stores neither the Adam slots nor beta1_power/beta2_power, so a resumed run starts them from their initial values while
        global_step (parsed from the checkpoint name, run_dnn.py:119-122) keeps driving the learning-rate schedule."""
        s = self.store
        for t in (s.adam_m, s.adam_v, s.tab_m, s.tab_v, s.last_step, self.lr_hist):
            t.zero_()
        self.state.zero_()
        self.state[0], self.state[1] = self.b1, self.b2
        self.global_step = int(global_step)
        self._step_base = int(global_step)
        self._begun = False
        if self.stamp is not None:
            self.stamp.zero_()

    def rebase(self):
        """Restart the per-step lr history without changing any value: replay every pending zero-gradient row update (flush), then
        mark all rows as up to date at local step 0."""
        self.flush_tables()
        L.call("dmt_adam_rebase", ops.p(self.state), ops.p(self.store.last_step), self.store.last_step.numel(), ops.stream_ptr())
        self.lr_hist.zero_()
        self._step_base = self.global_step
        if self.stamp is not None:
            self.stamp.zero_()           # (local step numbers restart: old stamps must not match new steps)

    def flush_tables(self):
        """Replay pending zero-gradient updates on every table row (before checkpoint / full-table export)."""
        s = self.store
        L.call("dmt_adam_flush_rows", C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               ops.p(self.state), ops.p(self.lr_hist), self.b1, self.b2, self.eps, ops.stream_ptr())

    def apply_dense_tables(self, dense_grads: dict, grad_scale: float = 1.0):
        """Dense sweep over whole tables with the same kernel as the dense parameters (what TF does literally).
        Test hook for the bitwise lazy == dense property; `dense_grads`: table tf_name -> fp32 [rows, dim] tensor.
        Call between begin() and end()."""
        s = self.store
        for name, g in dense_grads.items():
            info = s.tables[name]
            sl = slice(info.offset, info.offset + info.numel)
            L.call("dmt_adam_dense", info.numel, ops.p(s.tab_p[sl]), ops.p(s.tab_m[sl]), ops.p(s.tab_v[sl]), ops.p(g), float(grad_scale),
                   ops.p(self.state), self.b1, self.b2, self.eps, None, ops.stream_ptr())


class TFSlotOptimizer:
    """tf.train.{GradientDescent, Adagrad, Adadelta, RMSProp, Ftrl}Optimizer(learning_rate) -- the other branches of get_optimizer
    (model/inference_mlp.py:264-280), each with TF 1.12's constructor defaults since the reference passes the learning rate only
    (include/dmt_hip.h: dmt_opt_*, oracle/dmt_oracle.py:TFOptimizer).  Same calling sequence as TFAdam (the Trainer drives either).

    None of these moves a variable at a zero-gradient step (FTRL: after the first step, see below), so the embedding rows need no
    catch-up in front of the gather: catch_up / catch_up_early / stamp_rows are no-ops, and the slots' idle decay (rmsprop, adadelta)
    is replayed when the row is next updated or flushed.  FTRL's dense update recomputes EVERY element from (accum, linear, lr) at every
    step: zero for an element whose gradient has been zero so far (linear == 0) -- the reference, which densifies the embedding
    gradients (run_dnn.py:45-80), wipes every table row the first batch did not read -- and a rescaled value once the schedule changes
    the learning rate.  end() of the first step after the slots were (re)initialised, and of every step whose learning rate differs
    from the previous step's, sweeps the tables for that."""
    KINDS = {"sgd": L.DMT_OPT_SGD, "adagrad": L.DMT_OPT_ADAGRAD, "adadelta": L.DMT_OPT_ADADELTA, "rmsprop": L.DMT_OPT_RMSPROP,
             "ftrl": L.DMT_OPT_FTRL}
    HP = {"sgd": (0.0, 0.0, 0.0), "adagrad": (0.0, 0.0, 0.0), "adadelta": (0.95, 1e-8, 0.0), "rmsprop": (0.9, 0.0, 1e-10),
          "ftrl": (0.0, 0.0, 0.0)}
    SLOT0_INIT = {"adagrad": 0.1, "ftrl": 0.1, "rmsprop": 1.0}

    def __init__(self, kind, store, learning_rate=(0.001, 0.0001), step_boundary=(300000000,)):
        if kind not in self.KINDS:
            raise ValueError("unknown optimizer %r" % (kind,))
        self.kind, self.code, self.hp = kind, self.KINDS[kind], self.HP[kind]
        self.store = store
        self.lrs = list(learning_rate) if isinstance(learning_rate, (list, tuple)) else [float(learning_rate)]
        self.bounds = list(step_boundary)[: len(self.lrs) - 1]
        self.global_step = 0
        self._step_base = 0
        self._begun = self._applied = False
        self._broken = None
        self.stamp = None
        self.tm = store.fill_table_map(L.TableMap())
        self._last_lr = None                     # learning rate of the last completed step (FTRL's var depends on it)
        self._init_slots()

    current_lr = TFAdam.current_lr

    def _init_slots(self):
        s = self.store
        v0 = self.SLOT0_INIT.get(self.kind, 0.0)
        for t in (s.adam_m, s.tab_m):
            t.fill_(v0)
        for t in (s.adam_v, s.tab_v, s.last_step):
            t.zero_()

    def begin(self):
        if self._begun:
            return
        if self._broken is not None:
            raise RuntimeError(self._broken)
        self._applied = False
        self._begun = True

    def end(self):
        lr = float(self.current_lr())            # (of the step that ends: global_step has not moved yet)
        refresh = self.kind == "ftrl" and lr != self._last_lr
        self._last_lr = lr
        self.global_step += 1
        self._begun = False
        if refresh:                              # the first step, and every step the schedule changed the learning rate at
            self.flush_tables()

    def abort_step(self):
        if not self._begun:
            return
        self._begun = False
        if self._applied:
            self._broken = ("TFSlotOptimizer: a step failed after its parameter update had been enqueued (step %d); the optimizer state "
                            "cannot be rolled back -- restore from a checkpoint" % (self.global_step + 1))

    def step_in_flight(self) -> int:
        assert self._begun
        return self.global_step - self._step_base + 1

    def stamp_rows(self, uniq, n_uniq, cap):
        pass

    def catch_up(self, uniq, n_uniq, cap):
        pass

    def catch_up_early(self, uniq, n_uniq, cap, to_step: int):
        pass

    def apply_dense(self, grad_scale: float = 1.0):
        s = self.store
        self._applied = True
        L.call("dmt_opt_dense", self.code, s.P, ops.p(s.params), ops.p(s.adam_m), ops.p(s.adam_v), ops.p(s.grads), float(grad_scale),
               float(self.current_lr()), *self.hp, None, ops.stream_ptr())

    def apply_sparse(self, sparse, grad_scale: float = 1.0):
        s = self.store
        uniq, n_uniq, grad_rows, cap = sparse
        if int(cap) == 0:
            return
        self._applied = True
        L.call("dmt_opt_sparse_rows", self.code, C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               ops.p(uniq), ops.p(n_uniq), int(cap), ops.p(grad_rows), int(grad_rows.dtype == torch.bfloat16), int(grad_rows.shape[1]),
               float(grad_scale), self.step_in_flight(), float(self.current_lr()), *self.hp, ops.stream_ptr())

    def step(self, sparse=None, grad_scale: float = 1.0):
        self.begin()
        self.apply_dense(grad_scale)
        if sparse is not None:
            self.apply_sparse(sparse, grad_scale)
        self.end()
        self.store.refresh_shadows()

    def flush_tables(self):
        """Every row's slots brought to the last completed step (FTRL: var recomputed from the slots, never-updated rows zeroed), before a
        checkpoint / export.  Nothing to do before the first step since the slots were (re)initialised: TF's variables only change in
        apply_gradients (and an FTRL sweep over fresh slots would zero every table)."""
        if self._last_lr is None:
            return
        s = self.store
        L.call("dmt_opt_flush_rows", self.code, C.byref(self.tm), ops.p(s.tab_p), ops.p(s.tab_m), ops.p(s.tab_v), ops.p(s.last_step),
               self.global_step - self._step_base, float(self._last_lr), *self.hp, ops.stream_ptr())

    def reset_slots(self, global_step: int = 0):
        """As TFAdam.reset_slots: the reference's Saver keeps no slots (run_dnn.py:258-261); global_step keeps driving the schedule.
        FTRL after a restore (global_step > 0): the first step's sweep recomputes EVERY variable from (accum 0.1, linear 0) -- TF 1.12's
        dense ApplyFtrl does the same after a slot-less restore -- so every restored embedding row that step does not touch becomes 0.
        That is the reference's behaviour, and it throws the restored model away: said aloud here."""
        if self.kind == "ftrl" and int(global_step) > 0:
            import warnings
            warnings.warn("ftrl: slots reset at global_step %d (the reference's checkpoints hold no slots); the next step's dense sweep "
                          "recomputes every variable from fresh (accum, linear) slots, which ZEROES all restored embedding rows the step does "
                          "not touch -- TF 1.12 semantics, not a way to continue training a restored model" % int(global_step))
        self._init_slots()
        self._last_lr = None
        self.global_step = int(global_step)
        self._step_base = int(global_step)
        self._begun = False

    def rebase(self):
        pass

    def apply_dense_tables(self, dense_grads: dict, grad_scale: float = 1.0):
        """Test hook (as TFAdam.apply_dense_tables): the literal dense sweep over whole tables, between begin() and end()."""
        s = self.store
        for name, g in dense_grads.items():
            info = s.tables[name]
            sl = slice(info.offset, info.offset + info.numel)
            L.call("dmt_opt_dense", self.code, info.numel, ops.p(s.tab_p[sl]), ops.p(s.tab_m[sl]), ops.p(s.tab_v[sl]), ops.p(g),
                   float(grad_scale), float(self.current_lr()), *self.hp, None, ops.stream_ptr())
            base, nr = s.table_rows[name]
            s.last_step[base: base + nr] = self.step_in_flight()        # (swept densely: nothing pending on these rows)


def make_optimizer(name, store, learning_rate=(0.001, 0.0001), step_boundary=(300000000,), max_steps=1 << 20):
    """get_optimizer(optimizer, learning_rate) (model/inference_mlp.py:264-280) on the flat arenas."""
    if name == "adam":
        return TFAdam(store, learning_rate, step_boundary, max_steps=max_steps)
    if name in TFSlotOptimizer.KINDS:
        return TFSlotOptimizer(name, store, learning_rate, step_boundary)
    raise ValueError("Unknow optimizer %r (sgd, adadelta, adagrad, adam, ftrl, rmsprop)" % (name,))
