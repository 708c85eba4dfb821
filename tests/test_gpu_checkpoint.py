"""Checkpoint round trip with the reference's semantics: trainable variables only, step from the file name, Adam restarted."""
import os

import numpy as np
import pytest
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import checkpoint as CK
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer
from tests.util import small_specs

pytestmark = pytest.mark.gpu


def test_save_restore_and_resume_match_the_oracle(cuda, tmp_path):
    so, sp = small_specs()
    P = O.init_params(so, seed=9)
    tr = Trainer(sp, device="cuda", compute_dtype=torch.float32, init=False, dropout=False)
    tr.store.load_state(P)
    adam = O.TFAdam(lr=1e-3)
    Pr = {k: v.copy() for k, v in P.items()}
    batches = [make_batch(sp, 8, seed=40 + i, lengths="ragged", weights="random") for i in range(5)]
    for i in range(3):
        inputs, mask, _ = batches[i]
        tr.train_step(tr.make_batch(inputs, mask))
        _l, _lg, G = OT.loss_and_grads(Pr, inputs, mask, so)
        adam.apply(Pr, G)
    model_path = str(tmp_path / "model") + os.sep
    path = CK.save(tr, model_path)
    assert os.path.basename(path) == "model.ckpt-3.npz" and os.path.exists(model_path + "step-3.model.DONE")
    assert CK.latest(model_path) == "model.ckpt-3" and CK.step_of("model.ckpt-3") == 3 and CK.step_of("model.ckpt-current") == 0
    with np.load(path) as z:
        assert all(k.startswith("DnnModel/") for k in z.files)
        assert sorted(k[len("DnnModel/"):] for k in z.files) == sorted(P)              # trainable variables only: no slots, no step
    # a fresh process: restore -> identical variables, step 3, Adam slots restarted
    tr2 = Trainer(sp, device="cuda", compute_dtype=torch.float32, seed=123, dropout=False)
    assert CK.restore(tr2, model_path) == 3
    s1, s2 = tr.store.state_dict(), tr2.store.state_dict()
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k
    assert float(tr2.store.adam_m.abs().max()) == 0.0 and float(tr2.store.tab_v.abs().max()) == 0.0
    # resumed training == oracle continuing from the saved variables with a NEW Adam (slots are not in the checkpoint)
    adam2 = O.TFAdam(lr=1e-3)
    Pc = {k: np.asarray(v, np.float64) for k, v in s1.items()}
    for i in range(3, 5):
        inputs, mask, _ = batches[i]
        tr2.train_step(tr2.make_batch(inputs, mask))
        _l, _lg, G = OT.loss_and_grads(Pc, inputs, mask, so)
        adam2.apply(Pc, G)
    tr2.opt.flush_tables()
    got = tr2.store.state_dict()
    total = sum(Pc[k].size for k in Pc)
    n_off = sum(int((np.abs(got[k] - Pc[k]) > 2e-5).sum()) for k in Pc)
    worst = max(float(np.abs(got[k] - Pc[k]).max()) for k in Pc)
    assert n_off <= 1e-4 * total and worst < 5e-4, (n_off, total, worst)
    assert tr2.opt.global_step == 5
    with pytest.raises(KeyError):
        CK.restore_arrays(tr2, {"DnnModel/click/click-output/weights": np.zeros((4, 1), np.float32)})


def test_tensorflow_container_round_trip(cuda, tmp_path):
    """checkpoint.save(container="tf"): the reference's own checkpoint files (`model.ckpt-N.index` + `.data-00000-of-00001` + `checkpoint`,
    tf.train.Saver at run_dnn.py:258-261) written without TensorFlow (cikm2020_dmt_amd/tf_bundle.py), restored into a fresh trainer:
    every trainable variable back bit for bit, the step from the name, Adam restarted."""
    so, sp = small_specs()
    tr = Trainer(sp, device="cuda", compute_dtype=torch.float32, seed=5, dropout=False)
    for i in range(2):
        inputs, mask, _ = make_batch(sp, 8, seed=70 + i, lengths="ragged", weights="random")
        tr.train_step(tr.make_batch(inputs, mask))
    model_path = str(tmp_path / "model") + os.sep
    path = CK.save(tr, model_path, container="tf")
    assert sorted(os.listdir(model_path)) == ["checkpoint", "model.ckpt-2.data-00000-of-00001", "model.ckpt-2.index", "step-2.model.DONE"]
    assert path.endswith("model.ckpt-2.index") and CK.latest(model_path) == "model.ckpt-2"
    assert 'model_checkpoint_path: "model.ckpt-2"' in open(model_path + "checkpoint").read()
    from cikm2020_dmt_amd import tf_bundle
    names = set(tf_bundle.read_bundle(model_path + "model.ckpt-2"))
    assert names == {"DnnModel/" + k for k in tr.store.state_dict()}               # the graph names of the trainable variables, nothing else
    tr2 = Trainer(sp, device="cuda", compute_dtype=torch.float32, seed=99, dropout=False)
    assert CK.restore(tr2, model_path) == 2
    s1, s2 = tr.store.state_dict(), tr2.store.state_dict()
    for k in s1:
        assert np.array_equal(s1[k], s2[k]), k
    assert float(tr2.store.adam_m.abs().max()) == 0.0
