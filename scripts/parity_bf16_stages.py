"""Where does the bf16 engine leave the storage-rounding oracle?  Forward intermediates (encoder memory, MMoE input z, gates,
logits) of the HIP engine against the oracle with float64 sums, beside the oracle with float32 sums against the same: the second
column is what a different accumulation precision alone does to the same rounded function.

    python scripts/parity_bf16_stages.py [B]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import dmt_oracle_torch as OT                   # noqa: E402
from cikm2020_dmt_amd import spec as S                      # noqa: E402
from cikm2020_dmt_amd.data_feed.synthetic import make_batch  # noqa: E402
from cikm2020_dmt_amd.train import Trainer                  # noqa: E402
from scripts.parity_bf16_report import ROWS, params         # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)), float(np.abs(a - b).max())


def main(B):
    sp = S.scaled_spec(S.e64_spec(), ROWS)
    so = dict(sp)
    P = params(so)
    inputs, mask, label = make_batch(sp, B, seed=5, lengths="ragged", weights="random")
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, init=False, dropout=False)
    tr.store.load_state(P)
    batch = tr.make_batch(inputs, mask, label)
    with torch.no_grad():
        (c, o), yb = tr.engine.inference(batch)
    torch.cuda.synchronize()
    it = tr.engine.intermediates
    hip = {"memory_%d" % i: it["memory_%d" % i].float().cpu().numpy() for i in range(3)}
    K = tr.engine.plan.K
    hip["mmoe_input"] = it["zbuf"].float().cpu().numpy()[:, :K]
    g = it["gates"].float().cpu().numpy()          # [T, B, E]
    hip["gate_0"], hip["gate_1"] = g[0], g[1]
    hip["logit_click"], hip["logit_order"], hip["logit_bias"] = (x.float().cpu().numpy() for x in (c, o, yb))
    refs = {}
    for name, dt, st in (("o64", torch.float64, "bf16"), ("o32", torch.float32, "bf16"), ("exact", torch.float64, None)):
        Pt = OT.to_torch(P, dt, requires_grad=False)
        ((lc, lo), lyb), inter = OT.forward(Pt, inputs, so, return_intermediates=True, storage=st)
        d = {k: v.numpy() for k, v in inter.items() if k.startswith("memory") or k.startswith("gate") or k == "mmoe_input"}
        d["logit_click"], d["logit_order"], d["logit_bias"] = lc.numpy(), lo.numpy(), lyb.numpy()
        refs[name] = d
    lens = {i: None for i in range(3)}
    print("B = %d: relative L2 (max abs) against the storage-rounding oracle with float64 sums" % B)
    print("%-14s %-26s %-26s %-26s" % ("tensor", "HIP engine", "oracle, float32 sums", "exact oracle (no rounding)"))
    for k in ("memory_0", "memory_1", "memory_2", "mmoe_input", "gate_0", "gate_1", "logit_click", "logit_order", "logit_bias"):
        a = hip[k]
        r = refs["o64"][k]
        if k.startswith("memory"):
            # padded positions of the memory are not defined alike (never read: masked keys): compare valid positions only
            i = int(k[-1])
            uf = sp["attention_embed_pairs"][i][-1][0]
            ln = np.asarray(batch.feats[uf].lens.cpu().numpy())
            m = np.arange(a.shape[1])[None, :] < ln[:, None]
            sel = lambda x: np.asarray(x)[m]
        else:
            sel = lambda x: np.asarray(x)
        print("%-14s %-26s %-26s %-26s" % (k, "%.2e (%.2e)" % rel(sel(a), sel(r)), "%.2e (%.2e)" % rel(sel(refs["o32"][k]), sel(r)),
                                           "%.2e (%.2e)" % rel(sel(refs["exact"][k]), sel(r))))


if __name__ == "__main__":
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 352)
