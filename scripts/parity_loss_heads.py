"""Isolate the loss kernel and the towers' output-layer gradients of the bf16 engine: d loss / d logits of dmt_loss_unbias against the
oracle's loss evaluated ON THE ENGINE'S OWN LOGITS (float64 autograd), and the output-layer bias gradients against the sum of those.

    python scripts/parity_loss_heads.py [B]
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


def bf(x):
    return torch.tensor(x).to(torch.bfloat16).double().numpy()


def main(B):
    sp = S.scaled_spec(S.e64_spec(), ROWS)
    so = dict(sp)
    P = params(so)
    inputs, mask, label = make_batch(sp, B, seed=5, lengths="ragged", weights="random")
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, init=False, dropout=False)
    tr.store.load_state(P)
    batch = tr.make_batch(inputs, mask, label)
    tr.sync_rows(batch)
    tr.store.zero_grad()
    out = tr.engine.inference(batch)
    (c, o), yb = out
    for t in (c, o, yb):
        t.retain_grad()
    loss, _pc, _pv = tr.engine.loss_unbias(out, batch.mask)
    loss.backward()
    torch.cuda.synchronize()
    hip = [t.grad.double().cpu().numpy().reshape(-1) for t in (c, o, yb)]
    lt = [torch.tensor(t.detach().double().cpu().numpy(), requires_grad=True) for t in (c, o, yb)]
    lref = OT.loss_unbias(((lt[0], lt[1]), lt[2]), mask, so)
    lref.backward()
    ref = [t.grad.numpy().reshape(-1) for t in lt]
    print("B = %d, loss HIP %.9g oracle(on HIP logits) %.9g" % (B, float(loss), float(lref)))
    for name, h, r in zip(("click", "order", "bias"), hip, ref):
        print("  d loss / d logit_%-5s: rel L2 %.3e, sum HIP %.6e ref %.6e (sum |.| %.4e), sum of bf16-rounded ref %.6e" % (
            name, np.linalg.norm(h - r) / np.linalg.norm(r), h.sum(), r.sum(), np.abs(r).sum(), bf(r).sum()))
    G = tr.store.grad_dict()
    for name, h in (("click/click-output/biases", hip[0]), ("order/order-output/biases", hip[1]), ("layer_bias2/bias", hip[2])):
        g = float(np.asarray(G[name]).reshape(-1)[0])
        print("  %-28s HIP %.6e | sum dl %.6e | sum bf16(dl) %.6e" % (name, g, h.sum(), bf(h).sum()))


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 352)
