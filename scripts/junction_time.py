"""How long the step's junction is in a free-running loop (no profiler): HIP-event span on the compute stream from the join of the three
sequence lanes (engine.junction_event) to the moment dL/dz exists (the MMoE layer-0 input gradient), i.e. layer-0 GEMM, experts, towers,
loss and their backward; and from there to the end of the backward.
    [DMT_EARLY_CATCHUP=1] python scripts/junction_time.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cikm2020_dmt_amd import spec as S                      # noqa: E402
from cikm2020_dmt_amd.data_feed.synthetic import make_batch  # noqa: E402
from cikm2020_dmt_amd.train import Trainer                  # noqa: E402


def main():
    sp = S.e64_spec()
    tr = Trainer(sp, device="cuda:0", compute_dtype=torch.bfloat16, seed=1234, dropout=True)
    batches = [tr.make_batch(*make_batch(sp, 4096, seed=7 + i, lengths="full", law="zipf")) for i in range(8)]
    spans = []
    eng = tr.engine
    orig_inf = eng.inference

    def inference(batch, *a, **k):
        out = orig_inf(batch, *a, **k)
        z = eng.intermediates.get("zbuf")
        rec = {"j0": eng.junction_event}
        if z is not None and z.requires_grad:
            def hook(g, rec=rec):
                e = torch.cuda.Event(enable_timing=True)
                e.record(torch.cuda.current_stream())
                rec["j1"] = e
                return g
            z.register_hook(hook)
        spans.append(rec)
        return out
    eng.inference = inference
    # timing-enabled junction event
    orig_et = eng.embedding_trans

    def embedding_trans(batch):
        z = orig_et(batch)
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream())
        eng.junction_event_t = e
        return z
    eng.embedding_trans = embedding_trans

    def step(i):
        b, nxt = batches[i % 8], batches[(i + 1) % 8]
        nxt._prep = None
        e0 = torch.cuda.Event(enable_timing=True)
        e0.record()
        loss = tr.train_step(b, prefetch=nxt)
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        spans[-1]["s0"], spans[-1]["s1"], spans[-1]["jt"] = e0, e1, eng.junction_event_t
        return loss
    for i in range(60):
        step(i)
    torch.cuda.synchronize()
    use = spans[20:]
    fwd = sum(r["s0"].elapsed_time(r["jt"]) for r in use) / len(use)
    jun = sum(r["jt"].elapsed_time(r["j1"]) for r in use) / len(use)
    rest = sum(r["j1"].elapsed_time(r["s1"]) for r in use) / len(use)
    print("step start -> lanes joined %.3f ms | junction (join -> dL/dz) %.3f ms | dL/dz -> step end %.3f ms | sum %.3f ms" % (fwd, jun, rest, fwd + jun + rest))


if __name__ == "__main__":
    main()
