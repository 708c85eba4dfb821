"""Cost of the data-parallel embedding-gradient merge at N ranks, measured on ONE GPU without the wire: N different batches give
N per-rank (row id, bf16 gradient row) sets, concatenated as the all-gather would leave them; timed: second-level sort + segment
reduce (Trainer.merge_gathered) and the lazy Adam over the union of rows."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cikm2020_dmt_amd import spec as S
from cikm2020_dmt_amd.data_feed.synthetic import make_batch
from cikm2020_dmt_amd.train import Trainer

dev = torch.device("cuda", 0)
sp = S.e64_spec()
tr = Trainer(sp, device=dev, compute_dtype=torch.bfloat16, seed=1, dropout=True)
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
keys, rows, ns = [], [], []
for r in range(W):
    inputs, mask, label = make_batch(sp, 4096, seed=100 + 1000 * r, lengths="full", law="zipf")
    b = tr.make_batch(inputs, mask, label)
    tr.forward_backward(b)
    uniq, n_uniq, grad_rows, cap = tr.engine.sparse
    n = int(n_uniq.item())
    keys.append(uniq[:n].clone()); rows.append(grad_rows[:n].to(torch.bfloat16)); ns.append(n)
cap = max(ns)
all_k = torch.full((W * cap,), tr.store.total_rows, dtype=torch.int32, device=dev)
all_r = torch.zeros((W * cap, rows[0].shape[1]), dtype=torch.bfloat16, device=dev)
for r in range(W):
    all_k[r * cap: r * cap + ns[r]] = keys[r]
    all_r[r * cap: r * cap + ns[r]] = rows[r]


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

# ---- owner-reduce form: rank o reduces only the pairs with key % W == o; the reduced bf16 shards are all-gathered
shards_k, shards_r, t_owner = [], [], []
for o in range(W):
    ks = torch.cat([keys[r][keys[r] % W == o] for r in range(W)])
    rs = torch.cat([rows[r][keys[r] % W == o] for r in range(W)])
    u2, n2, sr, _c = tr.merge_gathered(ks, rs)
    m = int(n2.item())
    shards_k.append(u2[:m].clone()); shards_r.append(sr[:m].to(torch.bfloat16))
    if o == 0:
        t_owner.append(timeit(lambda: tr.merge_gathered(ks, rs)))
        recv_pairs = int(ks.numel())
cap2 = max(int(k.numel()) for k in shards_k)
u_k = torch.full((W * cap2,), tr.store.total_rows, dtype=torch.int32, device=dev)
u_r = torch.zeros((W * cap2, rows[0].shape[1]), dtype=torch.bfloat16, device=dev)
for o in range(W):
    u_k[o * cap2: o * cap2 + shards_k[o].numel()] = shards_k[o]
    u_r[o * cap2: o * cap2 + shards_k[o].numel()] = shards_r[o]
n_dev = torch.full((1,), W * cap2, dtype=torch.int32, device=dev)
t_adam_owner = timeit(lambda: tr.opt.step((u_k, n_dev, u_r, W * cap2), grad_scale=1.0 / W))
owner = {"owner_recv_pairs": recv_pairs, "owner_merge_ms": round(t_owner[0], 3), "shard_rows_max": cap2,
         "owner_wire_MB_per_rank_recv": round((recv_pairs * (W - 1) / W * 132 + (W - 1) * cap2 * 132) / 1e6, 1),
         "owner_optimizer_step_union_ms": round(t_adam_owner, 3)}

merged = tr.merge_gathered(all_k, all_r)
t_merge = timeit(lambda: tr.merge_gathered(all_k, all_r))
t_adam = timeit(lambda: tr.opt.step(merged, grad_scale=1.0 / W))
t_adam1 = timeit(lambda: tr.opt.step(tr.engine.sparse, grad_scale=1.0))
print(json.dumps({"ranks": W, "rows_per_rank": ns[0], "gathered_rows": int(W * cap), "union_rows": int(merged[1].item()),
                  "wire_MB_per_rank_recv": round((W - 1) * cap * (all_r.shape[1] * 2 + 4) / 1e6, 1),
                  "merge_ms": round(t_merge, 3), "optimizer_step_union_ms": round(t_adam, 3), "optimizer_step_1rank_ms": round(t_adam1, 3), **owner}))
