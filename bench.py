#!/usr/bin/env python
"""Headline benchmark: DMT train step (forward + backward + TF-Adam) throughput in samples/s.

Workload = BASELINE.json configs[1]: full DMT (3 target-as-query Transformers + MMoE + bias tower, 2 tasks), bf16
compute, "emb_dim=64" (every id field 64 wide -> d_model 320, d_ff 1280, 4 heads of 80; SURVEY.md §8d "E64"),
batch 4096 per GPU, sequences at full length 50/50/10, Zipf(1.05) ids over the reference vocabularies
(5 M SKU rows), synthetic data resident in HBM.  A step = one pass of the hot path over one batch: index sort,
lazy-Adam row catch-up, gather, Transformers, MMoE, heads, loss, backward, gradient exchange, optimizer.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

Rank 0 prints ONE JSON line (see README / DESIGN.md §Measurement for the roofline and cpu_baseline legs).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np
import torch


# kernel families whose arithmetic is a few operations per byte moved: priced against the HBM roof, not the MFMA peak
HBM_BOUND_FAMILIES = ("q1mem",)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=4096, help="per-GPU batch")
    ap.add_argument("--dims", default="e64", choices=["e64", "ref"])
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--law", default="zipf", choices=["zipf", "uniform"])
    ap.add_argument("--lengths", default="full", choices=["full", "ragged"], help="sequence lengths of the synthetic batches: full = every history at its "
                    "maximum (the headline, roofline runs); ragged = len ~ U{1..L} per example and sequence (SURVEY.md section 8d variant (ii))")
    ap.add_argument("--optimizer", default="adam", choices=["adam", "sgd", "adagrad", "adadelta", "rmsprop", "ftrl"],
                    help="get_optimizer branch (model/inference_mlp.py:264-280); the metric is quoted on adam (dmt.conf:70)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dropout", action="store_true", help="disable the train-mode dropout of the reference (0.1 / 0.5)")
    ap.add_argument("--long-seq", type=int, default=0, help="BASELINE long-seq variant: click / order histories of this length (e.g. 200) instead of 50")
    ap.add_argument("--attn-dtype", default="bf16", choices=["bf16", "fp8"], help="BASELINE configs[4]: fp8 = the long-sequence attention forward "
                    "(64 < L <= 256) multiplies in OCP e4m3 (v_mfma_f32_32x32x16_fp8_fp8); L <= 64 kernels are bf16 either way")
    ap.add_argument("--shard-tables", action="store_true", help="BASELINE configs[3]: row-sharded embedding tables (rank r holds rows id %% W == r; "
                    "all-to-all id / row / gradient-row exchange, owner-only Adam) instead of one replica per GPU")
    ap.add_argument("--sku-rows", type=int, default=0, help="SKU vocabulary (default: the reference's 5 M); configs[3] quotes 100000000")
    ap.add_argument("--fresh-batches", type=int, default=64, help="distinct resident batches the steps cycle through (a row's gap between two reads "
                    "is then the law's own up to this many steps; 4 was round 2's setting: every touched row at most 4 steps stale)")
    ap.add_argument("--age-tables", type=int, default=100000, help="start the run as if this many steps had been trained: per-row last-touch ages drawn "
                    "from the id law's inter-arrival distribution, Adam slots non-zero -- the lazy Adam's replay then does its steady-state work "
                    "from the first step (0: fresh tables, nothing to replay)")
    ap.add_argument("--cpu-batch", type=int, default=256)
    ap.add_argument("--cpu-warmup", type=int, default=5)
    ap.add_argument("--cpu-steps", type=int, default=20)
    ap.add_argument("--cpu-big-batch", type=int, default=4096, help="one extra CPU step at this batch (0: skip)")
    ap.add_argument("--cpu-budget", type=float, default=120.0, help="seconds of CPU work the baseline leg may take (it shortens itself to fit)")
    ap.add_argument("--record-files", type=int, default=8, help="input-inclusive leg (one GPU): this many files of --batch synthetic TFRecords are "
                    "written, parsed by libdmt_input.so on --parser-threads host threads, uploaded and trained on, beside the same batches "
                    "kept resident (0: skip the leg)")
    ap.add_argument("--record-steps", type=int, default=30)
    ap.add_argument("--parser-threads", type=int, default=32)
    ap.add_argument("--cpu-threads", type=int, default=32)
    ap.add_argument("--inline-events", default="chain2",
                    help="kernel families timed with HIP events INSIDE the timed region: a comma list of ops._Timed keys, 'all' or 'none'.  Every "
                         "pair of events costs the launch stream ~3 us (all families: ~90 pairs, +0.3 ms per step, measured); the default times the "
                         "dominant kernel there and every family in the serialised steps right after the region")
    ap.add_argument("--inline-every", type=int, default=10,
                    help="the in-region events are recorded on every n-th timed step (an event between two long kernels drains the stream: ~8 us "
                         "each around chain2, 0.2 ms per step if every step is timed)")
    args = ap.parse_args()

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU, the line the driver uses) and
    # hand this process over to the launcher.  A --gpus N request never runs -- or prints -- as fewer than N ranks.
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        import socket
        port = os.environ.get("MASTER_PORT")
        if not port:
            with socket.socket() as s:
                s.bind(("127.0.0.1", 0))
                port = str(s.getsockname()[1])
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible); the HIP path has no CPU fallback")
    # test hooks (a one-GPU box cannot run RCCL with two ranks): DMT_BENCH_ONE_DEVICE=1 puts every rank on cuda:0,
    # DMT_DIST_BACKEND=gloo swaps the transport.  The driver's runs use neither.
    if os.environ.get("DMT_BENCH_ONE_DEVICE") == "1":
        local_rank = 0
    backend = os.environ.get("DMT_DIST_BACKEND", "nccl")
    # DMT_BENCH_FORCE_DP=1 (with --gpus 1): a ONE-rank RCCL group and the full N-rank exchange path (all-reduces, all_to_all
    # to the owners, shard all-gather, bf16-row Adam) -- the floor of the data-parallel overhead, measurable on a one-GPU box
    force_dp = os.environ.get("DMT_BENCH_FORCE_DP") == "1" and world == 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if force_dp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from cikm2020_dmt_amd import ops
    from cikm2020_dmt_amd import spec as S
    from cikm2020_dmt_amd.data_feed.synthetic import make_batch
    from cikm2020_dmt_amd.train import Trainer
    # one process per GPU: each rank keeps to its share of the host's cores (its GPU's NUMA node when sysfs tells) -- launcher, autograd
    # thread and the input stage's parser pool alike (cikm2020_dmt_amd/parallel.py:pin_rank_to_cores; DMT_PIN_CORES=0: off)
    cpu_plan = {}
    if world > 1:
        from cikm2020_dmt_amd import parallel as _par
        cpu_plan = _par.pin_rank_to_cores(local_rank if os.environ.get("DMT_BENCH_ONE_DEVICE") != "1" else rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), local_rank)

    sp = S.e64_spec() if args.dims == "e64" else S.default_spec()
    if args.sku_rows:
        sp = S.scaled_spec(sp, {"Sku": args.sku_rows})
    seq_lens = None
    if args.long_seq:
        sp = dict(sp, maxlen_k=max(sp["maxlen_k"], args.long_seq))
        seq_lens = {grp[0][0]: args.long_seq for grp in sp["attention_embed_pairs"][:2]}
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    tr = Trainer(sp, device=dev, compute_dtype=cdt, seed=1234, dropout=not args.no_dropout, force_dp=force_dp,
                 table_layout="sharded" if args.shard_tables else "replicated", attn_dtype=args.attn_dtype, optimizer=args.optimizer)
    nb = max(2, args.fresh_batches)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(min(8, os.cpu_count() or 1)) as ex:      # (numpy releases the GIL: ~0.15 s per batch on 8 threads)
        raw = list(ex.map(lambda i: make_batch(sp, args.batch, seed=20200101 + 1000 * rank + i, lengths=args.lengths, law=args.law, seq_lens=seq_lens), range(nb)))
    batches = [tr.make_batch(inputs, mask, label) for (inputs, mask, label) in raw]
    del raw
    age_info = age_tables(tr, sp, args, seq_lens) if args.age_tables > 0 and args.optimizer == "adam" else None

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def step(i):
        b, nxt = batches[i % nb], batches[(i + 1) % nb]
        nxt._prep = None        # every step sorts the ids of one batch, as fresh batches would need: the NEXT one's, while this step runs
        return tr.train_step(b, prefetch=nxt)

    for i in range(args.warmup):
        step(i)
    barrier()
    inline = "all" if not tr.engine.seq_streams else args.inline_events        # (no serialised leg without sequence lanes: time everything here)
    ops.PROFILE = None if inline == "none" else {}
    keys_on = None if inline in ("all", "none") else set(inline.split(",")) | {"gather_fwd"}
    every = max(1, args.inline_every) if tr.engine.seq_streams else 1
    steps_k = len(range(0, args.steps, every))          # timed steps that carry events
    diag = {}
    t0 = time.perf_counter()
    for i in range(args.steps):
        ops.PROFILE_KEYS = keys_on if i % every == 0 else set()
        tr.diag = diag if i % every == 0 else None            # (the phase spans of step_phases_ms: the same sampled steps)
        loss = step(args.warmup + i)
    host_dt = time.perf_counter() - t0          # the host has ENQUEUED the timed steps (it runs ahead of the GPU when the step is GPU-bound)
    barrier()
    dt = time.perf_counter() - t0
    prof, ops.PROFILE = (ops.PROFILE or {}), None
    ops.PROFILE_KEYS = None
    tr.diag = None
    # the same kernels with the chip to themselves: a few extra steps (outside the timed region) with the sequences' streams serialised
    prof_x, steps_x = None, 4
    if tr.engine.seq_streams:
        tr.engine.seq_streams = False
        step(args.warmup + args.steps)
        torch.cuda.synchronize()
        ops.PROFILE = {}
        for i in range(steps_x):
            step(args.warmup + args.steps + 1 + i)
        torch.cuda.synchronize()
        prof_x, ops.PROFILE = ops.PROFILE, None
        tr.engine.seq_streams = True
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        dist.barrier()
        if rank != 0:
            dist.destroy_process_group()
    if rank != 0:
        return

    def agg(key):
        ent = prof.get(key, [])
        ms = sum(e0.elapsed_time(e1) for (e0, e1, _w) in ent)
        return len(ent), ms * 1e-3, sum(w for (_a, _b, w) in ent)

    # ---- roofline: every MFMA kernel family is timed with HIP events on its launch stream (ops._Timed); the family with the largest
    #      share of the step is `roofline`, the others are listed beside it
    peak = 2500.0 if args.dtype == "bf16" else 157.3
    fam_desc = {
        "chain2": "chain2_kernel<Geo<320,1280,320>> (dmt_chain.hip): fused feed-forward + residual + LayerNorm forward (mode 0) and its input-gradient pass (mode 1)",
        "gemm_bf16": "gemm_glds_kernel / gemm_dw_glds_kernel / gemm_kernel<bf16> (dmt_gemm.hip): QKV / decoder / MMoE / tower GEMMs and their gradients",
        "gemm_f32": "gemm_kernel<float> (dmt_gemm.hip)",
        "proj": "proj_kernel<PGeo<320,960>> (dmt_chain.hip): packed Q | K | V projection, input rows stationary in registers, weights streamed as an LDS image",
        "wgrad320": "wgrad320_kernel (dmt_dw.hip): K=320 / N=320 weight gradients over the long row dimension",
        "attn_long": "attn_long_fwd/bwd_kernel + attn_q1_long_kernel (dmt_attn_long.hip): flash-style attention core, 64 < T <= 256",
        "attn": "attn_fwd/bwd_co_kernel + attn_q1v_kernel (dmt_attn.hip): attention core, T <= 64",
        "mhsa_block": "mhsa_fwd_kernel (dmt_mhsa.hip): fused self-attention block",
        "mhsa_bwd": "mhsa_bwd_kernel (dmt_mhsa_bwd.hip): the block's backward in one launch, attention gradient + dx GEMM (opt-in: DMT_FUSED_MHSA_BWD=1)",
        "q1mem": "q1m_fwd/bwd_kernel (dmt_q1mem.hip): decoder cross attention over the raw memory rows (HBM-bound: memory rows read once, d mem written once)",
        "mmoe_experts": "mmoe_experts_fwd/bwd_kernel (dmt_mmoe.hip): expert layers 1-2 + gates + mixtures",
    }
    fams = []

    def agg_x(key):
        ent = (prof_x or {}).get(key, [])
        return len(ent), sum(e0.elapsed_time(e1) for (e0, e1, _w) in ent) * 1e-3, sum(w for (_a, _b, w) in ent)

    for key, desc_ in fam_desc.items():
        n_k, t_k, fl_k = agg(key)
        n_x, t_x, fl_x = agg_x(key)
        if (n_k == 0 or t_k <= 0) and (n_x == 0 or t_x <= 0):
            continue
        src = prof if n_k > 0 else prof_x
        byts = src.get(key + "_bytes", src.get("gemm_bytes", []) if key.startswith("gemm") else [])
        if n_k > 0:
            in_step = {"avg_launch_us": round(t_k / n_k * 1e6, 2), "achieved": round(fl_k / t_k / 1e12, 2), "frac": round(fl_k / t_k / 1e12 / peak, 4),
                       "note": "HIP events around the launch INSIDE the timed region (every %s timed step): with three sequence lanes in flight this includes "
                               "the time the launch queues behind, and shares the chip with, kernels of the other lanes" % ("%d-th" % every if every > 1 else "single")}
        else:
            in_step = None          # (--inline-events: this family is timed in the serialised steps only)
            n_k, fl_k = n_x * steps_k / steps_x, fl_x * steps_k / steps_x       # launches / work of the event-carrying steps, from the same step
        if n_x > 0 and t_x > 0:
            # the kernel's own duration: the same launches, same process, lanes serialised (steps_x extra steps right after the timed region)
            n_m, t_m, fl_m, how = n_x, t_x, fl_x, ("HIP events on the launch stream, %d steps of the same workload run right after the timed region with "
                                                   "the sequence lanes serialised (one kernel on the chip at a time)" % steps_x)
        else:
            n_m, t_m, fl_m, how = n_k, t_k, fl_k, "HIP events on the launch stream over the timed region"
        alg_bytes = (sum(byts) / len(byts)) if byts else None
        if key in HBM_BOUND_FAMILIES and alg_bytes:
            # an HBM kernel (its rows are read once and written once; its arithmetic is a few dot2 per byte): priced against the HBM roof
            gbs = alg_bytes * n_m / t_m / 1e9
            head = {"key": key, "kernel": desc_, "bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(gbs / 8000.0, 4)}
        else:
            head = {"key": key, "kernel": desc_, "bound": "mfma", "achieved": round(fl_m / t_m / 1e12, 2), "peak": peak, "unit": "TFLOP/s",
                    "frac": round(fl_m / t_m / 1e12 / peak, 4)}
        fams.append({**head, "traffic": None, "launches_per_step": round(n_k / max(steps_k, 1), 1),
                     "avg_launch_us": round(t_m / n_m * 1e6, 2), "measured": how,
                     "time_share": round((t_m / n_m) * (n_k / max(steps_k, 1)) / (dt / args.steps), 3),
                     "algorithmic_flop_per_launch": int(fl_k / n_k),
                     "algorithmic_bytes_per_launch": int(sum(byts) / len(byts)) if byts else None,
                     "in_step": in_step})
    fams.sort(key=lambda f: -f["time_share"])
    # HBM bytes per launch from the PMC counters cannot be collected from inside this process (rocprofv3 wraps it); the committed
    # measurement of the same command is quoted (profiles/r03_traffic.json, made by scripts/pmc_traffic.sh: separate --pmc passes,
    # FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes)
    # (the newest committed counter file whose kernel-source sha is this build's: profiles/rNN_traffic.json)
    import glob
    default_cfg = (args.dims == "e64" and args.dtype == "bf16" and args.batch == 4096 and not args.long_seq and not args.shard_tables and not args.sku_rows
                   and args.fresh_batches == 64 and args.age_tables == 100000 and args.lengths == "full" and args.optimizer == "adam")
    tj, traffic_stale, tname = {}, None, None
    if default_cfg:
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_traffic.json")), reverse=True)
        for tfile in cands:
            try:
                tall = json.load(open(tfile))
            except Exception:
                continue
            # the counters describe the kernels they were collected from: a traffic file made from other kernel sources is not quoted
            if tall.get("kernel_source_sha") == kernel_source_sha():
                tj, tname, traffic_stale = tall.get(args.law, {}), "profiles/" + os.path.basename(tfile), None
                break
            if traffic_stale is None:
                traffic_stale = "profiles/%s was collected from other kernel sources (sha %s, now %s): not quoted" % (
                    os.path.basename(tfile), str(tall.get("kernel_source_sha"))[:12], kernel_source_sha()[:12])
    for f in fams:
        t = tj.get(f["key"])
        if traffic_stale:
            f["traffic_source"] = traffic_stale
        if t:
            f["traffic"] = int(t["hbm_bytes_per_step"] / max(f["launches_per_step"], 1e-9))      # per launch as counted here
            f["traffic_source"] = "%s (rocprofv3 --pmc FETCH_SIZE x2 [gfx950] + WRITE_SIZE, separate passes, same command, %d launches)" % (tname, t["launches"])
    # `roofline` names ONE kernel (its rocprofv3 row must agree): the single-kernel family with the largest share of the step; the
    # dmt_gemm family spans three kernels and is listed with the others
    single = [f for f in fams if f["key"] not in ("gemm_bf16", "gemm_f32", "attn", "attn_long", "q1mem", "mmoe_experts", "proj")] or fams
    roofline = dict(single[0]) if fams else {"kernel": None, "bound": "mfma", "achieved": None, "peak": peak, "unit": "TFLOP/s", "frac": None, "traffic": None}
    ga_x = bool((prof_x or {}).get("gather_fwd"))
    n_ga, t_ga, by_ga = agg_x("gather_fwd") if ga_x else agg("gather_fwd")
    ga_per_step = n_ga / max(steps_x if ga_x else steps_k, 1)
    gather = {"kernel": "gather_group_kernel (embedding gather+concat+pool fwd)", "bound": "hbm",
              "achieved": round(by_ga / t_ga / 1e9, 1) if t_ga > 0 else None, "peak": 8000.0, "unit": "GB/s",
              "frac": round(by_ga / t_ga / 8e12, 4) if t_ga > 0 else None, "avg_launch_us": round(t_ga / max(n_ga, 1) * 1e6, 2),
              "bytes_per_launch": int(by_ga / max(n_ga, 1)), "traffic": None}
    tg = tj.get("gather_fwd")
    if tg:
        gather["traffic"] = int(tg["hbm_bytes_per_step"] / max(ga_per_step, 1e-9))   # one dmt_gather_fwd call = its group kernels
        gather["traffic_source"] = "%s (same method; the Zipf head is served by L2 / Infinity Cache, uniform ids are not)" % tname

    out = {
        "metric": "train samples/sec", "value": round(args.batch * world * args.steps / dt, 1), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "full DMT train step (3 seq-Transformers + MMoE + bias tower, CTR+CTVR), %s dims d_model=%d d_ff=%d heads=%d, "
                               "per-GPU batch %d, L=%s %s%s, %s ids over %s/500/12k/190k/230k vocab%s, %s, train-mode dropout %s, "
                               "%d distinct resident batches, tables %s"
                               % (args.dims, sp["d_model"], sp["d_ff"], sp["num_heads"], args.batch, ("%d/%d/10" % (args.long_seq, args.long_seq)) if args.long_seq else "50/50/10", "full" if args.lengths == "full" else "ragged (len ~ U{1..L})",
                                  (" (flash-style attention kernels, %s MFMA forward%s)" % (args.attn_dtype, ": e4m3 attention DOES NOT MEET north_star's 1e-4 AUC bar (BASELINE.md section 5)" if args.attn_dtype == "fp8" else "")) if args.long_seq > 64 else "", args.law,
                                  ("%dM" % (args.sku_rows // 1000000)) if args.sku_rows >= 1000000 else ("%d" % args.sku_rows if args.sku_rows else "5M"),
                                  " (tables ROW-SHARDED over the ranks: %.1f GB of table+Adam state per rank)" % (tr.store.tab_p.numel() * 12 / 1e9) if args.shard_tables else "",
                                  "TF-Adam (exact lazy rows)" if args.optimizer == "adam" else "tf.train %s (sparse rows, lazily decayed slots)" % args.optimizer,
                                  "off" if args.no_dropout else "on (0.1 Transformer / 0.5 bias tower)", max(2, args.fresh_batches),
                                  ("pre-aged to step %d (per-row last-touch gaps from the id law: the lazy Adam replays them)" % args.age_tables) if args.age_tables > 0 and args.optimizer == "adam" else "fresh (nothing for the lazy rows to replay)"),
                   "global_batch": args.batch * world, "parallelism": ("dp%d" % world) + (" + row-sharded tables" if args.shard_tables else "") + (" (forced exchange path, one-rank RCCL group)" if force_dp else "")},
        "step_phases_ms": phases(diag, steps_k, world, tr), "lazy_adam": age_info,
        "roofline": roofline, "other_mfma_kernels": [f for f in fams if f["key"] != roofline.get("key")], "gather_roofline": gather, "final_loss": round(float(loss), 5),
    }
    # "bound by neither": algorithmic MFMA FLOP of every timed family over the step time against the bf16 dense peak, and the counter
    # HBM bytes of EVERY kernel of a step (profiles/rNN_traffic.json "__all__", same sha rule) against 8 TB/s
    if cpu_plan:
        cs = cpu_plan["cores"]
        out["cpu_affinity_rank0"] = {"n_cores": len(cs), "first": cs[0], "last": cs[-1], "numa_node": cpu_plan.get("numa_node")}
    out["host_enqueue_ms_per_step"] = round(host_dt / args.steps * 1e3, 3)   # < ms_per_step: the host runs ahead, the step is GPU-bound
    step_s = dt / args.steps
    flop_step = sum(f["algorithmic_flop_per_launch"] * f["launches_per_step"] for f in fams if f.get("unit") == "TFLOP/s")
    out["mfma_frac_whole_step"] = round(flop_step / step_s / (peak * 1e12), 4) if fams else None
    tall_ = tj.get("__all__")
    out["hbm_frac_whole_step"] = round(tall_["hbm_bytes_per_step"] / step_s / 8e12, 4) if tall_ else None
    out["whole_step_note"] = ("mfma: %.3f TFLOP of algorithmic MFMA work per step; hbm: %s" % (
        flop_step / 1e12, ("%.2f GB of counter traffic per step (%s, every kernel from the first gather on)" % (tall_["hbm_bytes_per_step"] / 1e9, tname))
        if tall_ else "no counter file for this build (profiles/rNN_traffic.json is tied to the kernel sources' sha)"))
    # wasted traffic as ONE number for the step: counter bytes over algorithmic bytes, over the families that carry both
    both = [f for f in fams if f.get("traffic") and f.get("algorithmic_bytes_per_launch")]
    if gather.get("traffic") and gather.get("bytes_per_launch"):
        both = both + [{"key": "gather_fwd", "traffic": gather["traffic"], "algorithmic_bytes_per_launch": gather["bytes_per_launch"], "launches_per_step": ga_per_step}]
    if both:
        alg = sum(f["algorithmic_bytes_per_launch"] * f["launches_per_step"] for f in both)
        cnt = sum(f["traffic"] * f["launches_per_step"] for f in both)
        out["algorithmic_hbm_bytes_per_step"] = {"families": [f["key"] for f in both], "algorithmic": int(alg), "counter": int(cnt), "counter_over_algorithmic": round(cnt / alg, 3),
                                                 "not_covered": "families without an algorithmic byte model (attention core, LayerNorm passes, embedding-gradient / Adam row kernels, MMoE / heads): "
                                                                "%.2f GB of the step's counter bytes" % ((tall_["hbm_bytes_per_step"] - cnt) / 1e9) if tall_ else None}
    if args.record_files > 0 and world == 1 and not force_dp and not args.shard_tables and not args.long_seq and args.dims == "e64":
        try:
            out["input_inclusive"] = input_inclusive(tr, sp, args, out["ms_per_step"])
        except Exception as e:          # (the leg must not take the headline line down with it)
            out["input_inclusive"] = {"error": "%s: %s" % (type(e).__name__, e)}
    if not args.no_cpu_baseline and world == 1:
        out["cpu_baseline"] = cpu_baseline(sp, args)
    print(json.dumps(out), flush=True)
    if world > 1 or force_dp:
        import torch.distributed as dist
        dist.destroy_process_group()


def kernel_source_sha():
    """sha256 over the kernel sources (csrc/*.hip, dmt_common.h): what a committed counter file is tied to."""
    import glob
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "cikm2020_dmt_amd", "csrc")
    for fn in sorted(glob.glob(os.path.join(d, "*.hip")) + [os.path.join(d, "dmt_common.h")]):
        h.update(os.path.basename(fn).encode())
        h.update(open(fn, "rb").read())
    return h.hexdigest()


def phases(diag, steps, world, tr):
    """Mean milliseconds per step of the phases Trainer.train_step brackets with HIP events (Trainer.diag): the index plane, the lazy
    Adam's row catch-up, the optimizer, and -- in a data-parallel step -- each collective from issue to completion as seen by the
    stream that waits for it (so time hidden behind compute still shows: these are spans, not exclusive times)."""
    import torch.distributed as dist
    out = {"world_size_seen_by_torch_distributed": dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1,
           "data_parallel_path": bool(tr._dp_active()), "table_layout": tr.table_layout,
           "note": "HIP-event spans on the stream that issues / waits for the phase, mean over the event-carrying timed steps (--inline-every); phases overlap each other and the step's compute"}
    for key, ent in sorted((diag or {}).items()):
        ms = [e0.elapsed_time(e1) for (e0, e1) in ent]
        if ms:
            out[key + "_ms"] = round(sum(ms) / max(steps, 1), 4)
    return out


def age_tables(tr, sp, args, seq_lens):
    """Put the tables and the optimizer in the state a run of K = --age-tables steps leaves behind, as far as the lazy Adam's cost is
    concerned: the device step counter and the lr_t history at K, every row's last-touch step K - age with age drawn from the row's
    own inter-arrival law, non-zero Adam slots on every row that was ever touched.
    A row's per-step touch probability follows from the id law of BASELINE.md section 3 (data_feed/synthetic.py): id = 1 + (z mod (V - 1)),
    z ~ Zipf(1.05) truncated at 2^63 as numpy draws it; a feature with T ids per example makes B * T draws per step; the pooled path
    touches row id, the Transformer path row id - 1 (base.py:87-89).  The model is checked against the run: `expected_distinct_rows`
    beside the measured distinct rows of a batch."""
    def zeta(s_, _q=1, N=4096):
        # Riemann zeta by Euler-Maclaurin (no scipy needed on the bench box): sum_{n<N} n^-s + N^(1-s)/(s-1) + N^-s/2 + s N^(-s-1)/12 - ...
        n = np.arange(1, N, dtype=np.float64)
        return float((n ** -s_).sum() + N ** (1.0 - s_) / (s_ - 1.0) + 0.5 * N ** -s_ + s_ * N ** (-s_ - 1.0) / 12.0
                     - s_ * (s_ + 1.0) * (s_ + 2.0) * N ** (-s_ - 3.0) / 720.0)
    K = int(args.age_tables)
    st, opt, dev = tr.store, tr.opt, tr.device
    B = args.batch
    a = 1.05
    xmax = 2.0 ** 63
    Z = float(zeta(a, 1)) - xmax ** (1 - a) / (a - 1)
    seq_feats = set()
    for grp in sp["attention_embed_pairs"]:
        for (uf, itf) in grp:
            seq_feats.update((uf, itf))
    per_table = {}
    groups = {grp[0][0]: [uf for (uf, _i) in grp] for grp in sp["attention_embed_pairs"]}
    len_of = {}
    for g0, ufs in groups.items():
        L_ = int((seq_lens or {}).get(g0, 0) or int(g0.rsplit("_", 1)[1]))
        for uf in ufs:
            len_of[uf] = L_
    for gi, tsf in enumerate(sp["attention_embed_seq_ts"]):
        len_of[tsf] = len_of[sp["attention_embed_pairs"][gi][0][0]]
    for (n, rows, _d, f, side) in sp["embedding_list"]:
        T = 1 if side == "i" else len_of.get(f, 1)
        e = per_table.setdefault("embedding_trans/%s/embedding" % n, [rows, 0, 0])
        e[1] += B * T
        if f in seq_feats:
            e[2] += B * T
    for (n, rows, _d, f, side) in sp["embedding_list_bias"]:
        e = per_table.setdefault("%s/embedding" % n, [rows, 0, 0])
        e[1] += B * (1 if side == "i" else 6)
    g = torch.Generator(device=dev)
    g.manual_seed(4242)
    expected = 0.0
    with torch.no_grad():
        opt.global_step, opt._step_base = K, 0
        t = torch.arange(1, K + 1, dtype=torch.float64, device=dev)
        lr_t = (opt.current_lr() * torch.sqrt(1.0 - opt.b2 ** t) / (1.0 - opt.b1 ** t)).float()
        opt.lr_hist[1: K + 1] = lr_t
        opt.state[0], opt.state[1], opt.state[2] = float(opt.b1 ** (K + 1)), float(opt.b2 ** (K + 1)), float(lr_t[-1])
        opt.state.view(torch.int32)[3] = K
        for name, (V, n_pool, n_seq) in per_table.items():
            base, rows = st.table_rows[name]
            j = torch.arange(0, rows + 1, dtype=torch.float64, device=dev)            # id j (0 is never drawn)
            if args.law == "uniform" or V <= 2:
                q = torch.full_like(j, 1.0 / max(V - 1, 1))
            else:
                w = float(V - 1)
                c = (j - 1.0).clamp_min(0.0)
                head = torch.where(j >= 2, c.clamp_min(1.0) ** (-a), torch.zeros_like(j))
                tail = ((c + 0.5 * w) ** (1 - a) - xmax ** (1 - a)) / ((a - 1) * w)        # sum over the wrapped copies z = j - 1 + m w, m >= 1 (midpoint rule)
                q = (head + tail) / Z
            q[0] = 0.0
            if V <= 1:
                q.zero_()
            log_miss = n_pool * torch.log1p(-q[:rows].clamp(max=0.999999)) + n_seq * torch.log1p(-q[1: rows + 1].clamp(max=0.999999))
            p_touch = (1.0 - torch.exp(log_miss)).clamp(1e-12, 1.0)
            expected += float(p_touch.sum())
            u = torch.rand(rows, generator=g, device=dev, dtype=torch.float64).clamp_min(1e-300)
            age = torch.floor(torch.log(u) / torch.log1p(-p_touch.clamp(max=1.0 - 1e-12))).clamp(0, K)      # geometric: steps since the last touch
            rk, W = st.shard if st.shard is not None else (0, 1)
            age = age[rk::W]                     # row-sharded tables: this rank holds the rows with id % W == rank, densely (local row = id // W)
            rows = int(age.numel())
            touched = age < K
            st.last_step[base // W: base // W + rows] = (K - age).clamp(0, K).to(torch.int32)
            info = st.tables[name]
            dim = info.shape[1]
            gs = 1e-4
            m = st.tab_m[info.offset: info.offset + rows * dim].view(rows, dim)
            v = st.tab_v[info.offset: info.offset + rows * dim].view(rows, dim)
            chunk = 1 << 20
            for r0 in range(0, rows, chunk):
                r1 = min(rows, r0 + chunk)
                tt = touched[r0:r1, None].float()
                m[r0:r1] = 0.1 * gs * torch.randn((r1 - r0, dim), generator=g, device=dev) * tt
                v[r0:r1] = 1e-3 * gs * gs * (1.0 + torch.rand((r1 - r0, dim), generator=g, device=dev)) * tt
    torch.cuda.synchronize()
    ages = (K - st.last_step.long()).clamp_min(0)
    live = ages < K
    return {"simulated_steps": K, "resident_batches": max(2, args.fresh_batches),
            "expected_distinct_rows_per_batch": int(expected),
            "rows_ever_touched_frac": round(float(live.float().mean()), 4),
            "median_age_of_touched_rows": int(ages[live].median()) if bool(live.any()) else None,
            "note": "per-row last-touch ages drawn from the id law's inter-arrival distribution at step K (bench.py:age_tables); the timed steps then "
                    "replay those gaps in dmt_adam_catchup_rows (bounded replay: DESIGN.md section 5)"}


def input_inclusive(tr, sp, args, resident_ms):
    """north_star: "throughput on synthetic TFRecords of the named shape".  The headline `value` keeps its batches resident in HBM
    (contract); this leg feeds the SAME trainer from TFRecord files: files -> libdmt_input.so (CRC checked, ids hashed into the
    vocabularies, --parser-threads host threads) -> one page-locked buffer -> one asynchronous upload -> train_step, a producer thread
    one batch ahead (data_feed/tfrecord_mask.py:120-158: the reference's tf.data pipeline + feed).  Beside it the very same
    record-derived batches kept resident, so the difference is the input stage and nothing else."""
    import multiprocessing as mp
    import queue
    import shutil
    import tempfile
    import threading
    from cikm2020_dmt_amd.data_feed import native
    from cikm2020_dmt_amd.data_feed.synthetic import write_records_file
    from cikm2020_dmt_amd.engine import DeviceBatch
    B, nf, K = args.batch, args.record_files, args.record_steps
    dev = tr.device
    tmp = tempfile.mkdtemp(prefix="dmt_records_")
    try:
        files = [os.path.join(tmp, "part-r-%05d" % i) for i in range(nf)]
        t0 = time.perf_counter()
        with mp.get_context("spawn").Pool(min(nf, 8, os.cpu_count() or 1)) as pool:
            sizes = pool.map(write_records_file, [(args.dims, args.sku_rows, B, 777000 + i, args.law, f) for i, f in enumerate(files)])
        t_write = time.perf_counter() - t0
        emb = list(sp["embedding_list"]) + list(sp["embedding_list_bias"])
        feats = list(dict.fromkeys(e[3] for e in emb))
        name_of = {e[3]: e[0] for e in reversed(emb)}
        vocabs = {}
        for (name, nrows, _d, _f, _s) in emb:
            vocabs.setdefault(name, native.Vocab(["unknow"], nrows) if nrows > 23 else native.Vocab(["unknow"] + [str(i) for i in range(1, nrows)], nrows))
        from cikm2020_dmt_amd.data_feed.synthetic import make_batch
        probe, _m, _l = make_batch(sp, 2, seed=1, lengths="full", law=args.law)
        T = {f: max(int(probe[f].dense_shape[1]), 1) for f in feats}
        threads = max(1, min(args.parser_threads, os.cpu_count() or 1))
        parser = native.BatchParser([(f, vocabs[name_of[f]], T[f]) for f in feats], [("features", sp["feature_dimension"]), ("mask", 5), ("label", 1)],
                                    n_threads=threads)
        parser.pinned = True
        parser.ring = 4                       # four page-locked output buffers in turn (a fresh 32 MB buffer per batch was 8 000 first-touch faults)

        up = torch.cuda.Stream(dev)           # the uploads get a stream of their own: on the compute stream a 42 MB copy queues between kernels

        def stream(n):
            k = 0
            while k < n:
                for cols in parser.batches(files, B, verify_crc=True):
                    with torch.cuda.stream(up):
                        b = DeviceBatch.from_columns(cols, sp, dev)
                    yield b
                    k += 1
                    if k >= n:
                        return

        # (a) the producer alone: parse + upload per batch (second pass: the first one pays the page-locked allocations and the
        #     worker pool's start)
        res = list(stream(nf))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _b in stream(nf):
            pass
        torch.cuda.synchronize()
        t_prod = (time.perf_counter() - t0) / nf
        del _b

        def run(get, n):
            cur = get()
            for _ in range(n):
                nxt = get()
                nxt._prep = None
                tr.train_step(cur, prefetch=nxt)
                cur = nxt

        # (b) the record-derived batches resident
        it = {"i": 0}

        def get_res():
            b = res[it["i"] % nf]
            it["i"] += 1
            return b
        run(get_res, 2 * nf)                 # (first visits of these rows: let the lazy Adam reach its steady state for this id set)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(get_res, K)
        torch.cuda.synchronize()
        ms_res = (time.perf_counter() - t0) / K * 1e3
        # (c) the same batches through the input stage every step
        q = queue.Queue(maxsize=3)

        def producer():
            for b in stream(K + 8):
                q.put(b)
        th = threading.Thread(target=producer, daemon=True)
        th.start()
        run(q.get, 5)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(q.get, K)
        torch.cuda.synchronize()
        ms_in = (time.perf_counter() - t0) / K * 1e3
        while th.is_alive():
            try:
                q.get(timeout=0.2)
            except queue.Empty:
                pass
        th.join()
        rec_s = B / t_prod
        need = B / (resident_ms * 1e-3)
        return {"value": round(B / (ms_in * 1e-3), 1), "unit": "samples/s", "ms_per_step": round(ms_in, 3),
                "same_batches_resident_ms_per_step": round(ms_res, 3), "input_stage_cost": round(ms_in / ms_res - 1.0, 4),
                "headline_resident_ms_per_step": resident_ms,
                "producer_alone_ms_per_batch": round(t_prod * 1e3, 3), "parser_threads": threads, "parser_records_per_s": round(rec_s, 0),
                "records": "%d files x %d synthetic tf.Example records of the model's schema (%.1f KB each, masked CRC-32C verified, every id string "
                           "hashed into its vocabulary), written in %.0f s by the Python encoder" % (nf, B, sizes[0] / B / 1e3, t_write),
                "note": "one GPU at the headline rate consumes %.0f records/s; the producer (parse + pinned buffer + one upload) delivers %.0f/s on %d "
                        "threads, i.e. %.1fx; eight GPUs on one host need %.1f M records/s = ~%d parser threads at this per-thread rate.  The %d "
                        "record batches revisit their rows every %d steps, so the lazy Adam replays less here than under the headline's "
                        "64-batch / aged-table protocol: compare ms_per_step with same_batches_resident_ms_per_step, not with the headline"
                        % (need, rec_s, threads, rec_s / need, 8 * need / 1e6, int(np.ceil(8 * need / (rec_s / threads))), nf, nf)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(sp, args):
    """The CPU oracle (independent torch restatement, fp32) on a bounded sample of the same workload, by BASELINE.md §3's protocol:
    5 warm-up + 20 timed steps at batch 256 (median step time), then ONE step at batch 4096.  A reported baseline (stand-in for the
    TF1.12 CPU path, which cannot run here), not the optimisation target."""
    from oracle import dmt_oracle as O
    from oracle import dmt_oracle_torch as OT
    from cikm2020_dmt_amd.data_feed.synthetic import make_batch
    import torch as th
    so = dict(sp)
    # torch's intra-op pool scales badly past a few dozen threads on the many small ops of this model (measured on this host class:
    # 256 threads are ~40x SLOWER than 8), so the port uses at most --cpu-threads (32) of the host's cores and reports both numbers.
    host = os.cpu_count() or 1
    cores = min(host, args.cpu_threads)
    th.set_num_threads(cores)
    P = O.init_params(so, seed=1, dtype=np.float32)
    trainer = OT.TorchTrainer(P, so, dtype=th.float32)
    del P
    inputs, mask, _l = make_batch(sp, args.cpu_batch, seed=7, lengths="full", law=args.law)
    budget = float(args.cpu_budget)
    t_all = time.perf_counter()
    warm = []
    for _ in range(args.cpu_warmup):
        t0 = time.perf_counter(); trainer.step(inputs, mask); warm.append(time.perf_counter() - t0)
        if time.perf_counter() - t_all > 0.35 * budget:
            break
    per = min(warm)
    steps = max(1, min(args.cpu_steps, int((0.85 * budget - (time.perf_counter() - t_all)) / max(per, 1e-3))))
    times = []
    for _ in range(steps):
        t0 = time.perf_counter(); trainer.step(inputs, mask); times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    big = None
    if args.cpu_big_batch and time.perf_counter() - t_all < budget:
        inputs, mask, _l = make_batch(sp, args.cpu_big_batch, seed=8, lengths="full", law=args.law)
        t0 = time.perf_counter(); trainer.step(inputs, mask); big = time.perf_counter() - t0
    out = {"value": round(args.cpu_batch / med, 1), "unit": "samples/s", "cores": cores, "host_cores": host, "kind": "port",
           "sample": "batch %d: %d warm-up + %d timed train steps, median %.3f s/step, on %d of the host's %d cores (same model / dims / ids; fp32 "
                     "torch-CPU restatement of the reference step incl. its dense TF-Adam sweep over all 5.4M table rows)"
                     % (args.cpu_batch, len(warm), steps, med, cores, host),
           "protocol": ("BASELINE.md §3 (5 warm-up + 20 timed at B=256; one B=4096 step), bounded to --cpu-budget %.0f s of CPU work.  DEVIATION from §3's "
                        "'all host cores': %d of %d cores (--cpu-threads) -- torch's intra-op pool is ~40x SLOWER at 256 threads than at 8-32 on this "
                        "model's many small ops (measured on this host class), so all cores would understate the CPU path" % (budget, cores, host))}
    if big is not None:
        out["value_b%d" % args.cpu_big_batch] = round(args.cpu_big_batch / big, 1)
        out["sample"] += "; batch %d: 1 step, %.2f s" % (args.cpu_big_batch, big)
    return out


if __name__ == "__main__":
    main()
