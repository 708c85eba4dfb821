#!/bin/bash
# 1 / 2 / 4 / 8-GPU weak-scaling sweep of bench.py on ONE node (the driver's own launch line), with the per-phase spans of every N:
#   scripts/scale_sweep.sh [steps] [warmup] [extra bench.py flags...]      -> gpurun_out/scale/bench_n<N>.json, gpurun_out/scale/sweep.jsonl
#   (one JSON line per N: ms/step, samples/s, speed-up, step_phases_ms) + a summary table
# Needs as many visible GPUs as the largest N (the round's GPU boxes have one: the driver runs the real sweep at round end).
#
#   scripts/scale_sweep.sh --dry-run     the SAME launch line, environment and output schema for N = 2 and 8 ranks on ONE device (every rank on
#   cuda:0, gloo instead of RCCL, batch 256, 3 steps): what a one-GPU box can check of the 8-GPU run before a node is available -- the
#   rendezvous, the rank / LOCAL_RANK plumbing, the per-rank CPU plan, the data-parallel step end to end, and that rank 0's line carries
#   every key the driver and DESIGN.md section 6 read (value, ms_per_step, n_gpus, scaling, step_phases_ms with its collectives' spans).
R=${GRAFT_REPO_ROOT:-$(pwd)}
if [ "$1" = "--dry-run" ]; then
  O=$R/gpurun_out/scale_dry; mkdir -p $O
  rc=0
  for N in 2 8; do
    DMT_BENCH_ONE_DEVICE=1 DMT_DIST_BACKEND=gloo HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N \
      --master-addr 127.0.0.1 --master-port $((29700 + N)) $R/bench.py --gpus $N --steps 3 --warmup 1 --batch 256 --fresh-batches 2 --age-tables 0 \
      --no-cpu-baseline --record-files 0 2>$O/bench_n$N.err | grep '^{' > $O/bench_n$N.json
    python - <<PY || rc=1
import json, sys
f = "$O/bench_n$N.json"
try:
    d = json.load(open(f))
except Exception as e:
    print("dry run N=$N: no JSON line (%s); see $O/bench_n$N.err" % e); sys.exit(1)
need = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "step_phases_ms"]
missing = [k for k in need if k not in d]
ph = d.get("step_phases_ms", {})
ok = (not missing and d["n_gpus"] == $N and d["scaling"] == "weak" and ph.get("world_size_seen_by_torch_distributed") == $N and ph.get("data_parallel_path") is True
      and d["config"]["global_batch"] == 256 * $N)
print("dry run N=$N: %s  %.2f ms/step  phases: %s  cpu plan: %s" % ("ok" if ok else "FAILED (missing %s)" % missing, d["ms_per_step"],
      {k: v for k, v in ph.items() if k.endswith("_ms")}, d.get("cpu_affinity_rank0")))
sys.exit(0 if ok else 1)
PY
  done
  exit $rc
fi
steps=${1:-50}; warm=${2:-10}; shift 2 2>/dev/null
O=$R/gpurun_out/scale; mkdir -p $O
ngpu=$(python -c "import torch; print(torch.cuda.device_count())")
for N in 1 2 4 8; do
  [ "$N" -gt "$ngpu" ] && { echo "skip N=$N: $ngpu GPU(s) visible"; continue; }
  if [ "$N" = 1 ]; then
    python $R/bench.py --gpus 1 --steps $steps --warmup $warm --no-cpu-baseline --record-files 0 "$@" 2>$O/bench_n$N.err | grep '^{' > $O/bench_n$N.json
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29600 + N)) \
      $R/bench.py --gpus $N --steps $steps --warmup $warm --no-cpu-baseline "$@" 2>$O/bench_n$N.err | grep '^{' > $O/bench_n$N.json
  fi
done
python - <<PY
import json, glob, os
base = None
for N in (1, 2, 4, 8):
    f = "$O/bench_n%d.json" % N
    if not os.path.exists(f) or not os.path.getsize(f):
        continue
    d = json.load(open(f))
    base = base or d["value"]
    ph = {k: v for k, v in d.get("step_phases_ms", {}).items() if k.endswith("_ms")}
    print("N=%d  %.3f ms/step  %.0f samples/s  x%.2f of N=1  world=%s  phases(ms): %s"
          % (N, d["ms_per_step"], d["value"], d["value"] / base, d.get("step_phases_ms", {}).get("world_size_seen_by_torch_distributed"), ph))
    # one JSON line per N (the whole sweep as ONE artefact: gpurun_out/scale/sweep.jsonl)
    line = {"n_gpus": N, "ms_per_step": d["ms_per_step"], "value": d["value"], "unit": d["unit"], "speedup_vs_n1": round(d["value"] / base, 3),
            "step_phases_ms": d.get("step_phases_ms", {}), "config": d.get("config", {})}
    open("$O/sweep.jsonl", "a" if N > 1 else "w").write(json.dumps(line) + "\n")
PY
