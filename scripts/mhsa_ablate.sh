# Phase ablation of mhsa_fwd_kernel (library built with `make EXPERIMENTS=1`): DMT_MHSA_DEBUG bits switch parts off, DMT_MHSA_SHAPE picks
# the workgroup shape (0: 8 compute + 4 loaders, one per CU; 1: 4 + 1, two per CU; 2: 4 + 2); timing only.
for sh in ${SHAPES:-0}; do for d in ${DBGS:-0 47 63 111 175 303 239 495 511}; do
  echo "== SHAPE=$sh DEBUG=$d"; DMT_MHSA_SHAPE=$sh DMT_MHSA_DEBUG=$d timeout 120 python scripts/mhsa_micro.py 4096 50 2>&1 | grep "infer"
done; done
