"""dmt_proj at M = 204800 with its phases switched off one at a time (DMT_PROJ_DEBUG, results garbage): where does the time go?"""
import os, sys, torch
sys.path.insert(0, ".")
from cikm2020_dmt_amd import ops
dev = torch.device("cuda:0")
M, kin, n = 204800, 320, 960
w = torch.randn(kin, n, device=dev) * 0.05
b = torch.randn(n, device=dev)
x = torch.randn(M, kin, device=dev).to(torch.bfloat16)
img = torch.empty(ops.proj_image_bytes(kin, n), dtype=torch.uint8, device=dev)
ops.proj_image_build(w, b, img)
W = ops.Weight(w); W.proj = img
grids = [int(g) for g in os.environ.get("PROJ_GRIDS", "512").split(",")]
for grid, dbg in [(g, d) for g in grids for d in (0, 64, 15, 79)] if len(grids) > 1 else [(grids[0], d) for d in (0, 1, 2, 4, 5, 7, 8, 15)]:
    os.environ["DMT_PROJ_GRID"] = str(grid)
    os.environ["DMT_PROJ_DEBUG"] = str(dbg)
    for _ in range(3):
        ops.proj_forward(x, W, n)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(20):
        ops.proj_forward(x, W, n)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print("grid %d " % grid + "dbg %2d (%s): %.1f us  %.0f TFLOP/s" % (dbg, ",".join(nm for bit, nm in ((1, "no DMA"), (2, "no stores"), (4, "no LDS reads"), (8, "no MFMA"), (32, "one slice per tile"), (64, "no barriers")) if dbg & bit) or "full", us, 2.0 * M * kin * n / us / 1e6))
