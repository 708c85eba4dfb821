"""Aggregate FETCH_SIZE / WRITE_SIZE (rocprofv3 --pmc, ONE counter per pass -- the two do not fit one pass and combining TCC counters
hung nodes on this pool) over the kernel families of a bench.py run.
usage: pmc_traffic.py <fetch_dir> <write_dir> [steps]  -> JSON on stdout: {family: bytes per launch}, with the gfx950 correction of
MI355X_MICROARCH.md (HBM section): FETCH_SIZE counts 64 B per 128-B request -> doubled; WRITE_SIZE as reported; both in KB."""
import csv, glob, json, sys

FAMILIES = {
    "chain2": ["chain2_kernel"],
    "gemm_bf16": ["gemm_glds_kernel", "gemm_dw_glds_kernel", "gemm_kernel<"],
    "wgrad320": ["wgrad320_kernel"],
    "proj": ["proj_kernel"],
    "attn": ["attn_fwd_co_kernel", "attn_bwd_co_kernel", "attn_q1v_kernel"],
    "attn_long": ["attn_long_fwd", "attn_long_bwd", "attn_q1_long"],
    "q1mem": ["q1m_fwd_kernel", "q1m_bwd_kernel"],
    "mmoe_experts": ["mmoe_experts_", "mmoe_split_", "mmoe_mix_finish", "mmoe_dgate_finish"],
    "heads": ["heads_fwd_kernel", "heads_bwd_kernel"],
    "gather_fwd": ["gather_group_kernel"],
    "embgrad_reduce": ["embgrad_reduce_kernel"],
    "adam_sparse": ["adam_sparse_kernel"],
    "mhsa_block": ["mhsa_fwd_kernel"],
    "mhsa_bwd": ["mhsa_bwd_kernel"],
    "gemm_small": ["gemm_small_kernel"],
    "adam_catchup": ["adam_catchup_kernel"],
    "ln": ["ln_fwd", "ln_bwd"],
    "index_sort": ["rs_hist_kernel", "rs_scan_kernel", "rs_scatter_kernel", "sh_count_kernel", "sh_scan_kernel", "sh_write_kernel"],     # csrc/dmt_sort.hip (round 6)
    "embgrad_keys": ["embgrad_keys_kernel"],
}


def load(d, counter):
    f = glob.glob("%s/*/*counter_collection.csv" % d)[0]
    tot, n = {}, {}
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    # "__all__": every kernel from the first step's gather on (the set-up kernels of bench.py -- table init, ageing -- come before it)
    first = min((int(r["Dispatch_Id"]) for r in rows if "gather_group_kernel" in r["Kernel_Name"]), default=0)
    for r in rows:
        k = r["Kernel_Name"]
        if int(r["Dispatch_Id"]) >= first:
            tot["__all__"] = tot.get("__all__", 0.0) + float(r["Counter_Value"])
            n["__all__"] = n.get("__all__", 0) + 1
        for fam, pats in FAMILIES.items():
            if any(p in k for p in pats):
                tot[fam] = tot.get(fam, 0.0) + float(r["Counter_Value"])
                n[fam] = n.get(fam, 0) + 1
    return tot, n


fetch, n1 = load(sys.argv[1], "FETCH_SIZE")
write, n2 = load(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
out = {}
for fam in fetch:
    assert n1[fam] == n2.get(fam), (fam, n1[fam], n2.get(fam))
    fb, wb = fetch[fam] * 1024 * 2, write[fam] * 1024
    out[fam] = {"launches": n1[fam], "fetch_bytes_per_launch": fb / n1[fam], "write_bytes_per_launch": wb / n1[fam],
                "hbm_bytes_per_launch": (fb + wb) / n1[fam], "hbm_bytes_per_step": (fb + wb) / steps}
print(json.dumps(out))
