#!/usr/bin/env python
"""Generates tests/golden/offline_metrics_golden.json IN THIS CONTAINER by running the reference's own
DMT_code/metrics/metrics.py (pandas + sklearn + multiprocessing; importable here) on seeded synthetic evaluation rows.

Rows: header = 13 tab-separated fields in the order of dmt.conf:87 (`header_schema`), scores in (0, 1) with deliberate ties.
Every `uuid` / `sid` group of more than one row contains both classes for the AUC case "mixed" (independent of how the installed
scikit-learn treats single-class groups); the case "single_class" pins the reference's `except: return 1` rule only through its
own code path: scikit-learn 1.7 returns NaN there instead of raising, which the script records as `sklearn_nan_groups`."""
import json
import os
import sys
import warnings

import numpy as np

REF = "/root/reference/DMT_code"
sys.path.insert(0, os.path.join(REF, "metrics"))
import metrics as ref   # noqa: E402  (the reference module)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "offline_metrics_golden.json")
SCHEMA = ["label", "uuid", "sid", "sku", "pos", "page", "ts", "f7", "f8", "f9", "f10", "f11", "f12"]


def make_rows(seed, n_users, mixed):
    rng = np.random.default_rng(seed)
    headers, scores = [], []
    for u in range(n_users):
        for s in range(int(rng.integers(1, 4))):
            n = int(rng.integers(1, 20))
            labels = rng.choice([0, 1, 2, 4, 5], size=n, p=[0.6, 0.05, 0.2, 0.05, 0.1])
            if mixed and n > 1:
                labels[0], labels[1 % n] = 5, 0          # both classes for both actions in every multi-row group
            for i in range(n):
                sc = float(np.round(rng.random(), 2))    # two decimals -> ties
                headers.append(("\t".join([str(int(labels[i])), "u%d" % u, "u%d_s%d" % (u, s), "sku%d" % rng.integers(0, 999), str(i), "0", "0"] + ["x"] * 6)).encode())
                scores.append(sc)
    return headers, scores


def run(headers, scores):
    res, at = ref.get_offline_metrics(SCHEMA, headers, scores)
    out = {"at_list": list(at), "pre_clk": list(map(float, res[ref.CLICK][0])), "mrr_clk": list(map(float, res[ref.CLICK][1])),
           "pre_ord": list(map(float, res[ref.ORDER][0])), "mrr_ord": list(map(float, res[ref.ORDER][1]))}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for gm in ("uuid", "sid"):
            a = ref.get_offline_metrics_auc(SCHEMA, headers, scores, group_method=gm)
            out["auc_clk_" + gm] = float(a[ref.CLICK][0])
            out["auc_ord_" + gm] = float(a[ref.ORDER][0])
    return out


if __name__ == "__main__":
    cases = {}
    for name, seed, users, mixed in (("mixed", 1, 40, True), ("mixed_small", 2, 3, True), ("single_class", 3, 25, False)):
        h, s = make_rows(seed, users, mixed)
        cases[name] = {"headers": [x.decode() for x in h], "scores": s, "expected": run(h, s)}
    json.dump({"schema": SCHEMA, "cases": cases}, open(OUT, "w"))
    for k, v in cases.items():
        print(k, len(v["scores"]), "rows", {a: (round(b, 6) if isinstance(b, float) else b) for a, b in v["expected"].items() if a.startswith("auc")})
