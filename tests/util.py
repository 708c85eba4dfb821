"""Shared helpers for the parity tests (tests may import oracle/, the product may not)."""
import numpy as np
import torch

from oracle import dmt_oracle as O
from oracle import dmt_oracle_torch as OT
from cikm2020_dmt_amd import spec as S

SMALL_ROWS = {"Sku": 3000, "Brand": 800, "Shopid": 900, "Cid3": 300, "Cid2": 50}


def small_specs(ord_suffix="12m_50"):
    """(oracle spec, product spec) with small vocabularies; the two dicts must describe the same model."""
    so = O.scaled_spec(O.default_spec(ord_suffix), SMALL_ROWS)
    sp = S.scaled_spec(S.default_spec(ord_suffix), SMALL_ROWS)
    return so, sp


def sparse_to_dense_tables(store, sparse):
    """(uniq_keys, n_uniq, grad_rows, cap) -> {table tf_name: dense fp64 [rows, dim]} (host side, for comparison)."""
    uniq, n_uniq, grad_rows, _cap = sparse
    n = int(n_uniq.item())
    keys = uniq[:n].cpu().numpy().astype(np.int64)
    rows = grad_rows[:n].float().cpu().numpy().astype(np.float64)
    out = {}
    for name, (base, nrows) in store.table_rows.items():
        dim = store.tables[name].shape[1]
        g = np.zeros((nrows, dim))
        sel = (keys >= base) & (keys < base + nrows)
        g[keys[sel] - base] = rows[sel][:, :dim]
        out[name] = g
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))
