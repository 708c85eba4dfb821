"""String id -> int index, mirroring /root/reference/DMT_code/data_feed/index_tables.py:8-45 (LookupTables).

Each embedding name has a vocabulary list `ID_TABLES[name]` (first entry 'unknow'); an id maps to its position in the
list, an out-of-vocabulary id to len(vocab) + Fingerprint64(id) % (id_size - len(vocab)) and -- when the table has
no OOV buckets (Time*: 23 == id_size) -- to default 0.
The vocabulary data files (conf/idtables/*.py, 3.7 MB of the reference's data) are NOT shipped here: pass the
directory that holds them (`idtables_dir`), or a dict name -> list.  A missing vocabulary (the reference's own
Sku.py is absent) degrades to ['unknow'] only: every id is hashed.
"""
from __future__ import annotations

import os
import runpy
from typing import Dict, List, Optional

import numpy as np

from ..sparse import SparseTensorValue
from .farmhash import fingerprint64


class LookupTables(object):
    def __init__(self, wnd_conf, idtables_dir: Optional[str] = None, id_tables: Optional[Dict[str, List[str]]] = None):
        self.wnd_conf = wnd_conf
        emb = list(wnd_conf.embedding_list) + list(wnd_conf.embedding_list_bias)
        self.embLookupTables: Dict[str, dict] = {}
        for e in emb:
            name, id_size = e[0], int(e[1])
            if name in self.embLookupTables:
                continue
            vocab = None
            if id_tables is not None and name in id_tables:
                vocab = list(id_tables[name])
            elif idtables_dir is not None and os.path.exists(os.path.join(idtables_dir, name + ".py")):
                vocab = list(runpy.run_path(os.path.join(idtables_dir, name + ".py"))["ID_TABLES"][name])
            if vocab is None:
                vocab = ["unknow"]
            index = {}
            for i, s in enumerate(vocab):
                index.setdefault(s.encode("utf-8") if isinstance(s, str) else bytes(s), i)
            self.embLookupTables[name] = dict(index=index, vocab_size=len(vocab), buckets=id_size - len(vocab))
        self.featureLookupTables = {}
        for e in emb:
            self.featureLookupTables.setdefault(e[3], self.embLookupTables[e[0]])

    @staticmethod
    def _lookup(table: dict, ids) -> np.ndarray:
        out = np.empty(len(ids), dtype=np.int64)
        index, vs, nb = table["index"], table["vocab_size"], table["buckets"]
        for i, s in enumerate(ids):
            b = s if isinstance(s, (bytes, bytearray)) else str(s).encode("utf-8")
            j = index.get(bytes(b))
            if j is None:
                j = vs + fingerprint64(bytes(b)) % nb if nb > 0 else 0     # default_value=0
            out[i] = j
        return out

    def transform_id2index(self, features: dict):
        """In place, like the reference: every id feature's string values become int64 indices."""
        for key in list(features.keys()):
            if key in self.featureLookupTables:
                raw = features[key]
                features[key] = SparseTensorValue(raw.indices, self._lookup(self.featureLookupTables[key], raw.values), raw.dense_shape)

    def inf_transform(self, id_name, ids):
        return self._lookup(self.featureLookupTables[id_name], ids)

    def lookup_embedding(self, emb_name, ids):
        return self._lookup(self.embLookupTables[emb_name], ids)
