#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REFERENCE's own compiled native module
(pgl/graph_kernel.pyx built in place by oracle/build_ref.py).  Run in the build container, where
/root/reference exists; the fixtures travel to the GPU box, the reference does not.

    python tests/golden/make_golden.py

Each fixture = seeded inputs + the reference's outputs:
  build_index_*.npz   edges, num_nodes -> degree, sorted_v, sorted_u, sorted_eid, indptr for BOTH the
                      dst-keyed (adj_dst_index) and src-keyed (adj_src_index) calls
  map_ids.npz         map_nodes / map_edges
  metis_*.npz         METIS k-way partition of a planted-community graph (cut + part sizes only are
                      asserted against; ids are METIS-specific)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import ref_native  # noqa: E402

gk = ref_native.load()
assert gk is not None, "reference native module unavailable"


def rmat_like(n_log, e, seed):
    rng = np.random.default_rng(seed)
    src = np.zeros(e, np.int64); dst = np.zeros(e, np.int64)
    for _ in range(n_log):
        r = rng.random(e)
        src = (src << 1) | (r >= 0.76)
        dst = (dst << 1) | (((r >= 0.57) & (r < 0.76)) | (r >= 0.95))
    perm = rng.permutation(1 << n_log)
    return np.stack([perm[src], perm[dst]], 1).astype(np.int64)


cases = {
    "empty": (np.zeros((0, 2), np.int64), 6),
    "single": (np.array([[2, 2]], np.int64), 4),
    "tiny": (np.array([(0, 1), (1, 2), (3, 4), (4, 1), (1, 0)], np.int64), 5),
    "uniform": (np.random.default_rng(1).integers(0, 500, (6000, 2)).astype(np.int64), 500),
    "rmat12": (rmat_like(12, 60000, 2), 1 << 12),
    "hub": (np.concatenate([np.random.default_rng(3).integers(0, 300, (3000, 2)),
                            np.stack([np.random.default_rng(4).integers(0, 300, 5000), np.full(5000, 7)], 1)]).astype(np.int64), 300),
}
for name, (edges, n) in cases.items():
    out = {"edges": edges, "num_nodes": np.int64(n)}
    for tag, (u, v) in (("dst", (edges[:, 1], edges[:, 0])), ("src", (edges[:, 0], edges[:, 1]))):
        deg, sv, su, se, ip = gk.build_index(np.ascontiguousarray(u), np.ascontiguousarray(v), n)
        out.update({tag + "_degree": deg, tag + "_sorted_v": sv, tag + "_sorted_u": su, tag + "_sorted_eid": se,
                    tag + "_indptr": ip})
    np.savez_compressed(os.path.join(HERE, "build_index_%s.npz" % name), **out)

reindex = {int(k): int(v) for v, k in enumerate(np.random.default_rng(5).permutation(1000)[:200])}
nodes = np.random.default_rng(6).choice(list(reindex.keys()), 500).astype(np.int64)
edges = np.random.default_rng(7).choice(list(reindex.keys()), (300, 2)).astype(np.int64)
np.savez_compressed(os.path.join(HERE, "map_ids.npz"), keys=np.array(list(reindex.keys()), np.int64),
                    vals=np.array(list(reindex.values()), np.int64), nodes=nodes, mapped_nodes=gk.map_nodes(nodes, dict(reindex)),
                    edges=edges, mapped_edges=gk.map_edges(np.arange(300, dtype=np.int64), edges, dict(reindex)))

rng = np.random.default_rng(8)
n, comm = 3000, 12
a = rng.integers(0, n, 30000)
b = np.where(rng.random(30000) < 0.9, (a // (n // comm)) * (n // comm) + rng.integers(0, n // comm, 30000), rng.integers(0, n, 30000))
keep = a != b
und = np.unique(np.stack([np.minimum(a[keep], b[keep]), np.maximum(a[keep], b[keep])], 1), axis=0)
sym = np.concatenate([und, und[:, ::-1]], 0).astype(np.int64)
deg, sv, su, se, ip = gk.build_index(np.ascontiguousarray(sym[:, 1]), np.ascontiguousarray(sym[:, 0]), n)
for k in (2, 4, 8):
    part = gk.metis_partition(n, ip, sv, k, None, None, False)
    np.savez_compressed(os.path.join(HERE, "metis_k%d.npz" % k), edges=sym, num_nodes=np.int64(n), nparts=np.int64(k), part=part,
                        cut=np.int64((part[sym[:, 0]] != part[sym[:, 1]]).sum()), sizes=np.bincount(part, minlength=k))
print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".npz")))
