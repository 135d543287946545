"""Host-side (numpy mode) API of pgl_amd.Graph against the REFERENCE's own Graph, run side by side in the build
container (the reference's Python package imports on the oracle's paddle stand-in, oracle/ref_python.py).  Covers
what sits either side of the hot path: index arrays, neighbour queries, batching, and the on-disk format (a graph
dumped by one implementation must load in the other).  Skipped where /root/reference does not exist (GPU box).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/pgl"), reason="reference tree not present (GPU box)")

SCRIPT = r'''
import os, sys, tempfile, json
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ref_python
ref = ref_python.load()
assert ref is not None
sys.path.insert(0, %(root)r)
import importlib
mine = importlib.import_module("pgl_amd")

def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape and np.array_equal(a, b), what

rng = np.random.default_rng(3)
n, e = 60, 400
edges = rng.integers(0, n, (e, 2)).astype(np.int64)
edges[:40, 1] = 7
nf = {"h": rng.standard_normal((n, 5)).astype(np.float32)}
ef = {"w": rng.standard_normal((e, 2)).astype(np.float32)}
gr = ref.Graph(edges=edges, num_nodes=n, node_feat=nf, edge_feat=ef)
gm = mine.Graph(edges=edges, num_nodes=n, node_feat=nf, edge_feat=ef)

assert gr.num_nodes == gm.num_nodes and gr.num_edges == gm.num_edges and gr.num_graph == gm.num_graph
same(gr.nodes, gm.nodes, "nodes"); same(gr.edges, gm.edges, "edges")
same(gr.indegree(), gm.indegree(), "indegree"); same(gr.outdegree(), gm.outdegree(), "outdegree")
q = np.array([7, 0, 59, 7, 13])
same(gr.indegree(q), gm.indegree(q), "indegree(nodes)"); same(gr.outdegree(q), gm.outdegree(q), "outdegree(nodes)")
for by in ("src", "dst"):
    for a, b in zip(gr.sorted_edges(by), gm.sorted_edges(by)):
        same(a, b, "sorted_edges " + by)
for name in ("adj_src_index", "adj_dst_index"):
    ir, im = getattr(gr, name), getattr(gm, name)
    for f in ("_degree", "_sorted_v", "_sorted_u", "_sorted_eid", "_indptr"):
        same(getattr(ir, f), getattr(im, f), name + f)
for fn in ("successor", "predecessor"):
    ra, rb = getattr(gr, fn)(q, return_eids=True); ma, mb = getattr(gm, fn)(q, return_eids=True)
    for x, y in zip(ra, ma): same(x, y, fn)
    for x, y in zip(rb, mb): same(x, y, fn + " eids")
    for x, y in zip(getattr(gr, fn)(), getattr(gm, fn)()): same(x, y, fn + " all")
for a, b in zip(gr.node_batch_iter(16, shuffle=False), gm.node_batch_iter(16, shuffle=False)):
    same(a, b, "node_batch_iter")

# batching
parts_r, parts_m = [], []
for k, m in enumerate([5, 1, 9, 20]):
    ee = rng.integers(0, m, (3 * m, 2)).astype(np.int64)
    ff = {"x": rng.standard_normal((m, 3)).astype(np.float32)}
    parts_r.append(ref.Graph(edges=ee, num_nodes=m, node_feat=ff)); parts_m.append(mine.Graph(edges=ee, num_nodes=m, node_feat=ff))
for merged in (False, True):
    br = ref.Graph.disjoint(parts_r, merged_graph_index=merged); bm = mine.Graph.disjoint(parts_m, merged_graph_index=merged)
    assert br.num_graph == bm.num_graph and br.num_nodes == bm.num_nodes
    same(br.edges, bm.edges, "disjoint edges"); same(br.graph_node_id, bm.graph_node_id, "graph_node_id")
    same(br.graph_edge_id, bm.graph_edge_id, "graph_edge_id"); same(br.node_feat["x"], bm.node_feat["x"], "disjoint feat")
same(ref.Graph.batch(parts_r).graph_node_id, mine.Graph.batch(parts_m).graph_node_id, "batch")
br = ref.Graph.disjoint(parts_r); bm = mine.Graph.disjoint(parts_m)

# on-disk format, both directions (with and without built indices)
with tempfile.TemporaryDirectory() as td:
    gm.dump(os.path.join(td, "m")); gr.dump(os.path.join(td, "r"))
    assert sorted(os.listdir(os.path.join(td, "m"))) == sorted(os.listdir(os.path.join(td, "r")))
    for sub in ("adj_src", "adj_dst", "node_feat", "edge_feat"):
        assert sorted(os.listdir(os.path.join(td, "m", sub))) == sorted(os.listdir(os.path.join(td, "r", sub))), sub
    a = ref.Graph.load(os.path.join(td, "m")); b = mine.Graph.load(os.path.join(td, "r"))
    same(a.edges, edges, "ref loads mine"); same(b.edges, edges, "mine loads ref")
    same(a.indegree(), gr.indegree(), "ref loads mine: indegree"); same(b.outdegree(), gr.outdegree(), "mine loads ref: outdegree")
    same(a.node_feat["h"], nf["h"], "feat"); same(b.edge_feat["w"], ef["w"], "feat")
    for x, y in zip(a.sorted_edges("dst"), b.sorted_edges("dst")): same(x, y, "loaded sorted_edges")
    mm = gm.to_mmap(os.path.join(td, "mm")); same(mm.edges, edges, "to_mmap")
    # a batched graph keeps its per-graph index through the round trip, in both directions
    bm.dump(os.path.join(td, "bm")); br.dump(os.path.join(td, "br"))
    x = ref.Graph.load(os.path.join(td, "bm")); y = mine.Graph.load(os.path.join(td, "br"))
    assert x.num_graph == y.num_graph == br.num_graph
    same(x.graph_node_id, br.graph_node_id, "ref loads my batch"); same(y.graph_node_id, br.graph_node_id, "mine loads ref batch")
    same(y.graph_edge_id, br.graph_edge_id, "mine loads ref batch (edges)")
# graph transforms (pgl/utils/transform.py)
from pgl.utils.transform import to_undirected as r_und, add_self_loops as r_loops
from pgl_amd.utils.transform import to_undirected as m_und, add_self_loops as m_loops
for rf, mf in ((r_und, m_und), (r_loops, m_loops)):
    a, b = rf(gr), mf(gm)
    same(a.edges, b.edges, rf.__name__); assert a.num_nodes == b.num_nodes
    same(a.node_feat["h"], b.node_feat["h"], rf.__name__ + " feat")
print("HOST_API_OK")
'''


def test_numpy_mode_graph_matches_reference_graph():
    # a subprocess: the stand-in monkey-patches torch.Tensor (shape, +=, ...) and must not leak into this test session
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "HOST_API_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


BI_SCRIPT = r'''
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ref_python
ref = ref_python.load()
sys.path.insert(0, %(root)r)
import pgl_amd as mine

def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == object:
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), what
    else:
        assert a.shape == b.shape and np.array_equal(a, b), what

def mk(mod, ns, nd, e, seed):
    r = np.random.default_rng(seed)
    edges = np.stack([r.integers(0, ns, e), r.integers(0, nd, e)], 1).astype(np.int64)
    return mod.BiGraph(edges, src_num_nodes=ns, dst_num_nodes=nd, src_node_feat={"a": r.standard_normal((ns, 3)).astype(np.float32)},
                       dst_node_feat={"b": r.standard_normal((nd, 2)).astype(np.float32)},
                       edge_feat={"w": r.standard_normal((e, 1)).astype(np.float32)})

gr, gm = mk(ref, 30, 45, 300, 1), mk(mine, 30, 45, 300, 1)
q = np.array([3, 0, 7, 3])
for f in ("indegree", "outdegree"):
    same(getattr(gr, f)(), getattr(gm, f)(), f); same(getattr(gr, f)(q), getattr(gm, f)(q), f + "(nodes)")
for f in ("successor", "predecessor"):
    (ra, rb), (ma, mb) = getattr(gr, f)(q, return_eids=True), getattr(gm, f)(q, return_eids=True)
    same(ra, ma, f); same(rb, mb, f + " eids")
for by in ("src", "dst"):
    for x, y in zip(gr.sorted_edges(by), gm.sorted_edges(by)): same(x, y, "sorted_edges " + by)
same(gr.src_nodes, gm.src_nodes, "src_nodes"); same(gr.dst_nodes, gm.dst_nodes, "dst_nodes")
lr = [gr, mk(ref, 4, 9, 20, 2), mk(ref, 11, 3, 0, 3)]; lm = [gm, mk(mine, 4, 9, 20, 2), mk(mine, 11, 3, 0, 3)]
for merged in (False, True):
    br, bm = ref.BiGraph.disjoint(lr, merged), mine.BiGraph.disjoint(lm, merged)
    for f in ("edges", "graph_src_node_id", "graph_dst_node_id", "graph_edge_id", "num_graph", "src_num_nodes", "dst_num_nodes"):
        same(getattr(br, f), getattr(bm, f), "disjoint(%%s) %%s" %% (merged, f))
    same(br.src_node_feat["a"], bm.src_node_feat["a"], "src feat"); same(br.dst_node_feat["b"], bm.dst_node_feat["b"], "dst feat")
    same(br.edge_feat["w"], bm.edge_feat["w"], "edge feat")
br, bm = ref.BiGraph.batch(lr), mine.BiGraph.batch(lm)
br.indegree(); bm.indegree()
with tempfile.TemporaryDirectory() as td:
    br.dump(td + "/r"); bm.dump(td + "/m")
    assert sorted(os.listdir(td + "/r")) == sorted(os.listdir(td + "/m"))
    x, y = ref.BiGraph.load(td + "/m"), mine.BiGraph.load(td + "/r")
    same(x.edges, br.edges, "ref loads mine"); same(y.edges, br.edges, "mine loads ref")
    same(x.graph_dst_node_id, br.graph_dst_node_id, "ids"); same(y.graph_src_node_id, br.graph_src_node_id, "ids")
    same(y.indegree(), br.indegree(), "indegree after load"); same(y.dst_node_feat["b"], br.dst_node_feat["b"], "feat after load")
print("BIGRAPH_HOST_API_OK")
'''


def test_numpy_mode_bigraph_matches_reference_bigraph():
    r = subprocess.run([sys.executable, "-c", BI_SCRIPT % {"root": ROOT}], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "BIGRAPH_HOST_API_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


HET_SCRIPT = r'''
import os, sys, tempfile
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import ref_python
ref = ref_python.load()
sys.path.insert(0, %(root)r)
import pgl_amd as mine

def same(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    if a.dtype == object:
        assert len(a) == len(b) and all(np.array_equal(x, y) for x, y in zip(a, b)), what
    else:
        assert a.shape == b.shape and np.array_equal(a, b), what

rng = np.random.default_rng(9)
n = 40
node_types = [(i, "user" if i %% 3 else "item") for i in range(n)]
edges = {"click": [tuple(e) for e in rng.integers(0, n, (120, 2)).tolist()], "buy": [tuple(e) for e in rng.integers(0, n, (30, 2)).tolist()]}
nf = {"h": rng.standard_normal((n, 4)).astype(np.float32)}
ef = {"click": {"w": rng.standard_normal((120, 1)).astype(np.float32)}, "buy": {"w": rng.standard_normal((30, 1)).astype(np.float32)}}
hr = ref.HeterGraph(edges=edges, node_types=node_types, node_feat=nf, edge_feat=ef)
hm = mine.HeterGraph(edges=edges, node_types=node_types, node_feat=nf, edge_feat=ef)
assert hr.num_nodes == hm.num_nodes and hr.num_edges == hm.num_edges and list(hr.edge_types) == list(hm.edge_types)
assert hr.num_nodes_by_type("user") == hm.num_nodes_by_type("user")
same(hr.node_types, hm.node_types, "node_types"); same(hr.nodes, hm.nodes, "nodes"); same(hr.node_feat["h"], hm.node_feat["h"], "feat")
q = np.array([1, 5, 5, 39])
for f in ("indegree", "outdegree"):
    same(getattr(hr, f)(), getattr(hm, f)(), f); same(getattr(hr, f)(q), getattr(hm, f)(q), f + "(q)")
    same(getattr(hr, f)(q, "buy"), getattr(hm, f)(q, "buy"), f + "(q, buy)")
for f in ("successor", "predecessor"):
    (ra, rb), (ma, mb) = getattr(hr, f)("click", q, True), getattr(hm, f)("click", q, True)
    same(ra, ma, f); same(rb, mb, f + " eids")
for a, b in zip(hr.node_batch_iter(7, n_type="item"), hm.node_batch_iter(7, n_type="item")): same(a, b, "node_batch_iter")
same(hr["buy"].edges, hm["buy"].edges, "per-type graph"); same(hr.edge_feat["click"]["w"], hm.edge_feat["click"]["w"], "edge feat")
with tempfile.TemporaryDirectory() as td:
    hr.dump(td + "/r", indegree=True, outdegree=True); hm.dump(td + "/m", indegree=True, outdegree=True)
    assert sorted(os.listdir(td + "/r")) == sorted(os.listdir(td + "/m"))
    x, y = ref.HeterGraph.load(td + "/m"), mine.HeterGraph.load(td + "/r")
    same(x["click"].edges, hr["click"].edges, "ref loads mine"); same(y["buy"].edges, hr["buy"].edges, "mine loads ref")
    same(y.indegree(q), hr.indegree(q), "indegree after load"); same(x.outdegree(q, "click"), hr.outdegree(q, "click"), "outdegree after load")
    assert list(y.edge_types) == list(hr.edge_types) and y.num_nodes_by_type("item") == hr.num_nodes_by_type("item")
print("HETER_HOST_API_OK")
'''


def test_numpy_mode_hetergraph_matches_reference_hetergraph():
    r = subprocess.run([sys.executable, "-c", HET_SCRIPT % {"root": ROOT}], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0 and "HETER_HOST_API_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
