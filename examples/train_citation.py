#!/usr/bin/env python3
"""Two-layer GCN / GAT / GraphSAGE node classification on a synthetic citation-style graph, written
against the pgl_amd API exactly as the reference's examples/gcn|gat|graphsage train.py scripts are
written against pgl (build Graph with numpy -> indegree() on the numpy graph -> tensor() -> layers).

The reference's Cora files are incomplete in its checkout (pgl/data/cora/cora.content is missing) and
there is no network, so the data is a seeded stand-in with the same shape: N = 2708 nodes, 1433-d
row-normalised bag-of-words features, 7 classes, a planted-partition citation graph symmetrised with
self loops (what pgl/dataset.py:221-238 does to Cora).  This is BASELINE config 0: plumbing only.

    python examples/train_citation.py --model gcn --epochs 100
"""
import argparse
import os
import sys
import time

import numpy as np
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl  # noqa: E402


def synthetic_cora(seed=0, n=2708, d=1433, classes=7, avg_deg=4):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes, n)
    m = n * avg_deg // 2
    a = rng.integers(0, n, m)
    same = rng.random(m) < 0.8
    b = np.where(same, rng.permutation(n)[np.searchsorted(np.sort(y), y[a]) % n], rng.integers(0, n, m))
    # draw "same class" partners properly: pick a random node of the same class
    idx_by_class = [np.flatnonzero(y == c) for c in range(classes)]
    b = np.array([rng.choice(idx_by_class[y[u]]) if s else rng.integers(0, n) for u, s in zip(a, same)])
    und = {(int(u), int(v)) for u, v in zip(a, b)} | {(int(v), int(u)) for u, v in zip(a, b)} | {(i, i) for i in range(n)}
    edges = np.array(sorted(und), dtype=np.int64)
    proto = (rng.random((classes, d)) < 0.02).astype(np.float32)
    x = ((rng.random((n, d)) < 0.005) | (proto[y] * (rng.random((n, d)) < 0.5) > 0)).astype(np.float32)
    x = x / np.maximum(x.sum(1, keepdims=True), 1.0)
    perm = rng.permutation(n)
    return edges, x, y.astype(np.int64), perm[:140], perm[140:640], perm[640:1640]


class Net(torch.nn.Module):
    def __init__(self, model, d_in, hidden, classes):
        super().__init__()
        self.model = model
        if model == "gcn":
            self.l1 = pgl.nn.GCNConv(d_in, hidden, activation="relu")
            self.l2 = pgl.nn.GCNConv(hidden, classes)
        elif model == "gat":
            self.l1 = pgl.nn.GATConv(d_in, hidden // 8, feat_drop=0.6, attn_drop=0.6, num_heads=8, activation="elu")
            self.l2 = pgl.nn.GATConv(hidden, classes, feat_drop=0.6, attn_drop=0.6, num_heads=1, concat=False)
        else:
            self.l1 = pgl.nn.GraphSageConv(d_in, hidden, aggr_func="mean")
            self.l2 = pgl.nn.GraphSageConv(hidden, classes, aggr_func="mean", normalize=False)
        self.drop = torch.nn.Dropout(0.5)

    def forward(self, g, x):
        if self.model == "sage":
            return self.l2(g, self.drop(self.l1(g, x, act="relu")))
        return self.l2(g, self.drop(self.l1(g, self.drop(x) if self.model == "gcn" else x)))


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="gcn", choices=["gcn", "gat", "sage"])
    ap.add_argument("--epochs", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=64)
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args(argv)
    torch.manual_seed(0)
    edges, x, y, tr, va, te = synthetic_cora()
    graph = pgl.Graph(num_nodes=x.shape[0], edges=edges, node_feat={"words": x})
    graph.indegree()                                  # index built on the numpy graph, as examples/gcn/train.py:83 does
    graph.tensor()
    dev = graph.edges.device
    yt = torch.from_numpy(y).to(dev)
    tr, va, te = (torch.from_numpy(i).to(dev) for i in (tr, va, te))
    net = Net(args.model, x.shape[1], args.hidden, 7).to(dev)
    opt = torch.optim.Adam(net.parameters(), lr=args.lr, weight_decay=5e-4, capturable=torch.cuda.is_available())   # step counter on the device: no host bookkeeping per step
    hist = []
    for ep in range(args.epochs):
        t0 = time.time()
        net.train()
        loss = F.cross_entropy(net(graph, graph.node_feat["words"])[tr], yt[tr])
        opt.zero_grad(); loss.backward(); opt.step()
        net.eval()
        with torch.no_grad():
            pred = net(graph, graph.node_feat["words"]).argmax(1)
        acc = lambda idx: float((pred[idx] == yt[idx]).float().mean())
        hist.append((float(loss.detach()), acc(va), acc(te)))
        if ep % 20 == 0 or ep == args.epochs - 1:
            print("epoch %3d loss %.4f val %.3f test %.3f  (%.1f ms)" % (ep, hist[-1][0], hist[-1][1], hist[-1][2], (time.time() - t0) * 1e3))
    return hist


if __name__ == "__main__":
    main()
