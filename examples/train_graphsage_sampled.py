#!/usr/bin/env python3
"""Mini-batch GraphSAGE with neighbour sampling, written against the pgl_amd API the way the reference's
examples/graphsage/cpu_sample_version/train.py is written against pgl:

  --sampler host : the reference's flow.  `pgl.sampling.graphsage_sample` on the numpy graph returns one relabelled
                   subgraph per batch (sample_index = its nodes in the full graph, index = the batch nodes inside it);
                   the model runs every layer on that subgraph and the loss is taken at `index`.
  --sampler gpu  : the same model on blocks sampled on the device (`pgl.sampling.NeighborSampler`, one small graph per
                   layer, outermost first); features never leave HBM.

reddit.npz / reddit_adj.npz are not in the reference checkout and there is no network: the data is a seeded stand-in
with the same structure (planted classes, 602-d standardised features, 41 classes) at a size that trains in seconds.

    python examples/train_graphsage_sampled.py --sampler gpu --epochs 3
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


def synthetic_reddit(n=20000, d=602, classes=41, avg_deg=25, seed=0):
    rng = np.random.default_rng(seed)
    y = rng.integers(0, classes, n)
    members = [np.flatnonzero(y == c) for c in range(classes)]
    m = n * avg_deg // 2
    a = rng.integers(0, n, m)
    same = rng.random(m) < 0.7
    pick = rng.random(m)
    b = np.where(same, np.array([members[c][int(p * len(members[c]))] for c, p in zip(y[a], pick)]), rng.integers(0, n, m))
    edges = np.concatenate([np.stack([a, b], 1), np.stack([b, a], 1)]).astype(np.int64)        # symmetric, as --symmetry
    centers = rng.standard_normal((classes, d)).astype(np.float32)
    x = centers[y] * 0.5 + rng.standard_normal((n, d)).astype(np.float32)
    x = (x - x.mean(0)) / x.std(0)                                                               # StandardScaler, as --normalize
    perm = rng.permutation(n)
    return edges, x.astype(np.float32), y.astype(np.int64), perm[: n // 2], perm[n // 2: n // 2 + n // 8]


class GraphSage(torch.nn.Module):
    """examples/graphsage/cpu_sample_version/model.py: num_layers GraphSageConv (mean) + a Linear classifier."""

    def __init__(self, input_size, num_class, num_layers=2, hidden_size=128, drop=0.5):
        super().__init__()
        self.convs = torch.nn.ModuleList(
            [pgl.nn.GraphSageConv(input_size if i == 0 else hidden_size, hidden_size, "mean") for i in range(num_layers)])
        self.linear = torch.nn.Linear(hidden_size, num_class)
        self.dropout = torch.nn.Dropout(drop)

    def forward(self, graphs, feature):
        """graphs: ONE graph used by every layer (host flow) or a list of (block, n_dst), outermost first (device flow)."""
        if not isinstance(graphs, (list, tuple)):
            graphs = [(graphs, None)] * len(self.convs)
        for conv, (g, n_dst) in zip(self.convs, graphs):
            feature = conv(g, feature if n_dst is None else (feature, feature[:n_dst]))
            feature = self.dropout(feature)
        return self.linear(feature)


def batches(index, size, rng=None):
    index = index if rng is None else rng.permutation(index)
    for i in range(0, len(index), size):
        yield index[i:i + size]


def run_epoch(args, model, optim, graph_np, graph_dev, sampler, feature, labels, index, rng, train):
    model.train(train)
    tot_loss = tot_acc = tot = 0
    for nodes in batches(index, args.batch_size, rng if train else None):
        if args.sampler == "host":
            g, sample_index, idx = pgl.sampling.graphsage_sample(graph_np, nodes, args.samples)[0]
            g.tensor()
            feat = feature[torch.as_tensor(sample_index, device=feature.device)]
            pred = model(g, feat)[torch.as_tensor(idx, device=feature.device)]
        else:
            blocks, sample_index = sampler.sample_neighbors(torch.as_tensor(nodes, device=feature.device))
            pred = model(blocks, feature[sample_index])                    # rows of the last block = the batch, in order
        y = labels[torch.as_tensor(nodes, device=feature.device)]
        loss = F.cross_entropy(pred, y)
        if train:
            optim.zero_grad(); loss.backward(); optim.step()
        tot_loss += loss.item() * len(nodes); tot_acc += int((pred.argmax(1) == y).sum().item()); tot += len(nodes)
    return tot_loss / tot, tot_acc / tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sampler", default="gpu", choices=["gpu", "host"])
    ap.add_argument("--epochs", type=int, default=3)
    ap.add_argument("--batch_size", type=int, default=512)
    ap.add_argument("--samples", type=int, nargs="+", default=[25, 10])
    ap.add_argument("--hidden_size", type=int, default=128)
    ap.add_argument("--nodes", type=int, default=20000)
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args()
    torch.manual_seed(0)
    dev = torch.device("cuda:0")
    edges, x, y, train_index, val_index = synthetic_reddit(n=args.nodes)
    graph_np = pgl.Graph(edges=edges, num_nodes=len(x))
    graph_dev = pgl.Graph(edges=edges, num_nodes=len(x)).tensor() if args.sampler == "gpu" else None
    sampler = pgl.sampling.NeighborSampler(graph_dev, args.samples[::-1], seed=1) if graph_dev is not None else None
    feature, labels = torch.as_tensor(x).to(dev), torch.as_tensor(y).to(dev)
    model = GraphSage(x.shape[1], int(y.max()) + 1, len(args.samples), args.hidden_size).to(dev)
    optim = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=torch.cuda.is_available())
    rng = np.random.default_rng(1)
    for epoch in range(args.epochs):
        t0 = time.time()
        tl, ta = run_epoch(args, model, optim, graph_np, graph_dev, sampler, feature, labels, train_index, rng, True)
        with torch.no_grad():
            vl, va = run_epoch(args, model, optim, graph_np, graph_dev, sampler, feature, labels, val_index, rng, False)
        torch.cuda.synchronize()
        print("epoch %d  train loss %.4f acc %.3f | val loss %.4f acc %.3f | %.2f s (%s sampler)" % (
            epoch, tl, ta, vl, va, time.time() - t0, args.sampler))


if __name__ == "__main__":
    main()
