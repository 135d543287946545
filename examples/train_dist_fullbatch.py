#!/usr/bin/env python3
"""Full-batch multi-GPU training on a row-partitioned graph: BASELINE config 3's flow (GraphSAGE mean-aggregate, METIS
partition + RCCL halo all-to-all-v) at whatever size is asked for.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_dist_fullbatch.py \
        --model sage --scale 20 --edges 20000000 --dim 100 --epochs 10

What the reference does for this (examples/citation_benchmark/multi_gpu_train.py with pgl.DistGPUGraph) is replicate all
node features on every GPU, shard the edges and all-reduce the whole [N, d] output after every aggregation.  Here every
rank holds only its own rows: features, labels, activations and gradients of the OWNED nodes; the layers of pgl_amd.nn take
the DistGraph in place of a Graph, halo rows travel once per aggregation (forward) and once per aggregation (backward), and
the only other collective is the all-reduce of the (small) parameter gradients.

PGLAMD_DRYRUN=1 puts every rank on cuda:0 over gloo (single-GPU boxes: exercises the code path, times mean nothing).
"""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import pgl_amd as pgl  # noqa: E402
from pgl_amd.distributed import DistGraph  # noqa: E402
from pgl_amd.utils.rmat import rmat_edges  # noqa: E402


class Net(torch.nn.Module):
    def __init__(self, kind, din, hidden, classes):
        super().__init__()
        if kind == "sage":
            self.convs = torch.nn.ModuleList([pgl.nn.GraphSageConv(din, hidden, "mean"), pgl.nn.GraphSageConv(hidden, hidden, "mean")])
        elif kind == "gcn":
            self.convs = torch.nn.ModuleList([pgl.nn.GCNConv(din, hidden, activation="relu"), pgl.nn.GCNConv(hidden, hidden, activation="relu")])
        else:
            self.convs = torch.nn.ModuleList([pgl.nn.GATConv(din, hidden // 8, 0.0, 0.0, 8, activation="elu"),
                                              pgl.nn.GATConv(hidden, hidden // 8, 0.0, 0.0, 8, activation="elu")])
        self.kind = kind
        self.out = pgl.nn.Linear(hidden, classes)            # (split-reduction weight / bias gradients: the rank's rows are many)

    def forward(self, g, x):
        for conv in self.convs:
            x = conv(g, x, act="relu") if self.kind == "sage" else conv(g, x)
        return self.out(x)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="sage", choices=["sage", "gcn", "gat"])
    ap.add_argument("--scale", type=int, default=16)
    ap.add_argument("--edges", type=int, default=1_000_000)
    ap.add_argument("--dim", type=int, default=100)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--classes", type=int, default=47)
    ap.add_argument("--epochs", type=int, default=10)
    ap.add_argument("--partition", default="kway", choices=["kway", "random"])
    ap.add_argument("--lr", type=float, default=0.01)
    args = ap.parse_args()

    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    dry = os.environ.get("PGLAMD_DRYRUN") == "1"
    dev = torch.device("cuda", 0 if dry else int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if dry else "nccl")
    N = 1 << args.scale
    edges = rmat_edges(args.scale, args.edges, seed=42, device=dev)                     # identical on every rank
    sym = torch.cat([edges, edges.flip(1)], 0)                                           # undirected, as the examples train
    t0 = time.time()
    dg = DistGraph.from_global(sym, N, rank, world, method=args.partition, device=dev)
    if rank == 0:
        print("partition + plan: %.1f s   %s" % (time.time() - t0, dg.stats()), flush=True)
    # synthetic task: labels = a function of the community-free RMAT id (learnable from features), features = noisy one-hot-ish
    gen = torch.Generator(device=dev); gen.manual_seed(7)
    y_all = torch.randint(0, args.classes, (N,), generator=gen, device=dev)
    proto = torch.randn(args.classes, args.dim, generator=gen, device=dev)
    x_all = proto[y_all] * 0.5 + torch.randn(N, args.dim, generator=gen, device=dev)
    train_all = torch.rand(N, generator=gen, device=dev) < 0.5
    x, y, train = dg.take_owned(x_all), dg.take_owned(y_all), dg.take_owned(train_all)     # from here on: owned rows only
    del x_all, y_all, train_all

    torch.manual_seed(0)                                                                  # same initial parameters on every rank
    model = Net(args.model, args.dim, args.hidden, args.classes).to(dev)
    opt = torch.optim.Adam(model.parameters(), lr=args.lr, capturable=torch.cuda.is_available())
    n_train = torch.tensor([float(train.sum())], device=dev)
    if world > 1:
        buf = n_train.cpu() if dry else n_train
        dist.all_reduce(buf); n_train = buf.to(dev)
    losses = []
    for epoch in range(args.epochs):
        torch.cuda.synchronize(); t0 = time.time()
        logits = model(dg, x)
        loss = pgl.nn.functional.cross_entropy(logits[train], y[train], reduction="sum") / n_train          # global mean over all ranks' train nodes
        opt.zero_grad(set_to_none=True)
        loss.backward()
        if world > 1:                                                                      # data-parallel parameters: sum of the ranks' gradients
            for p in model.parameters():
                if dry:
                    g = p.grad.cpu(); dist.all_reduce(g); p.grad.copy_(g)
                else:
                    dist.all_reduce(p.grad)
        opt.step()
        torch.cuda.synchronize(); dt = time.time() - t0
        tot = loss.detach().clone()
        if world > 1:
            b = tot.cpu() if dry else tot
            dist.all_reduce(b); tot = b
        losses.append(float(tot))
        if rank == 0:
            print("epoch %2d  loss %.4f  %.1f ms" % (epoch, losses[-1], dt * 1e3), flush=True)
    if rank == 0:
        print("LOSSES " + " ".join("%.6f" % v for v in losses), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
