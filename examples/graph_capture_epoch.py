#!/usr/bin/env python3
"""Launch-bound regime (BASELINE config 0, Cora-sized graph): one full GCN training step -- forward,
backward and Adam -- captured once into a HIP graph and replayed, next to the eager loop.
Every pgl_amd op is a plain kernel launch (plus one 8-byte memset node) on torch's current stream
with caller-owned buffers and no host synchronisation, so the whole step is capturable as is.

    python examples/graph_capture_epoch.py
(reference numbers for context, legacy PGL on a V100: GCN 4.7 ms / GAT 11.9 ms per Cora epoch,
 legacy/docs/source/md/introduction.md:44-53)"""
import os
import sys
import time

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pgl_amd as pgl  # noqa: E402
from train_citation import Net, synthetic_cora  # noqa: E402


def main(model="gcn", iters=200):
    torch.manual_seed(0)
    edges, x, y, tr, _, _ = synthetic_cora()
    g = pgl.Graph(num_nodes=x.shape[0], edges=edges, node_feat={"words": x})
    g.indegree(); g.tensor()
    dev = g.edges.device
    g.adj_src_index                                  # transposed index for the backward, built before capture
    yt, trt = torch.from_numpy(y).to(dev), torch.from_numpy(tr).to(dev)
    net = Net(model, x.shape[1], 64, 7).to(dev)
    if model == "gat":
        for l in (net.l1, net.l2):
            l.attn_drop = 0.0                        # the in-kernel dropout seed is a launch argument: frozen under replay
    opt = torch.optim.Adam(net.parameters(), lr=0.01, weight_decay=5e-4, capturable=True)
    feat = g.node_feat["words"]

    def step():
        loss = F.cross_entropy(net(g, feat)[trt], yt[trt])
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / iters * 1e3

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        static_loss = step()
    torch.cuda.synchronize()
    l0 = None
    t0 = time.perf_counter()
    for i in range(iters):
        graph.replay()
        if i == 0:
            l0 = float(static_loss.detach())
    torch.cuda.synchronize()
    replay = (time.perf_counter() - t0) / iters * 1e3
    print("%s  train step  eager %.3f ms   hipGraph replay %.3f ms   (loss %.4f -> %.4f over %d replays)" %
          (model, eager, replay, l0, float(static_loss.detach()), iters))
    return eager, replay, l0, float(static_loss.detach())


if __name__ == "__main__":
    for m in sys.argv[1:] or ["gcn", "gat", "sage"]:
        main(m)
