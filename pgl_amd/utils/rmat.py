"""Synthetic power-law graphs for benchmarks and tests.

RMAT / Graph500 generator: every edge picks one quadrant per bit level with probabilities
(a, b, c, d) = (0.57, 0.19, 0.19, 0.05); duplicates and self loops are kept; node ids are randomly
permuted once so that hubs are not clustered at low ids.  Runs on whatever device is asked for
(GPU generation of |E| = 100 M takes well under a second)."""
import torch


def rmat_edges(scale, num_edges, a=0.57, b=0.19, c=0.19, seed=42, device="cpu", permute=True):
    """Returns an int64 [num_edges, 2] tensor of (src, dst) over N = 2**scale nodes."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    src = torch.zeros(num_edges, dtype=torch.int64, device=dev)
    dst = torch.zeros(num_edges, dtype=torch.int64, device=dev)
    for _ in range(scale):
        r = torch.rand(num_edges, generator=g, device=dev)
        src_bit = (r >= a + b).to(torch.int64)                       # quadrants c, d -> lower half
        dst_bit = ((r >= a) & (r < a + b) | (r >= a + b + c)).to(torch.int64)   # quadrants b, d
        src = (src << 1) | src_bit
        dst = (dst << 1) | dst_bit
    if permute:
        perm = torch.randperm(1 << scale, generator=g, device=dev)
        src, dst = perm[src], perm[dst]
    return torch.stack([src, dst], dim=1)


def rmat_slabs(scale, num_edges, slab_edges, a=0.57, b=0.19, c=0.19, seed=42, device="cpu", fold=None):
    """The same kind of graph as rmat_edges, handed out slab by slab (a generator of int64 [<= slab_edges, 2] tensors) so that
    nobody ever holds the whole edge list: BASELINE config 5 (|E| = 1.6 B, SURVEY 8d "generated per-partition on device").  One
    node permutation (drawn first, from `seed`) serves every slab; slab k draws its edges from its own stream (seed, k), so any
    rank can regenerate any slab.  fold: node ids are taken modulo this many nodes (a node count that is not a power of two)."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    perm = torch.randperm(1 << scale, generator=g, device=dev)
    k, done = 0, 0
    while done < num_edges:
        n = min(int(slab_edges), num_edges - done)
        gk = torch.Generator(device=dev)
        gk.manual_seed((seed * 1000003 + k + 1) & 0x7FFFFFFFFFFF)
        src = torch.zeros(n, dtype=torch.int64, device=dev)
        dst = torch.zeros(n, dtype=torch.int64, device=dev)
        for _ in range(scale):
            r = torch.rand(n, generator=gk, device=dev)
            src = (src << 1) | (r >= a + b).to(torch.int64)
            dst = (dst << 1) | ((r >= a) & (r < a + b) | (r >= a + b + c)).to(torch.int64)
        e = torch.stack([perm[src], perm[dst]], dim=1)
        if fold:
            e = e % int(fold)
        yield e
        done += n
        k += 1
