import torch
dev = torch.device("cuda:0")
N = 1 << 20
x = torch.randn(N, 128, device=dev); g = torch.randn(N, 128, device=dev)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / reps
ref = g.t() @ x
print("g.t() @ x                 %.3f ms" % t(lambda: g.t() @ x))
print("(x.t() @ g).t()           %.3f ms" % t(lambda: (x.t() @ g).t()))
for S in (16, 64, 256, 1024):
    f = lambda: torch.bmm(g.view(S, N // S, 128).transpose(1, 2), x.view(S, N // S, 128)).sum(0)
    print("split-K bmm S=%-4d         %.3f ms  maxerr %.2e" % (S, t(f), float((f() - ref).abs().max() / ref.abs().max())))
print("g @ W (dX)                %.3f ms" % t(lambda: g @ torch.randn(128, 128, device=dev)))
g16 = torch.randn(N, 16, device=dev)
print("g16.t() @ x               %.3f ms" % t(lambda: g16.t() @ x))
print("x @ W[128,16]             %.3f ms" % t(lambda: x @ torch.randn(128, 16, device=dev)))
