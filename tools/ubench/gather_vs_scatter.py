# Random 8-byte gather against random 8-byte scatter (what the interp kernels' output write is),
# M = 1e7 complex64, permutation index: torch's index kernels as a quick reference.
import time, torch
M = 10_000_000
g = torch.Generator(device='cuda').manual_seed(0)
perm = torch.randperm(M, device='cuda', generator=g)
perm32 = perm.to(torch.int32)
src = torch.randn(M, 2, device='cuda').view(torch.float64).squeeze(-1)   # 8-byte elements
out = torch.empty_like(src)
def t(fn, n=20):
  for _ in range(3): fn()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(n): fn()
  torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e6
print('gather  out = src[perm]      %.1f us' % t(lambda: torch.index_select(src, 0, perm, out=out)))
print('scatter out[perm] = src      %.1f us' % t(lambda: out.index_copy_(0, perm, src)))
print('copy    out = src            %.1f us' % t(lambda: out.copy_(src)))
# tile-sorted-like locality: the permutation restricted to blocks of 4096 consecutive outputs
blk = 4096
p2 = (torch.arange(M, device='cuda') // blk) * blk
p2 = p2 + torch.argsort(torch.rand(M, device='cuda', generator=g).view(-1)[:M].reshape(-1))[:M] % blk if False else perm
