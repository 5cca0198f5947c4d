"""Lab: what one pass w_eff = values - bias[indices] costs at cfg3's size (100 M entries; bias table of 10 M floats, or the 8-byte pairs)."""
import torch
dev = torch.device("cuda:0")
nnz, m = 100_000_000, 10_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
idx = torch.randint(0, m, (nnz,), device=dev, generator=g, dtype=torch.int32)
idx, _ = torch.sort(idx.view(-1, 100), dim=1)          # rows of 100 sorted ids, like a CSR row
idx = idx.view(-1)
vals = torch.rand(nnz, device=dev)
bias = torch.rand(m, device=dev)
pairs = torch.rand(m, 2, device=dev)
idx64 = idx.long()
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
print("torch  vals - bias[idx]        (4-byte table, 40 MB): %.3f ms" % t(lambda: vals - bias[idx64]))
print("torch  vals - pairs[idx, 1]    (8-byte table, 80 MB): %.3f ms" % t(lambda: vals - pairs[idx64, 1]))
print("torch  vals - vals (streaming only)                 : %.3f ms" % t(lambda: vals - vals))
