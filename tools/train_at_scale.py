"""End-to-end WMF.train() through the public class surface at benchmark scale (cfg2: 1 M x 100 K, 20 M entries):
scipy CSR in, host factors out.  Prints wall times; not a test.  Usage: python tools/train_at_scale.py [cfg] [iterations]"""
import sys, time
import numpy as np
sys.path.insert(0, '.')
from recmodel_amd import WMF, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
t0 = time.perf_counter()
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995)
counts = synth.to_scipy(ip, idx, val, (n_users, n_items)).astype(np.float32)
print(f"synthetic matrix {counts.shape}, {counts.nnz} entries: {time.perf_counter() - t0:.1f} s")
m = WMF(num_items=n_items, num_users=n_users, dim=k, gamma=0.1, weighted=True, bias=bias)
t0 = time.perf_counter()
last = m.train(utility_mat=counts, count_mat=counts, iterations=iters, eval_mat=counts, stopping_rounds=iters + 1, verbose=0)
dt = time.perf_counter() - t0
print(f"train({iters} iterations incl. eval_prec each): {dt:.2f} s, last iteration index {last}")
t0 = time.perf_counter()
mse = m.eval_prec(counts)
print(f"eval_prec: {time.perf_counter() - t0:.2f} s, mse {mse:.4f}; factors {m.users.shape} {m.users.dtype}, {m.items.shape}")
t0 = time.perf_counter()
top = m.rank(np.arange(n_items), 12345, topn=10)
print(f"rank over the whole catalogue: {time.perf_counter() - t0:.3f} s -> {top}")
t0 = time.perf_counter()
tops = m.rank(np.arange(n_items), list(range(0, 2000)), topn=10)
dt = time.perf_counter() - t0
print(f"rank for 2000 users over the whole catalogue in one call: {dt:.3f} s ({2000 / dt:.0f} users/s), first: {tops[0][:5]}")
