"""Lab (not a test): cost of launch gaps at cfg2 -- one ALS iteration as eager launches vs replayed as one HIP graph."""
import sys, time
import torch
sys.path.insert(0, '.')
from recmodel_amd import _lib, synth
from recmodel_amd.engine import AlsEngine
from recmodel_amd import WMF

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda")
val = 10 * torch.log(1 + val)
eng = AlsEngine(n_users, n_items, k, bias, 0.1)
eng.set_interactions(ip, idx, val)
eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
def step():
    eng.half_step("users"); eng.half_step("items")
for _ in range(3): step()
torch.cuda.synchronize()
def timeit(fn, n=20):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print(f"{cfg} eager: {timeit(step):.3f} ms / iteration")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
print(f"{cfg} graph: {timeit(g.replay):.3f} ms / iteration")
eng.check_numerics()
