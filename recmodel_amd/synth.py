"""Synthetic implicit-feedback matrices for the WMF benchmarks and parity tests.

Row degrees are Poisson(mean_degree) clipped to >= 1; column ids are drawn from an item-popularity
distribution p_i ~ (i + 1)^-zipf_a (zipf_a = 0 is uniform) and de-duplicated per row; raw counts are
1 + Geometric(0.5).  Generated with torch so the same code runs on the GPU for the large
configurations (a 100 M-entry matrix takes well under a second there) and on the CPU in tests.
"""
import torch

CONFIGS = {
    # name: (n_users, n_items, mean user degree, k, bias)           BASELINE.json "configs"
    "cfg1": (943, 1682, 106, 16, False),        # ML-100K-shaped plumbing case
    "cfg2": (1_000_000, 100_000, 20, 64, False),
    "cfg3": (10_000_000, 1_000_000, 10, 128, True),
    "cfg4": (10_000_000, 1_000_000, 10, 128, False),  # the cfg3 matrix without biases (f = 128: eight 16-wide blocks)
    "cfg3m": (200_000, 1_000_000, 500, 128, True),   # cfg3's item side against a cache-resident user table (kernel lab only)
    "cfg5s": (2_000_000, 200_000, 20, 256, False),   # one GPU's 1/25 slice of cfg5 (k = 256; run with --zipf 1.1 for the power law)
    "tiny": (2000, 500, 12, 16, False),
}


def make_counts(n_users, n_items, mean_degree, seed, device="cpu", zipf_a=0.0, first_user=0):
    """Returns CSR (indptr int64, indices int64, raw counts fp32) of an [n_users, n_items] matrix.

    ``first_user`` offsets the per-row random stream so that disjoint user ranges of one logical
    matrix can be generated independently (weak-scaling runs build N times the users)."""
    gen = torch.Generator(device=device)
    gen.manual_seed(int(seed) + 7919 * int(first_user))
    deg = torch.poisson(torch.full((n_users,), float(mean_degree), device=device), generator=gen).clamp_(min=1)
    deg = deg.to(torch.int64).clamp_(max=n_items)
    rows = torch.repeat_interleave(torch.arange(n_users, device=device), deg)
    u = torch.rand(rows.numel(), device=device, generator=gen, dtype=torch.float64)
    if zipf_a == 0.0:
        cols = (u * n_items).to(torch.int64).clamp_(max=n_items - 1)
    else:
        pop = torch.arange(1, n_items + 1, device=device, dtype=torch.float64) ** (-float(zipf_a))
        cdf = torch.cumsum(pop / pop.sum(), 0)
        cols = torch.searchsorted(cdf, u).clamp_(max=n_items - 1)
    key = torch.unique(rows * n_items + cols)            # sorted by (row, col), duplicates dropped
    rows, cols = key // n_items, key % n_items
    counts = torch.bincount(rows, minlength=n_users)
    indptr = torch.zeros(n_users + 1, dtype=torch.int64, device=device)
    torch.cumsum(counts, 0, out=indptr[1:])
    # 1 + Geometric(0.5): number of fair-coin flips up to and including the first head, plus one
    g = torch.rand(key.numel(), device=device, generator=gen)
    vals = 2.0 + torch.floor(torch.log2(1.0 / (1.0 - g).clamp_(min=1e-12)))
    return indptr, cols, vals.to(torch.float32)


def to_scipy(indptr, indices, values, shape):
    import numpy as np
    import scipy.sparse as sp
    return sp.csr_matrix((values.cpu().numpy(), indices.cpu().numpy().astype(np.int32),
                          indptr.cpu().numpy()), shape=shape)
