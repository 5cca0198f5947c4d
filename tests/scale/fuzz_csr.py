"""Randomised sweep (not a pytest) of wmf_coo_to_csr and the radix sort under it (wmf_sort.hip, round 3): shapes from 1 x 1 to
millions of rows / columns (key widths 1 .. 44 bits), entry counts around the sort's 2048-key tiles, adversarial orders (sorted,
reversed, one row, one column, many duplicates), against NumPy's stable lexsort.  Usage: python tests/scale/fuzz_csr.py [cases] [seed]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from recmodel_amd.engine import HipKernels

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
K = HipKernels()
t0 = time.perf_counter()
for case in range(cases):
    n_rows = int(2 ** rng.uniform(0, 22)); n_cols = int(2 ** rng.uniform(0, 22))
    base = int(rng.choice([0, 1, 2, 2047, 2048, 2049, 4096, 65535, 65536, 65537, 1 << 20, 3000000]))
    nnz = max(0, base + int(rng.integers(-3, 4))) if rng.integers(2) else int(2 ** rng.uniform(0, 22))
    kind = rng.choice(["random", "sorted", "reversed", "one_row", "one_col", "dups", "zipf"])
    if kind == "one_row": rows = np.full(nnz, int(rng.integers(n_rows))); cols = rng.integers(0, n_cols, nnz)
    elif kind == "one_col": rows = rng.integers(0, n_rows, nnz); cols = np.full(nnz, int(rng.integers(n_cols)))
    elif kind == "dups": rows = rng.integers(0, min(n_rows, 3), nnz); cols = rng.integers(0, min(n_cols, 3), nnz)
    elif kind == "zipf": rows = np.minimum(rng.zipf(1.3, nnz) - 1, n_rows - 1); cols = rng.integers(0, n_cols, nnz)
    else: rows = rng.integers(0, n_rows, nnz); cols = rng.integers(0, n_cols, nnz)
    rows, cols = rows.astype(np.int64), cols.astype(np.int64)
    if kind in ("sorted", "reversed"):
        o = np.lexsort((cols, rows)); o = o[::-1] if kind == "reversed" else o
        rows, cols = rows[o], cols[o]
    vals = rng.standard_normal(nnz).astype(np.float32)
    print(f"case {case:3d}: {n_rows} x {n_cols}, nnz {nnz}, {kind:8s}", end=" ", flush=True)
    ptr, idx, val = K.coo_to_csr(torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda(), torch.from_numpy(vals).cuda(), n_rows, n_cols)
    order = np.lexsort((np.arange(nnz), cols, rows))
    assert np.array_equal(idx.cpu().numpy(), cols[order].astype(np.int32)), "column order"
    assert np.array_equal(val.cpu().numpy(), vals[order]), "values (stability)"
    assert np.array_equal(ptr.cpu().numpy(), np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_rows))])), "indptr"
    print("ok")
print(f"{cases} cases in {time.perf_counter() - t0:.1f} s")
