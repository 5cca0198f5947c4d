"""Oracle checks where the fast paths actually run: the bench workloads at their FULL size (cfg2: k = 64, 1 M x 100 K,
20 M entries; cfg3: k = 128 + biases, 10 M x 1 M, 100 M entries).  After each of the first two half steps, 2000+ rows of
the side just solved -- the heaviest, the emptiest and a random sample -- are recomputed by the oracle's
``solve_row`` (RecModel/wmf_model.py:231-239) in float64 from the fixed-side factors THE DEVICE used, and compared row by
row with what the device wrote.  Seconds of CPU per side; the matrices themselves never leave the GPU except for the
sampled rows."""
import numpy as np
import pytest
import torch

from oracle import wmf_oracle as orc

pytestmark = pytest.mark.gpu

ROW_TOL, FRO_TOL = 5e-4, 5e-5          # the stated fp32 tolerance of one half step


def _sample_rows(deg, n_random, n_extreme, rng):
    order = np.argsort(deg, kind="stable")
    pick = np.concatenate([order[-n_extreme:], order[:n_extreme], rng.choice(deg.size, n_random, replace=False)])
    return np.unique(pick)


def _gramian64(Y_dev, f, lam, bias):
    """Y~^T Y~ + lam I in float64 (wmf_model.py:215 / :328-332), accumulated over slabs of the device matrix."""
    G = np.zeros((f, f))
    for lo in range(0, Y_dev.shape[0], 1 << 20):
        y = Y_dev[lo: lo + (1 << 20), :f].double()
        if bias:
            y[:, 0] = 1.0
        G += (y.T @ y).cpu().numpy()
    return G + lam * np.eye(f)


def _check_side(eng, side, rng, n_random=1600, n_extreme=200):
    """Rows of ``side`` (just solved) against oracle.solve_row on the fixed side's device factors."""
    fixed = "items" if side == "users" else "users"
    f, bias = eng.f, eng.bias
    csr = eng.csr[side]
    indptr = csr.indptr.cpu().numpy()
    deg = np.diff(indptr)[: eng.n[side]]
    rows = _sample_rows(deg, n_random, n_extreme, rng)
    Y_dev = eng.X[fixed]                                  # one rank: the factor block itself, ids = positions
    G = _gramian64(Y_dev[: eng.n[fixed]], f, eng.gamma, bias)
    got_all = eng.factors[side]
    worst, num, den, classes = 0.0, 0.0, 0.0, {}
    rows_t = torch.from_numpy(rows).to(eng.device)
    got = got_all[rows_t][:, :f].double().cpu().numpy()
    for j, u in enumerate(rows):
        lo, hi = int(indptr[u]), int(indptr[u + 1])
        if hi == lo:
            assert not got[j].any(), f"{side} row {u} has no entries and is not zero"
            classes["d=0"] = classes.get("d=0", 0) + 1
            continue
        idx = csr.indices[lo:hi].long()
        w = csr.values[lo:hi].double().cpu().numpy()
        U = Y_dev[idx][:, :f].double().cpu().numpy()
        if bias:                                          # wmf_model.py:328-343: the fixed side's column 0 is its bias
            w = w - U[:, 0]
            U[:, 0] = 1.0
        want = orc.solve_row(G, U, np.arange(hi - lo), w)
        e, n_ = np.linalg.norm(got[j] - want), np.linalg.norm(want)
        worst = max(worst, e / n_)
        num += e * e
        den += n_ * n_
        d = hi - lo
        c = "d<=8" if d <= 8 else "d<=16" if d <= 16 else "d<=32" if d <= 32 else "d<=4096" if d <= 4096 else "d>4096"
        classes[c] = classes.get(c, 0) + 1
    assert rows.size >= 1900                              # (a few of the random picks coincide with the extremes)
    assert worst <= ROW_TOL and np.sqrt(num / den) <= FRO_TOL, (side, worst, np.sqrt(num / den), classes)
    return worst, np.sqrt(num / den), classes


@pytest.mark.parametrize("cfg", ["cfg2", "cfg3"])
def test_full_size_half_steps_vs_oracle(cfg):
    from recmodel_amd import WMF, synth
    from recmodel_amd.engine import AlsEngine
    n_users, n_items, dbar, k, bias = synth.CONFIGS[cfg]
    ip, idx, val = synth.make_counts(n_users, n_items, dbar, 1995, device="cuda")
    w = 10 * torch.log(1 + val)
    eng = AlsEngine(n_users, n_items, k, bias, 0.1)
    eng.set_interactions(ip, idx, w)
    del ip, idx, val, w
    eng.set_factors("items", WMF(num_items=n_items, num_users=1, dim=k, gamma=0.1, weighted=True, bias=bias).items)
    rng = np.random.default_rng(7)
    report = {}
    for side in ("users", "items", "users"):              # the third half step runs against device-made item factors
        eng.half_step(side)
        eng.check_numerics()
        report[side] = _check_side(eng, side, rng)
    print(cfg, {s: (f"{r[0]:.1e}", f"{r[1]:.1e}", r[2]) for s, r in report.items()})
    # every kernel family of the bench ran: rows in the d <= 8 / <= 16 / <= 32 / heavy classes were sampled on the users side
    assert {"d<=8", "d<=16"} <= set(report["users"][2]) and "d<=4096" in report["items"][2]
