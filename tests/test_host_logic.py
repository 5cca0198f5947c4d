"""CPU-only tests of the host layer above the C ABI: constructor parity with the reference's RNG
draw, the synthetic generator, id<->position sharding maps, and early-stop bookkeeping."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from recmodel_amd import WMF, synth
from recmodel_amd.engine import coo_to_csr


def test_constructor_draws_like_reference():
    g = load_golden("init.npz")
    for bias, key in ((False, "items_bias0"), (True, "items_bias1")):
        m = WMF(num_items=37, num_users=5, dim=6, gamma=0.1, weighted=True, bias=bias, seed=1993)
        np.testing.assert_array_equal(m.items, g[key])
        assert m.users is None and m.dim == 6 and m.num_items == 37 and m.dtype == 'float32'
    m = WMF(num_items=9, num_users=5, dim=4, gamma=0.1, weighted=True, seed=7)
    np.testing.assert_array_equal(m.items, g["items_seed7"])


def test_synth_generator_shapes_and_determinism():
    a = synth.make_counts(500, 80, 7, seed=3)
    b = synth.make_counts(500, 80, 7, seed=3)
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    indptr, idx, val = a
    assert indptr[0] == 0 and indptr[-1] == idx.numel() == val.numel()
    deg = indptr[1:] - indptr[:-1]
    assert deg.min() >= 1 and abs(deg.float().mean().item() - 7) < 1.0
    assert val.min() >= 2 and idx.max() < 80
    m = synth.to_scipy(indptr, idx, val, (500, 80))
    assert m.has_sorted_indices and m.nnz == idx.numel()
    m.sum_duplicates()
    assert m.nnz == idx.numel()                       # no duplicates inside a row
    z = synth.make_counts(300, 2000, 5, seed=3, zipf_a=1.0)
    pop = torch.bincount(z[1], minlength=2000)
    assert pop[:20].sum() > pop[-200:].sum()          # skewed towards the head


def test_coo_to_csr_keeps_duplicates_and_order():
    rows = torch.tensor([2, 0, 2, 2, 1])
    cols = torch.tensor([1, 3, 1, 0, 2])
    vals = torch.tensor([1., 2., 3., 4., 5.])
    indptr, idx, v = coo_to_csr(rows, cols, vals, 4)
    assert indptr.tolist() == [0, 1, 2, 5, 5]
    assert idx.tolist() == [3, 2, 0, 1, 1]
    assert v.tolist() == [2., 5., 4., 1., 3.]         # stable: the duplicate (2,1) keeps its stored order


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("chunks", [1, 3, 4])
def test_gathered_positions_are_a_bijection_and_chunk_contiguous(world, chunks):
    """Chunk-major layout of the gathered factor matrix (engine module docstring): every id has its own row,
    and the rows rank r contributes to chunk c are one contiguous range at W * s_c + r * len_c -- what an
    all_gather_into_tensor of that chunk writes."""
    import torch
    from recmodel_amd.engine import gathered_positions
    n = 103
    rpr = (n + world - 1) // world
    L = max(1, (rpr + chunks - 1) // chunks)
    ids = np.arange(n)
    pos = gathered_positions(torch.arange(n), world, rpr, L).numpy()
    assert [gathered_positions(int(i), world, rpr, L) for i in ids] == pos.tolist()      # int and tensor forms agree
    assert len(set(pos.tolist())) == n and pos.max() < world * rpr
    for r in range(world):
        mine = ids[r::world]                                     # local index j <-> id r + W j
        for s_c in range(0, rpr, L):
            len_c = min(L, rpr - s_c)
            j = np.arange(s_c, min(s_c + len_c, len(mine)))
            np.testing.assert_array_equal(pos[mine[j]], world * s_c + r * len_c + (j - s_c))
    if world == 1:
        np.testing.assert_array_equal(pos, ids)


def test_eval_topn_host_route_matches_reference_golden():
    """RecModel.eval_topn for a model WITHOUT the device hook (entry-by-entry rank(), like the reference),
    and with a hook fed by the pre-drawn candidates: both reproduce the reference's seeded Recall@N."""
    from conftest import csr_from, load_golden
    from oracle import wmf_oracle as orc
    from recmodel_amd.base_model import RecModel
    g = load_golden("eval_topn_bias1.npz")
    test = csr_from(g, "test")

    class HostModel(RecModel):
        num_items = test.shape[1]

        def rank(self, items, users, topn=None):
            return orc.rank(g["users"], g["items"], items, users, topn=topn, bias=True)

    res = HostModel().eval_topn(test_mat=test.copy(), topn=g["topn"], rand_sampled=int(g["rand_sampled"]),
                                random_state=int(g["random_state"]))
    np.testing.assert_array_equal(np.array([res[f"Recall@{n}"] for n in g["topn"]], dtype=np.float64), g["recall"])

    class HookModel(HostModel):
        def _hit_counts(self, pair_user, pair_item, pair_row, candidates, slot, topn):
            hits = np.zeros(len(topn), dtype=np.int64)
            for u, it, row in zip(pair_user, pair_item, pair_row):
                cand = candidates[row].copy()
                cand[slot[row]] = it
                s = orc.predict(g["users"], g["items"], [u], cand, True)
                higher = int((np.delete(s, slot[row]) > s[slot[row]]).sum())
                hits += higher < topn
            return hits

    res = HookModel().eval_topn(test_mat=test.copy(), topn=g["topn"], rand_sampled=int(g["rand_sampled"]),
                                random_state=int(g["random_state"]))
    np.testing.assert_array_equal(np.array([res[f"Recall@{n}"] for n in g["topn"]], dtype=np.float64), g["recall"])


def test_default_chunking_leaves_small_sides_whole():
    """With the default chunk count a side is only cut into chunks of at least MIN_CHUNK_ROWS rows (engine.py);
    an explicit count is honoured as given."""
    from fake_kernels import NumpyKernels             # tests/ is on sys.path (conftest lives there)
    from recmodel_amd import engine
    e = engine.AlsEngine(5 * engine.MIN_CHUNK_ROWS, 1000, 4, False, 0.1, device="cpu", kernels=NumpyKernels())
    assert e.world == 1 and len(e.chunk_bounds["users"]) == 1                      # one rank: never chunked by default
    e = engine.AlsEngine(5 * engine.MIN_CHUNK_ROWS, 1000, 4, False, 0.1, device="cpu", kernels=NumpyKernels(), chunks=3)
    assert len(e.chunk_bounds["users"]) == 3 and len(e.chunk_bounds["items"]) == 3
    assert sum(n for _, n in e.chunk_bounds["items"]) == 1000


def test_reduce_mode_is_chosen_from_the_bytes_per_link(monkeypatch):
    """engine.AlsEngine picks the exchange per updated side: reduce-scatter of partial systems when
    rows_per_rank x partial_row_floats is below REDUCE_GAIN x (rows_per_rank of the fixed side) x ld, all-gather otherwise.
    cfg2 under weak scaling over users: 8 ranks -> items in reduce mode (and the user block neither chunked nor gathered),
    2 and 4 ranks -> all-gather."""
    import torch.distributed as dist
    from fake_kernels import NumpyKernels
    from recmodel_amd import engine

    class K64(NumpyKernels):
        def partial_row_floats(self, f):
            return 2816 if f == 64 else super().partial_row_floats(f)      # what libwmf_hip reports for f = 64

    for world, want in ((2, False), (4, False), (8, True)):
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda group=None, w=world: w)
        monkeypatch.setattr(dist, "get_rank", lambda group=None: 1)
        e = engine.AlsEngine(1_000_000 * world, 100_000, 64, False, 0.1, device="meta", kernels=K64())
        assert e.reduce == {"users": False, "items": want}, (world, e.reduce)
        assert len(e.chunk_bounds["users"]) == (1 if want else 4)
        assert len(e.chunk_bounds["items"]) == (3 if want else 1)


def _tiny_engine(n_users=12, n_items=7, dim=3, gamma=0.1):
    from fake_kernels import NumpyKernels
    from recmodel_amd import engine
    return engine.AlsEngine(n_users, n_items, dim, False, gamma, device="cpu", kernels=NumpyKernels())


def test_column_ids_outside_the_catalogue_raise_like_the_reference():
    """The reference indexes the fixed factors with the stored column ids (wmf_model.py:233) and raises IndexError
    for an id >= num_items; the engine turns the ids into gather positions of device kernels, so it has to refuse
    them when the matrix is handed over (count_mat, eval_mat and utility_mat alike)."""
    eng = _tiny_engine()
    indptr = torch.arange(0, 13, dtype=torch.int64)
    vals = torch.ones(12)
    ok = torch.arange(12) % 7
    eng.set_interactions(indptr, ok, vals)                                   # fits
    eng.make_eval_shard(indptr, ok, vals)
    wide = ok.clone()
    wide[5] = 7                                                              # one id == num_items
    with pytest.raises(IndexError):
        eng.set_interactions(indptr, wide, vals)
    with pytest.raises(IndexError):
        eng.make_eval_shard(indptr, wide, vals)
    with pytest.raises(IndexError):
        eng.set_interactions(indptr, -wide, vals)                            # negative ids
    with pytest.raises(ValueError):
        eng.set_interactions(indptr[:-1], ok[:-1], vals[:-1])                # 11 rows for 12 users
    with pytest.raises(IndexError):                                          # the building block refuses too
        coo_to_csr(torch.tensor([0, 4]), torch.tensor([0, 0]), torch.ones(2), 4)


def test_failed_gramian_factorisation_is_sticky():
    """gamma = 0 and fewer items than factors: Gram(items) is singular, the users half step fails.  The next Gramian
    (of all-zero users, plus gamma I = 0) fails too in this case, but even a later success must not clear the flag:
    check_numerics() after both half steps has to raise, as the reference's np.linalg.solve does."""
    from recmodel_amd import _lib
    eng = _tiny_engine(n_users=6, n_items=2, dim=3, gamma=0.0)
    indptr = torch.arange(0, 7, dtype=torch.int64)
    eng.set_interactions(indptr, torch.arange(6) % 2, torch.ones(6))
    eng.set_factors("items", np.ones((2, 3), dtype=np.float32))
    eng.half_step("users")
    assert int(eng.info[0]) == 1
    eng.gamma = 1.0                                                          # the items half step factorises fine ...
    eng.half_step("items")
    assert int(eng.info[0]) == 1                                             # ... and does not hide the failure
    with pytest.raises(_lib.WmfNumericError):
        eng.check_numerics()
    assert int(eng.info[0]) == 0                                             # reset by the check that reported it
    eng.check_numerics()


def test_public_factor_arrays_are_read_only_after_training_and_writable_ones_are_never_cached():
    """predict / rank / eval_prec score device copies of model.users / model.items.  What train() leaves behind is
    read-only (an in-place edit raises instead of silently not reaching the device); arrays the caller assigns stay
    writable and are uploaded on every call (wmf_model.WMF._device_factors)."""
    m = WMF(num_items=5, num_users=4, dim=2, gamma=0.1, weighted=True)
    m.users = np.zeros((4, 2), dtype=np.float32)
    assert m._stale()                                     # writable arrays: never trusted to match the engine
    m._freeze()
    assert not m._stale()
    with pytest.raises(ValueError):
        m.items[0, 0] = 1.0
    m.items = m.items.copy()                              # assigning a fresh array works as in the reference
    m.items[0, 0] = 1.0
    assert m._stale()


def test_balanced_assignment_equalises_cost_on_a_power_law_matrix():
    """rows are dealt so that sum(nnz f^2 + f^3) per rank is equal.  Zipf(1.1) item popularity: the
    heaviest item row alone holds ~5 % of all entries; round-robin leaves the ranks 20 % and more apart, the balanced
    deal within 5 % (here: within 1 %), with equal row counts up to the few rows that level the heavy head."""
    from recmodel_amd.engine import balanced_assignment, row_cost
    n_users, n_items, f, world = 60_000, 20_000, 64, 8
    ip, idx, _ = synth.make_counts(n_users, n_items, 20, seed=5, zipf_a=1.1)
    deg_items = torch.bincount(idx, minlength=n_items)
    deg_users = ip[1:] - ip[:-1]
    for deg in (deg_items, deg_users):
        cost = row_cost(deg, f)
        owner, local, counts = balanced_assignment(cost, world)
        per_rank = torch.zeros(world, dtype=torch.float64).index_add_(0, owner.to(torch.int64), cost)
        spread = float((per_rank.max() - per_rank.min()) / per_rank.mean())
        assert spread <= 0.05, (spread, per_rank.tolist())
        assert sum(counts) == deg.numel() and max(counts) - min(counts) <= max(64, deg.numel() // 100)
        # local rows: a bijection onto [0, count) inside every rank, ascending with the id
        for r in range(world):
            mine = torch.nonzero(owner == r).flatten()
            assert torch.equal(local[mine].to(torch.int64), torch.arange(mine.numel()))
    # what round-robin does to the item side of the same matrix
    cost = row_cost(deg_items, f)
    rr = torch.zeros(world, dtype=torch.float64).index_add_(0, torch.arange(n_items) % world, cost)
    assert float((rr.max() - rr.min()) / rr.mean()) > 0.05
    # determinism: the same deal from the same costs (every rank computes it for itself)
    o2, l2, c2 = balanced_assignment(row_cost(deg_items, f), world)
    o1, l1, c1 = balanced_assignment(row_cost(deg_items, f), world)
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and c1 == c2
