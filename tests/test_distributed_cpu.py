"""world_size-2 gloo test of the multi-rank host path on CPU.

Each rank runs AlsEngine with the NumPy stand-in backend: users and items dealt round-robin, local rows
solved in four chunks whose all-gathers are started asynchronously, one all-reduce of the f x f Gramian per
half step, the gathered matrix whitened on every rank, factors read back in id order.  The result must
equal the single-process oracle on the full matrix."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, bias, out_path, reduce_mode=None, pipe_mode=None, dim=6, sizes=(203, 57)):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items = sizes                     # not multiples of the world size: padding rows exist
        indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=11)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=4, reduce_mode=reduce_mode, pipe_mode=pipe_mode)
        assert eng.split == (bias and dim in (16, 128))     # k = 16 m with biases: packed body + {last feature, bias} pairs
        assert eng.rolled == (bias and dim == 128 and os.environ.get("WMF_ROLLED", "1") != "0")   # ... in rolled coordinates at k = 128 (wmf_row_transform modes 3 / 4)
        if reduce_mode:
            assert eng.reduce == {"users": False, "items": True}             # the smaller side travels as partial systems
        assert len(eng.csr_chunks) == 0 and len(eng.chunk_bounds["users"]) == 4
        assert eng.world == world and eng.rank == rank
        eng.set_interactions(indptr, indices, values)
        if pipe_mode:
            assert eng.pipe == {"users": True, "items": True} and len(eng.csr_pipe["items"]) == 4
        # every stored entry lands on exactly one rank, in both orientations
        for side in ("users", "items"):
            t = torch.tensor([eng.csr[side].nnz], dtype=torch.int64)
            dist.all_reduce(t)
            assert int(t) == indices.numel()
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(2):
            eng.half_step("users")
            eng.half_step("items")
        eng.check_numerics()
        shard = eng.make_eval_shard(indptr, indices, counts)
        sq, ab, cnt = eng.eval_sums(shard)
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items, sums=np.array([sq, ab, cnt]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bias,reduce_mode,pipe_mode", [(False, None, None), (True, None, None), (False, True, None),
                                                        (True, True, None), (False, False, True), (True, False, True)])
def test_two_rank_als_matches_single_process_oracle(tmp_path, bias, reduce_mode, pipe_mode, dim=6, sizes=(203, 57)):
    """reduce_mode=True: the item half step accumulates partial systems on every rank and reduce-scatters them
    (here an all-reduce + slice: gloo has no reduce-scatter); the user block is then only gathered on demand.
    pipe_mode=True: both half steps accumulate over each gathered chunk of the fixed side as it arrives."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), bias, out, reduce_mode, pipe_mode, dim, sizes), nprocs=2, join=True)
    got = np.load(out)
    n_users, n_items = sizes
    indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=11)
    raw = synth.to_scipy(indptr, indices, counts, (n_users, n_items))
    C = raw.astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    for _ in range(2):
        users = step(items, C, 0.1)
        items = step(users, CT, 0.1)
    atol = 2e-5 if dim < 64 else 5e-5        # (f = 129: the float32 whitened side of the stand-in costs more, rolled or not)
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=atol)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=atol)
    mse = orc.eval_prec(users, items, raw, bias)
    assert abs(got["sums"][0] / got["sums"][2] - mse) <= 1e-4 * mse
    assert got["sums"][2] == raw.nnz


@pytest.mark.parametrize("reduce_mode,pipe_mode", [(None, None), (True, None), (False, True)])
def test_two_ranks_split_layout_of_the_whitened_side(tmp_path, reduce_mode, pipe_mode):
    """k = 16 with biases (f = 17): the library keeps the whitened fixed side as a packed body and {last feature, bias}
    pairs (include/wmf_hip.h, wmf_row_transform); the engine sizes and slices the two arrays accordingly in the gather,
    reduce and pipelined modes (row ranges per chunk, this rank's block only in reduce mode, no w_eff pass)."""
    test_two_rank_als_matches_single_process_oracle(tmp_path, True, reduce_mode, pipe_mode, dim=16)


@pytest.mark.parametrize("reduce_mode,pipe_mode", [(None, None), (True, None), (False, True)])
def test_two_ranks_rolled_coordinates_of_the_whitened_side(tmp_path, reduce_mode, pipe_mode):
    """k = 128 with biases (f = 129): the engine whitens with wmf_row_transform mode 3, solves with WMF_SOLVE_ROLLED and un-whitens
    with mode 4 (include/wmf_hip.h) in the gather, reduce and pipelined modes alike -- a permutation of the whitened features that
    must not show in the factors."""
    test_two_rank_als_matches_single_process_oracle(tmp_path, True, reduce_mode, pipe_mode, dim=128, sizes=(1203, 457))


@pytest.mark.parametrize("mode", ["gather", "reduce", "pipe"])
def test_three_ranks_all_modes(tmp_path, mode):
    """An odd world size (padding rows on two of the three ranks, uneven last chunk) through each exchange mode."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    kw = {"gather": (False, False), "reduce": (True, None), "pipe": (False, True)}[mode]
    mp.spawn(_worker, args=(3, _free_port(), False, out, kw[0], kw[1]), nprocs=3, join=True)
    got = np.load(out)
    n_users, n_items, dim = 203, 57, 6
    indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=11)
    C = synth.to_scipy(indptr, indices, counts, (n_users, n_items)).astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, False)
    for _ in range(2):
        users = orc.recompute_factors(items, C, 0.1)
        items = orc.recompute_factors(users, CT, 0.1)
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=2e-5)


@pytest.mark.parametrize("mode", ["reduce", "pipe"])
def test_eight_ranks(tmp_path, mode):
    """The world size of the driver's scaling run, through the two modes it will pick there (reduce mode for the items
    at 8 GPUs, pipelined gather at 2 and 4), on CPU ranks with the stand-in kernels."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    kw = {"reduce": (True, None), "pipe": (False, True)}[mode]
    mp.spawn(_worker, args=(8, _free_port(), False, out, kw[0], kw[1]), nprocs=8, join=True)
    got = np.load(out)
    n_users, n_items, dim = 203, 57, 6
    indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=11)
    C = synth.to_scipy(indptr, indices, counts, (n_users, n_items)).astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, False)
    for _ in range(2):
        users = orc.recompute_factors(items, C, 0.1)
        items = orc.recompute_factors(users, CT, 0.1)
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=2e-5)


def test_single_rank_engine_with_stand_in_matches_oracle():
    """world = 1 through the same code path (no process group)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from fake_kernels import NumpyKernels
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    from recmodel_amd.engine import AlsEngine
    n_users, n_items, dim = 90, 40, 5
    indptr, indices, counts = synth.make_counts(n_users, n_items, 4, seed=3)
    values = (10 * torch.log(1 + counts)).to(torch.float32)
    eng = AlsEngine(n_users, n_items, dim, False, 0.1, device="cpu", kernels=NumpyKernels())
    eng.set_interactions(indptr, indices, values)
    eng.set_factors("items", orc.init_items(n_items, dim))
    eng.half_step("users")
    C = synth.to_scipy(indptr, indices, values, (n_users, n_items)).astype(np.float64)
    want = orc.recompute_factors(orc.init_items(n_items, dim), C, 0.1)
    np.testing.assert_allclose(eng.get_factors("users"), want, rtol=2e-4, atol=2e-5)
    assert eng.algorithmic_bytes_half("users") == C.nnz * (4 * 5 + 8) + n_users * (4 * 5 + 4) + 4 * 25 + 4 * n_items * 5
    # the chunked solve (what multi-rank runs use to overlap the exchange) gives the same rows
    eng3 = AlsEngine(n_users, n_items, dim, False, 0.1, device="cpu", kernels=NumpyKernels(), chunks=3)
    eng3.set_interactions(indptr, indices, values)
    assert len(eng3.csr_chunks["users"]) == 3 and sum(c.nnz for c in eng3.csr_chunks["users"]) == C.nnz
    eng3.set_factors("items", orc.init_items(n_items, dim))
    eng3.half_step("users")
    np.testing.assert_array_equal(eng3.get_factors("users"), eng.get_factors("users"))


def _worker_singular(rank, world, port, out_path):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from recmodel_amd import _lib
        from recmodel_amd.engine import AlsEngine
        eng = AlsEngine(8, 6, 3, False, 0.1, device="cpu", kernels=NumpyKernels(), chunks=1)
        if rank == 1:
            eng.fail[0] = 3                      # a singular row system seen by ONE rank only
        raised = False
        try:
            eng.check_numerics()
        except _lib.WmfNumericError:
            raised = True
        # a collective right after the check: with a per-rank decision rank 1 would have left and rank 0 would hang here
        t = torch.tensor([1.0 if raised else 0.0])
        dist.all_reduce(t)
        if rank == 0:
            np.save(out_path, np.array([float(t), float(eng.fail[0])]))
    finally:
        dist.destroy_process_group()


def test_numeric_failure_on_one_rank_raises_on_all(tmp_path):
    out = str(tmp_path / "flags.npy")
    mp.spawn(_worker_singular, args=(2, _free_port(), out), nprocs=2, join=True)
    got = np.load(out)
    assert got[0] == 2.0 and got[1] == 0.0       # both ranks raised; the flag was reset


def _worker_distributed_build(rank, world, port, bias, out_path, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine, row_cost
        n_users, n_items, dim = 211, 64, 5
        indptr, indices, counts = synth.make_counts(n_users, n_items, 6, seed=13, zipf_a=1.1)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        # this rank's block of user rows: contiguous, unequal (the last rank gets the remainder)
        per = n_users // world
        lo = rank * per
        hi = n_users if rank == world - 1 else lo + per
        e0, e1 = int(indptr[lo]), int(indptr[hi])
        kw = {"gather": {}, "reduce": {"reduce_mode": True}, "pipe": {"reduce_mode": False, "pipe_mode": True}}[mode]
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=3, **kw)
        eng.set_interactions_distributed(lo, indptr[lo: hi + 1] - e0, indices[e0:e1], values[e0:e1], balance=True)
        # the deal is explicit and cost-balanced, and every stored entry landed on exactly one rank, in both orientations
        assert eng.shard["users"].owner is not None and eng.shard["items"].owner is not None
        for side in ("users", "items"):
            t = torch.tensor([eng.csr[side].nnz], dtype=torch.int64)
            dist.all_reduce(t)
            assert int(t) == indices.numel()
        deg_i = torch.bincount(indices, minlength=n_items)
        cost = row_cost(deg_i, eng.f)
        mine = torch.tensor([float(cost[eng.shard["items"].owner.to(torch.int64) == rank].sum())], dtype=torch.float64)
        allc = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allc, mine)
        allc = torch.cat(allc)
        assert float((allc.max() - allc.min()) / allc.mean()) < 0.2          # (64 rows over 2-3 ranks: coarse, but not 2x)
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(2):
            eng.half_step("users")
            eng.half_step("items")
        eng.check_numerics()
        shard = eng.make_eval_shard_distributed(lo, indptr[lo: hi + 1] - e0, indices[e0:e1], counts[e0:e1])
        sq, ab, cnt = eng.eval_sums(shard)
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items, sums=np.array([sq, ab, cnt]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bias,mode", [(2, False, "gather"), (3, True, "gather"), (2, False, "reduce"), (3, False, "pipe"),
                                             (2, True, "reduce")])
def test_distributed_build_with_balanced_deal_matches_oracle(tmp_path, world, bias, mode):
    """Every rank hands over only ITS block of user rows (set_interactions_distributed): degrees are shared, the rows of
    both sides are dealt by cost (balanced_assignment, explicit owner / local-row maps instead of id % W), the entries
    travel to their owners, and two ALS iterations equal the single-process oracle on the whole matrix -- in all three
    exchange modes, with biases too."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker_distributed_build, args=(world, _free_port(), bias, out, mode), nprocs=world, join=True)
    got = np.load(out)
    n_users, n_items, dim = 211, 64, 5
    indptr, indices, counts = synth.make_counts(n_users, n_items, 6, seed=13, zipf_a=1.1)
    raw = synth.to_scipy(indptr, indices, counts, (n_users, n_items))
    C = raw.astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    for _ in range(2):
        users = step(items, C, 0.1)
        items = step(users, CT, 0.1)
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=2e-5)
    mse = orc.eval_prec(users, items, raw, bias)
    assert abs(got["sums"][0] / got["sums"][2] - mse) <= 1e-4 * mse and got["sums"][2] == raw.nnz


def _worker_negative_weights(rank, world, port, out_path, mode):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim = 90, 40, 4
        indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=3)
        kw = {"reduce": {"reduce_mode": True}, "pipe": {"reduce_mode": False, "pipe_mode": True}}[mode]
        eng = AlsEngine(n_users, n_items, dim, True, 0.1, device="cpu", kernels=NumpyKernels(), chunks=2, **kw)
        eng.set_interactions(indptr, indices, counts)                # raw counts 2 .. as weights
        items0 = _negative_weight_items(n_items, dim)
        eng.set_factors("items", items0)
        eng.half_step("users")                                       # all-gather mode for the users here (reduce: items only)
        eng.half_step("items")
        eng.check_numerics()
        users, items = eng.get_factors("users"), eng.get_factors("items")
        if rank == 0:
            np.savez(out_path, users=users, items=items)
    finally:
        dist.destroy_process_group()


def _negative_weight_items(n_items, dim):
    rng = np.random.default_rng(0)
    y = rng.standard_normal((n_items, dim + 1)).astype(np.float32) * 0.3
    y[:, 0] = 30.0                    # item biases far above every count: all bias-adjusted weights are negative
    return y


@pytest.mark.parametrize("mode", ["reduce", "pipe"])
def test_bias_model_in_reduce_and_pipe_mode_survives_negative_weights(tmp_path, mode):
    """A row whose summed partial system is not positive definite (bias-adjusted weights below zero) cannot be pivoted by
    its owner in reduce / pipelined mode; the engine notices (one flag, max-reduced) and does that half step again through
    the all-gather path, whose kernels pivot.  Result = the oracle's LU solve."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "neg.npz")
    mp.spawn(_worker_negative_weights, args=(2, _free_port(), out, mode), nprocs=2, join=True)
    got = np.load(out)
    n_users, n_items, dim = 90, 40, 4
    indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=3)
    C = synth.to_scipy(indptr, indices, counts, (n_users, n_items)).astype(np.float64)
    users = orc.recompute_factors_bias(_negative_weight_items(n_items, dim), C, 0.1)
    items = orc.recompute_factors_bias(users, C.T.tocsr(), 0.1)
    np.testing.assert_allclose(got["users"], users, rtol=5e-4, atol=5e-5)
    np.testing.assert_allclose(got["items"], items, rtol=5e-4, atol=5e-5)


# ------------------------------------------------------------------ need-list (sparse) gather of the users
def _sparse_matrix(n_users, n_items):
    """Users with 1 .. 3 items each, most of them inside one eighth of the catalogue: with items dealt over 4 or 8 ranks a
    rank's items are touched by a fraction of the users only."""
    from recmodel_amd import synth
    indptr, indices, counts = synth.make_counts(n_users, n_items, 2, seed=21)
    return indptr, indices, counts


def _worker_sparse(rank, world, port, bias, out_path, mode, distributed):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim = 331, 96, 5
        indptr, indices, counts = _sparse_matrix(n_users, n_items)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        kw = {"gather": {"pipe_mode": False}, "pipe": {"pipe_mode": True}}[mode]
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=3, reduce_mode=False, **kw)
        if distributed:
            per = n_users // world
            lo = rank * per
            hi = n_users if rank == world - 1 else lo + per
            e0, e1 = int(indptr[lo]), int(indptr[hi])
            eng.set_interactions_distributed(lo, indptr[lo: hi + 1] - e0, indices[e0:e1], values[e0:e1], balance=True)
        else:
            eng.set_interactions(indptr, indices, values)
        assert eng.sparse == {"users": True, "items": False}, eng.sparse
        # what this rank must send, recomputed from the matrix alone: a user row goes to every OTHER rank that owns one of the
        # user's items -- and nothing else travels
        sh_u, sh_i = eng.shard["users"], eng.shard["items"]
        rows = torch.repeat_interleave(torch.arange(n_users), indptr[1:] - indptr[:-1])
        owner_u, owner_i = sh_u.owner_of(rows), sh_i.owner_of(indices.to(torch.int64))
        pairs = torch.unique(torch.stack([rows[owner_u == rank], owner_i[owner_u == rank]], 1), dim=0)     # (my user, rank that needs it)
        want_rows = int((pairs[:, 1] != rank).sum())
        sent = eng.exchange_bytes_sent("users")
        assert sent == want_rows * eng.ld * 4, (sent, want_rows)
        dense = (world - 1) * eng.rpr["users"] * eng.ld * 4
        assert eng.exchange_bytes_sent("items") == (world - 1) * eng.rpr["items"] * eng.ld * 4
        eng.set_factors("items", orc.init_items(n_items, dim, bias))
        for _ in range(2):
            eng.half_step("users")
            eng.half_step("items")
        eng.check_numerics()
        shard = eng.make_eval_shard(indptr, indices, counts)
        sq, ab, cnt = eng.eval_sums(shard)
        users, items = eng.get_factors("users"), eng.get_factors("items")
        tot = torch.tensor([float(sent), float(dense)], dtype=torch.float64)
        dist.all_reduce(tot)
        if rank == 0:
            np.savez(out_path, users=users, items=items, sums=np.array([sq, ab, cnt]), bytes=tot.numpy())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,bias,mode,distributed", [(4, False, "gather", False), (8, True, "gather", False), (4, True, "pipe", False),
                                                         (8, False, "pipe", True), (4, False, "gather", True)])
def test_need_list_gather_of_the_users(tmp_path, world, bias, mode, distributed):
    """SURVEY.md 8(e) / north star: users shard over the ranks, and the item half step needs the rows of the users that touched
    this rank's items -- a fraction of them.  With the need-list gather every rank sends a solved user row only to the ranks
    that asked for it (one all_to_all per chunk into a compact gathered matrix the item-major CSR is re-indexed to); the
    bytes each rank sends are exactly those rows (recomputed from the matrix in the worker), well below the all-gather's,
    and two ALS iterations equal the single-process oracle -- plain and pipelined gather, round-robin and cost-balanced deal."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker_sparse, args=(world, _free_port(), bias, out, mode, distributed), nprocs=world, join=True)
    got = np.load(out)
    n_users, n_items, dim = 331, 96, 5
    indptr, indices, counts = _sparse_matrix(n_users, n_items)
    raw = synth.to_scipy(indptr, indices, counts, (n_users, n_items))
    C = raw.astype(np.float64)
    C.data = 10 * np.log(1 + C.data)
    CT = C.T.tocsr()
    items = orc.init_items(n_items, dim, bias)
    step = orc.recompute_factors_bias if bias else orc.recompute_factors
    for _ in range(2):
        users = step(items, C, 0.1)
        items = step(users, CT, 0.1)
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=2e-5)
    mse = orc.eval_prec(users, items, raw, bias)
    assert abs(got["sums"][0] / got["sums"][2] - mse) <= 1e-4 * mse and got["sums"][2] == raw.nnz
    sent, dense = got["bytes"]
    assert sent < 0.5 * dense, (sent, dense)              # ~2 items per user: a user row travels to ~1.6 of the W - 1 other ranks
