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


def _worker(rank, world, port, bias, out_path, reduce_mode=None, pipe_mode=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from fake_kernels import NumpyKernels
        from oracle import wmf_oracle as orc
        from recmodel_amd import synth
        from recmodel_amd.engine import AlsEngine
        n_users, n_items, dim = 203, 57, 6           # not multiples of the world size: padding rows exist
        indptr, indices, counts = synth.make_counts(n_users, n_items, 5, seed=11)
        values = (10 * torch.log(1 + counts)).to(torch.float32)
        eng = AlsEngine(n_users, n_items, dim, bias, 0.1, device="cpu", kernels=NumpyKernels(), chunks=4, reduce_mode=reduce_mode, pipe_mode=pipe_mode)
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
def test_two_rank_als_matches_single_process_oracle(tmp_path, bias, reduce_mode, pipe_mode):
    """reduce_mode=True: the item half step accumulates partial systems on every rank and reduce-scatters them
    (here an all-reduce + slice: gloo has no reduce-scatter); the user block is then only gathered on demand.
    pipe_mode=True: both half steps accumulate over each gathered chunk of the fixed side as it arrives."""
    from oracle import wmf_oracle as orc
    from recmodel_amd import synth
    out = str(tmp_path / "out.npz")
    mp.spawn(_worker, args=(2, _free_port(), bias, out, reduce_mode, pipe_mode), nprocs=2, join=True)
    got = np.load(out)
    n_users, n_items, dim = 203, 57, 6
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
    np.testing.assert_allclose(got["users"], users, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(got["items"], items, rtol=2e-4, atol=2e-5)
    mse = orc.eval_prec(users, items, raw, bias)
    assert abs(got["sums"][0] / got["sums"][2] - mse) <= 1e-4 * mse
    assert got["sums"][2] == raw.nnz


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
