"""``WMF``: weighted matrix factorisation by alternating least squares, on MI355X.

Same class surface as RecModel/wmf_model.py:8-351 -- constructor, ``train`` / ``predict`` /
``rank``, ``recompute_factors[_bias][_par]``, public ``users`` / ``items`` host arrays -- with the
numerical work done by libwmf_hip.so through ``AlsEngine``.  ``cores`` selected a multiprocessing
pool in the reference; what its Pool variants change numerically is the dtype -- float64 rows stacked
without a cast -- and that is what ``cores > 1`` selects here too (the float64 device path,
``wmf_half_step_f64``).  There is no CPU fallback.
"""
import ctypes
import time

import numpy as np
import scipy.sparse
import torch

from . import _lib
from .base_model import RecModel
from .engine import AlsEngine, _ptr, _stream


def _csr_parts(mat):
    mat = scipy.sparse.csr_matrix(mat) if not scipy.sparse.isspmatrix_csr(mat) else mat
    return (torch.from_numpy(mat.indptr.astype(np.int64)), torch.from_numpy(mat.indices.astype(np.int64)),
            torch.from_numpy(mat.data.astype(np.float32)))


# A float64 count matrix makes the reference solve every row system in float64 (NumPy's promotion at wmf_model.py:237-239).  The
# float32 row kernels reproduce that to the stated tolerance (5e-4 per row; measured 1e-5) while the confidence weights are of
# the benchmark's order; their error grows with the weights (cond(A_u) eps_f32: 1e-3 .. 6e-3 per row at weights of 1e6).  Above
# this weight a float64 count matrix therefore takes the float64 device path -- the reference's own arithmetic -- and the
# result is rounded to the model's dtype where the reference rounds it.
F64_ROW_WEIGHT = 1024.0


def _transformed_dtype(count_dtype, alpha, beta, pre_process_count):
    """dtype of the confidence weights the reference ends up with (wmf_model.py:119-123), by NumPy's own rules."""
    probe = np.ones(1, dtype=count_dtype)
    with np.errstate(all="ignore"):
        return (alpha * np.log(1 + beta * probe)).dtype if pre_process_count == 'log' else (alpha * probe).dtype


class _Float64Steps:
    """The two half steps of an iteration in float64 on the device (wmf_half_step_f64: RecModel/wmf_model.py:242-309) for
    `train(cores > 1)` on a float64 count matrix.  Factors live here as dense float64 device tensors; after every iteration
    float32 copies go to the engine, whose evaluation kernels compute the MSE train() stops on."""

    def __init__(self, model, eng, count_mat, alpha, beta, pre_process_count, store_float32=False):
        # store_float32: the reference's cores = 1 variants solve a float64 count matrix's rows in float64 and STORE float32
        # (wmf_model.py:217, :237-239) -- every half step's result is rounded to float32 before the next one reads it
        self.store_float32 = bool(store_float32)
        self.eng, self.K, self.bias, self.gamma = eng, eng.K, bool(model.bias), float(model.gamma)
        dev = eng.device
        C = scipy.sparse.csr_matrix(count_mat)
        self.csr = {}
        for side, mat in (("users", C), ("items", C.T.tocsr())):       # the transpose as the reference takes it (:128)
            vals = torch.from_numpy(np.ascontiguousarray(mat.data, dtype=np.float64)).to(dev)
            self.K.confidence_transform(vals, alpha, beta, 0 if pre_process_count == 'log' else 1)
            self.csr[side] = (torch.from_numpy(mat.indptr.astype(np.int64)).to(dev),
                              torch.from_numpy(mat.indices.astype(np.int32)).to(dev), vals, mat.shape[0])
        self.f = model.items.shape[1]
        self.X = {"items": torch.from_numpy(np.ascontiguousarray(model.items, dtype=np.float64)).to(dev), "users": None}
        n_max = max(C.shape)
        self.ws = torch.empty(self.K.half_step_f64_workspace_bytes(self.f, n_max, n_max), dtype=torch.uint8, device=dev)
        self.fail = torch.zeros(4, dtype=torch.int32, device=dev)

    def _half(self, side, fixed):
        indptr, indices, vals, n = self.csr[side]
        Y = self.X[fixed]
        out = torch.empty(n, self.f, dtype=torch.float64, device=Y.device)
        self.K.half_step_f64(Y, Y.shape[0], self.f, self.bias, indptr, indices, vals, n, self.gamma, out, self.ws, self.fail)
        self.X[side] = out.to(torch.float32).to(torch.float64) if self.store_float32 else out

    def iteration(self):
        self._half("users", "items")
        self._half("items", "users")
        fail = int(self.fail[0])
        if fail:
            self.fail.zero_()
            raise _lib.WmfNumericError(f"{fail} row systems were singular")
        for side in ("users", "items"):
            self.eng.set_factors(side, self.X[side].to(torch.float32))

    def factors(self):
        users, items = self.X["users"].cpu().numpy(), self.X["items"].cpu().numpy()
        return (users.astype(np.float32), items.astype(np.float32)) if self.store_float32 else (users, items)


class WMF(RecModel):

    def __init__(self, num_items, num_users, dim, gamma, weighted=None, bias=False, seed=1993, dtype='float32'):
        # wmf_model.py:10-23 -- the reference seeds the global legacy RNG and draws float64 uniforms
        np.random.seed(seed)
        self.bias = bias
        self.gamma = gamma
        if self.bias is False:
            self.items = np.random.random((num_items, dim)).astype(dtype=dtype)
        elif self.bias is True:
            self.items = np.random.random((num_items, (dim + 1))).astype(dtype=dtype)
        self.users = None
        self.num_users = num_users
        self.num_items = num_items
        self.dim = dim
        self.weighted = weighted
        self.dtype = dtype
        self._engine = None
        self._dev = None           # (key, users_t, items_t, f, ld): device copies of the public arrays for predict() / rank()

    # ------------------------------------------------------------------ engine plumbing
    def _new_engine(self):
        return AlsEngine(self.num_users, self.num_items, self.dim, self.bias is True, self.gamma)

    def _device_factors(self):
        """Device copies of the public host arrays.  The arrays ``train`` leaves behind are read-only (assign a new array
        to change the factors): their copies are cached by identity.  A writable array -- the constructor's ``items``, or
        anything the caller assigned -- may have been edited in place since the last call, which the reference would see
        (it reads the arrays live, wmf_model.py:205-211), so such an array is uploaded again on every call."""
        _lib.require_gpu()
        frozen = not (self.users.flags.writeable or self.items.flags.writeable)
        key = (id(self.users), id(self.items), self.users.shape, self.items.shape)
        if self._dev is None or self._dev[0] != key or not frozen:
            lib = _lib.load()
            f = self.items.shape[1]
            ld = int(lib.wmf_ld_for(f))

            def up(a):
                t = torch.zeros(a.shape[0], ld, dtype=torch.float32, device="cuda")
                t[:, :f] = torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))
                return t
            self._dev = (key, up(self.users), up(self.items), f, ld)
        return self._dev[1:]

    # ------------------------------------------------------------------ a7: predict
    def predict(self, users, items):
        """Scores for (user, item) pairs; one user or one item broadcasts.  wmf_model.py:191-211."""
        if (type(users) == list or type(users) == np.ndarray) and (type(items) == list or type(items) == np.ndarray):
            if len(users) != len(items):
                if not (len(users) == 1 or len(items) == 0):
                    raise ValueError("users and items need to have the same length or only one user / item needs to be provided.")
        u = np.atleast_1d(np.asarray(users)).astype(np.int32)
        i = np.atleast_1d(np.asarray(items)).astype(np.int32)
        if len(u) == 0 or len(i) == 0:
            return np.zeros(0, dtype=self.items.dtype)
        if len(u) != len(i) and len(i) == 1:
            i = np.repeat(i, len(u))
        if (u.min() < -self.users.shape[0] or u.max() >= self.users.shape[0]
                or i.min() < -self.items.shape[0] or i.max() >= self.items.shape[0]):
            raise IndexError("user or item index out of bounds")
        u = np.where(u < 0, u + self.users.shape[0], u).astype(np.int32)
        i = np.where(i < 0, i + self.items.shape[0], i).astype(np.int32)
        users_t, items_t, f, ld = self._device_factors()
        lib = _lib.load()
        ut, it = torch.from_numpy(u).cuda(), torch.from_numpy(i).cuda()
        out = torch.empty(max(len(u), len(i)), dtype=torch.float32, device="cuda")
        _lib.check(lib.wmf_predict_pairs(_ptr(users_t), _ptr(items_t), f, ld, int(self.bias is True), _ptr(ut), len(u),
                                         _ptr(it), len(i), _ptr(out), _stream()))
        return out.cpu().numpy().astype(self.users.dtype, copy=False)

    # ------------------------------------------------------------------ a10: rank
    def rank(self, items, users, topn=None):
        """Top-n of the candidate ``items`` for a user, best first.  wmf_model.py:25-47.
        Scores and ordering both happen on the device (wmf_rank_topn); equal scores keep candidate order."""
        if topn is None:
            topn = len(items)
        if not type(items) == np.ndarray:
            items = np.array(items)
        n = len(items)
        keep = min(int(topn), n)
        if isinstance(users, list):
            # the reference recurses user by user (wmf_model.py:29-32); here all of them in one device call per batch
            if keep <= 0 or len(users) == 0:
                return [items[:0] for _ in users]
            return self._rank_many(items, users, keep)
        if keep <= 0:
            return items[:0]
        u = int(np.asarray(users).reshape(-1)[0])
        idx = np.asarray(items).astype(np.int64)
        if not (-self.users.shape[0] <= u < self.users.shape[0]) or idx.min() < -self.items.shape[0] or idx.max() >= self.items.shape[0]:
            raise IndexError("user or item index out of bounds")
        idx = np.where(idx < 0, idx + self.items.shape[0], idx).astype(np.int32)
        users_t, items_t, f, ld = self._device_factors()
        lib = _lib.load()
        ut = torch.tensor([u % self.users.shape[0]], dtype=torch.int32, device="cuda")
        it = torch.from_numpy(idx).cuda()
        ws_bytes = int(lib.wmf_rank_workspace_bytes(n))
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
        pos = torch.empty(keep, dtype=torch.int32, device="cuda")
        _lib.check(lib.wmf_rank_topn(_ptr(users_t), _ptr(items_t), f, ld, int(self.bias is True), _ptr(ut), _ptr(it), n, keep,
                                     _ptr(pos), None, _ptr(ws), ws_bytes, _stream()))
        return items[pos.cpu().numpy()]

    def _rank_many(self, items, users, keep):
        """rank() for a list of users: scores of 16 x 16 (user, candidate) tiles by MFMA, one segmented sort
        (wmf_rank_topn_batch), in batches of at most 2^26 scores."""
        n = len(items)
        u = np.asarray(users).reshape(-1).astype(np.int64)
        idx = np.asarray(items).astype(np.int64)
        if (u.min() < -self.users.shape[0] or u.max() >= self.users.shape[0] or idx.min() < -self.items.shape[0]
                or idx.max() >= self.items.shape[0]):
            raise IndexError("user or item index out of bounds")
        u = np.where(u < 0, u + self.users.shape[0], u).astype(np.int32)
        idx = np.where(idx < 0, idx + self.items.shape[0], idx).astype(np.int32)
        users_t, items_t, f, ld = self._device_factors()
        lib = _lib.load()
        it = torch.from_numpy(idx).cuda()
        per = max(1, min(len(u), (1 << 26) // max(n, 1)))
        out = []
        ws = None
        for b0 in range(0, len(u), per):
            ub = torch.from_numpy(u[b0: b0 + per]).cuda()
            nu = ub.numel()
            need = int(lib.wmf_rank_batch_workspace_bytes(nu, n))
            if ws is None or ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device="cuda")
            pos = torch.empty(nu * keep, dtype=torch.int32, device="cuda")
            _lib.check(lib.wmf_rank_topn_batch(_ptr(users_t), _ptr(items_t), f, ld, int(self.bias is True), _ptr(ub), nu, _ptr(it), n,
                                               keep, _ptr(pos), None, _ptr(ws), ws.numel(), _stream()))
            ph = pos.cpu().numpy().reshape(nu, keep)
            out.extend(items[ph[j]] for j in range(nu))
        return out

    def _hit_counts(self, pair_user, pair_item, pair_row, candidates, slot, topn):
        """compute_hit (base_model.py:51-98) for every test entry in one launch: wmf_hit_counts."""
        users_t, items_t, f, ld = self._device_factors()
        lib = _lib.load()
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()   # noqa: E731
        pu, pi, pr, cd, sl, tn = dev(pair_user), dev(pair_item), dev(pair_row), dev(candidates), dev(slot), dev(topn)
        hits = torch.zeros(len(topn), dtype=torch.int64, device="cuda")
        _lib.check(lib.wmf_hit_counts(_ptr(users_t), _ptr(items_t), f, ld, int(self.bias is True), _ptr(pu), _ptr(pi), _ptr(pr),
                                      len(pair_user), _ptr(cd), candidates.shape[1], _ptr(sl), _ptr(tn), len(topn), _ptr(hits),
                                      _stream()))
        return hits.cpu().numpy()

    # ------------------------------------------------------------------ a8: eval_prec backend
    def _eval_sums(self, utility_mat):
        eng = self._engine
        if eng is None or not (eng.has_factors["users"] and eng.has_factors["items"]) or self._stale():
            eng = self._new_engine()
            eng.set_factors("users", self.users)
            eng.set_factors("items", self.items)
        shard = eng.make_eval_shard(*_csr_parts(utility_mat))
        return eng.eval_sums(shard)

    def _stale(self):
        """True when the public arrays are not (or may no longer be) what the engine holds."""
        if self.users is None or self.users.flags.writeable or self.items.flags.writeable:
            return True
        return getattr(self, "_synced", None) != (id(self.users), id(self.items))

    def _freeze(self):
        """The arrays handed out after training mirror device state: in-place edits would silently not reach predict /
        rank / eval_prec, so they are made read-only (an edit raises; assigning a fresh array works as in the reference)."""
        self.users.flags.writeable = False
        self.items.flags.writeable = False
        self._synced = (id(self.users), id(self.items))
        self._dev = None

    def _pull(self, eng, sides=("users", "items")):
        for s in sides:
            setattr(self, s, eng.get_factors(s).astype(self.dtype, copy=False))
        self._freeze()

    # ------------------------------------------------------------------ a3 / a4: operator seam
    def recompute_factors(self, Y, C, lambda_reg):
        """X_new for fixed factors Y and CSR C.  wmf_model.py:213-240 (host buffers in, host out)."""
        return self._recompute(Y, C, lambda_reg, 0)

    def recompute_factors_bias(self, Y, C, lambda_reg, cores=1):
        """Bias variant: column 0 of Y is the fixed side's bias.  wmf_model.py:311-351."""
        return self._recompute(Y, C, lambda_reg, 1)

    # ------------------------------------------------------------------ a5 / a6: the Pool variants
    def recompute_factors_par(self, Y, C, lambda_reg, cores=4):
        """wmf_model.py:242-250 (+ :289-309): the rows are independent, so `cores` has nothing to distribute here; what the
        variant changes is the dtype -- float64 rows stacked without a cast when Y or C is float64."""
        return self._recompute_par(Y, C, lambda_reg, 0)

    def recompute_factors_bias_par(self, Y, C, lambda_reg, cores=3):
        """wmf_model.py:252-265 (+ :267-287).  Column 0 of Y is the fixed side's bias (the caller's array is not modified)."""
        return self._recompute_par(Y, C, lambda_reg, 1)

    def _recompute_par(self, Y, C, lambda_reg, bias):
        C = scipy.sparse.csr_matrix(C)
        if np.result_type(np.asarray(Y).dtype, C.dtype) != np.float64:
            return self._recompute(Y, C, lambda_reg, bias)            # all-float32 inputs: float32 rows, as in the reference
        _lib.require_gpu()
        lib = _lib.load()
        Y = np.ascontiguousarray(Y, dtype=np.float64)
        indptr = np.ascontiguousarray(C.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(C.indices, dtype=np.int32)
        values = np.ascontiguousarray(C.data, dtype=np.float64)
        X = np.empty((C.shape[0], Y.shape[1]), dtype=np.float64)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        _lib.check(lib.wmf_recompute_factors_f64_host(vp(Y), Y.shape[0], Y.shape[1], bias, vp(indptr), vp(indices),
                                                      vp(values), C.shape[0], float(lambda_reg), vp(X)))
        return X

    def _recompute(self, Y, C, lambda_reg, bias):
        _lib.require_gpu()
        lib = _lib.load()
        C = scipy.sparse.csr_matrix(C)
        if (np.result_type(np.asarray(Y).dtype, C.dtype) == np.float64 and C.nnz
                and float(np.nanmax(np.abs(C.data))) > F64_ROW_WEIGHT):
            # float64 row systems in the reference (:237-239), and weights beyond what the float32 kernels hold the tolerance
            # for: the float64 device path, rounded to the model's dtype where the reference rounds (:217)
            return self._recompute_par(np.asarray(Y, dtype=np.float64), C.astype(np.float64), lambda_reg, bias).astype(self.dtype)
        Y = np.ascontiguousarray(Y, dtype=np.float32)
        indptr = np.ascontiguousarray(C.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(C.indices, dtype=np.int32)
        values = np.ascontiguousarray(C.data, dtype=np.float32)
        X = np.empty((C.shape[0], Y.shape[1]), dtype=np.float32)
        vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)  # noqa: E731
        _lib.check(lib.wmf_recompute_factors_host(vp(Y), Y.shape[0], Y.shape[1], bias, vp(indptr), vp(indices),
                                                  vp(values), C.shape[0], float(lambda_reg), vp(X)))
        return X.astype(self.dtype, copy=False)

    # ------------------------------------------------------------------ a2: train
    def train(self, utility_mat, iterations, verbose=0, eval_mat=None, count_mat=None, alpha=10,
              cores=4, stopping_rounds=3, dtype='float64', min_improvement=0.0001,
              pre_process_count='log', beta=1, preprocess_mat=False):
        """Alternating least squares with early stopping on eval MSE.  wmf_model.py:49-189.
        Returns the index of the last iteration run."""
        utility_mat = utility_mat.copy()
        if count_mat is not None:
            count_mat = count_mat.copy()
        if self.bias is True and self.weighted is False:
            print("Bias computation is only implemented for weighted matrix factorization.")
        if eval_mat is None and verbose > 1:
            print("Since no explicit evaluation was provided the train matrix is used for evaluation.")
            eval_mat = utility_mat
        if preprocess_mat == True:  # noqa: E712  (the reference compares with ==)
            if pre_process_count == 'log':
                utility_mat.data = alpha * np.log(1 + beta * utility_mat.data)
            elif pre_process_count == 'linear':
                utility_mat.data = alpha * utility_mat.data

        eng = self._engine = self._new_engine()
        eng.set_factors("items", self.items)
        last_mse = - np.inf
        count_improvement = 0

        if self.weighted is not True:
            # closed-form un-weighted branch, wmf_model.py:73-115
            if self.bias is True:
                raise ValueError("operands could not be broadcast together: the un-weighted branch has no bias column")
            eng.set_interactions(*_csr_parts(utility_mat))
            eval_shard = self._train_eval_shard(eng, eval_mat)
            train_shard = eng.make_eval_shard(*_csr_parts(utility_mat)) if verbose > 1 else None
            for iter in range(iterations):
                if verbose > 0:
                    print(f"Starting fitting iteration {iter}")
                eng.half_step_unweighted("users")
                eng.half_step_unweighted("items")
                eng.check_numerics()
                mse_eval = self._mse(eng, eval_shard)
                if verbose > 0:
                    print(f"Current eval mse is {mse_eval}")
                if verbose > 1:
                    print(f"\tMSE Eval: {mse_eval}")
                    print(f"\tMSE Train: {self._mse(eng, train_shard)}")
                if mse_eval * (1 + min_improvement) > last_mse:
                    count_improvement += 1
                else:
                    count_improvement = 0
                last_mse = mse_eval
                if count_improvement >= stopping_rounds:
                    break
            self._pull(eng)
            # (the reference's dense-times-sparse products, :85 / :88, take NumPy's result type of the model dtype and the
            # utility matrix's: float64 for SciPy's default float64 matrices, float32 for float32 ones)
            out_dt = np.result_type(np.dtype(self.dtype), utility_mat.dtype)
            self.users, self.items = self.users.astype(out_dt), self.items.astype(out_dt)
            self._freeze()
            if verbose > 0:
                print("Training was completed.")
            if verbose > 1:
                print(f"MSE Eval at iteration {iter}: {self._mse(eng, eval_shard)}")
                print(f"MSE Train at iteration {iter}: {self._mse(eng, train_shard)}")
            return iter

        # weighted branch, wmf_model.py:116-189
        if pre_process_count not in ('log', 'linear'):
            raise ValueError(f"Pre_process_count {pre_process_count} is not implement please use log or linear.")
        if not cores >= 1:
            raise ValueError(f"Values of cores has to be positive not {cores}")
        if self.bias is not True and self.bias is not False:
            raise ValueError(f"self.bias = {self.bias} is unknown. Only True / False are allowed.")
        # cores > 1 with a float64 OR INTEGER count matrix: the reference's Pool variants keep float64 rows (:242-265) -- the
        # confidence transform of integer counts is float64 ('log', :120) or int64 ('linear', :123; the row products then
        # promote) -- so its training continues on float64 factors: the float64 device path then does the half steps, the
        # engine only the MSE (and builds neither float32 shards nor row plans).  float32 / float16 counts stay float32.
        # Which arithmetic the reference's rows run in: NumPy's result type of the factors and the TRANSFORMED counts
        # (np.log of int16 is float32, of int64 float64; alpha * int64 stays int64 and promotes with the float32 factors).
        tdt = _transformed_dtype(count_mat.dtype, alpha, beta, pre_process_count)
        rows64 = np.result_type(np.dtype(self.dtype), tdt) == np.float64
        f64 = None
        if rows64 and (cores > 1 or np.dtype(self.dtype) == np.float64):
            # (float64 factors: the Pool variants stack float64 rows, :242-265; a float64 model stores what it solves, :217)
            f64 = _Float64Steps(self, eng, count_mat, alpha, beta, pre_process_count)
        elif rows64 and count_mat.nnz and self._max_weight(count_mat, alpha, beta, pre_process_count) > F64_ROW_WEIGHT:
            # cores = 1: float64 rows, float32 factors (:217); the float32 kernels hold the tolerance up to F64_ROW_WEIGHT only
            f64 = _Float64Steps(self, eng, count_mat, alpha, beta, pre_process_count, store_float32=True)
        else:
            indptr, indices, values = _csr_parts(count_mat)
            values = values.to(eng.device)
            eng.K.confidence_transform(values, alpha, beta, 0 if pre_process_count == 'log' else 1)
            eng.set_interactions(indptr, indices, values)      # also builds the item-major shard (:128)
        eval_shard = self._train_eval_shard(eng, eval_mat)
        train_shard = eng.make_eval_shard(*_csr_parts(utility_mat)) if verbose > 1 else None

        for iter in range(iterations):
            if verbose > 0:
                print(f"Starting fitting iteration {iter}")
            start = time.time()
            if f64 is not None:
                f64.iteration()
            else:
                eng.half_step("users")
                eng.half_step("items")
                eng.check_numerics()
            if self.bias is True and cores == 1:
                print(f"Iteration {iter} took {round(time.time() - start, 4)} seconds.")
            mse_eval = self._mse(eng, eval_shard)
            if mse_eval * (1 + min_improvement) > last_mse:
                count_improvement += 1
            else:
                count_improvement = 0
            last_mse = mse_eval
            if verbose > 0:
                print(f"Current eval mse is {mse_eval}")
            if verbose > 1:
                print(f"\tMSE Eval: {mse_eval}")
                print(f"\tMSE Train: {self._mse(eng, train_shard)}")
            if count_improvement >= stopping_rounds:
                break
        if f64 is not None:
            self.users, self.items = f64.factors()          # float64, as the reference's Pool variants leave them (float32 for cores = 1)
            self._freeze()
        else:
            self._pull(eng)
        if verbose > 0:
            print("Training was completed.")
        if verbose > 1:
            print(f"MSE Eval at iteration {iter}: {self._mse(eng, eval_shard)}")
            print(f"MSE Train at iteration {iter}: {self._mse(eng, train_shard)}")
        return iter

    @staticmethod
    def _max_weight(count_mat, alpha, beta, pre_process_count):
        """Largest confidence weight in magnitude the transform (wmf_model.py:119-123) gives this count matrix."""
        data = np.asarray(count_mat.data, dtype=np.float64)
        with np.errstate(all="ignore"):
            w = alpha * np.log(1 + beta * data) if pre_process_count == 'log' else alpha * data
        return float(np.nanmax(np.abs(w))) if w.size else 0.0

    @staticmethod
    def _train_eval_shard(eng, eval_mat):
        if eval_mat is None:
            # the reference fails inside eval_prec here (wmf_model.py:61-63 only substitutes when verbose > 1)
            raise AttributeError("'NoneType' object has no attribute 'nonzero'")
        return eng.make_eval_shard(*_csr_parts(eval_mat))

    @staticmethod
    def _mse(eng, shard):
        sq, _, cnt = eng.eval_sums(shard)
        return sq / cnt if cnt else float('nan')
