"""Device-resident WMF/ALS engine: the host side of the hot path above the C ABI.

One ``AlsEngine`` per process and per GPU.  Everything numerical is a call into libwmf_hip.so
(``include/wmf_hip.h``) through ``HipKernels``; torch supplies device memory, the HIP stream and --
for more than one GPU -- ``torch.distributed`` (backend ``nccl`` = RCCL over xGMI).  The package
has no CPU compute path: constructing an engine without a GPU raises.

Restates the control structure of ``WMF.train``'s weighted branch (RecModel/wmf_model.py:134-161):

    users = recompute_factors(items, C,  gamma)      -> half_step("users")
    items = recompute_factors(users, CT, gamma)      -> half_step("items")
    mse   = eval_prec(eval_mat)                      -> eval_sums(...)

Sharding.  Rows of the side being updated are independent given the
fixed side, so each rank owns a slice of the users and a slice of the items.  Ids are dealt
round-robin (id i lives on rank i % W at local index i // W) so that popularity-sorted ids give
every rank the same number of stored entries.  Per half step the only exchanges are an all-reduce
of the f x f Gramian (fp64) and the all-gather of the freshly solved factor block, and the gather is
hidden behind the solve: the local rows are solved in a few CHUNKS, and the all-gather of chunk c runs
on RCCL's stream while chunk c + 1 is being solved.  What travels is the un-whitened block (its Gramian,
hence the whitening matrix, only exists once every chunk is done); each rank then whitens the whole
gathered matrix itself -- a streaming pass at HBM speed, W times redundant, instead of an exposed
transfer over xGMI.  The gathered matrix is chunk-major so that every all-gather writes one contiguous
range: with local index j = i // W in chunk c = j // chunk_len (chunk start s_c, length len_c), the
*position* of id i is  W * s_c + (i % W) * len_c + (j - s_c).  With one rank this is the identity.

Reduce mode.  When the side being updated is small and the fixed side large (few items, many users), the all-gather
of the fixed side is the wrong exchange: a rank's block crosses each of its xGMI links whole.  Instead every rank
accumulates, for EVERY row of the small side, the partial normal equations contributed by its own slice of the fixed
side (C-ABI wmf_accumulate_rows), the partial systems are summed with a reduce-scatter -- chunk by chunk, the
exchange of one chunk behind the accumulation of the next -- and each rank eliminates the rows it owns
(wmf_eliminate_rows).  Per link that is rows_per_rank x (f(f+1)/2 + f) floats instead of rows_per_rank_of_the_fixed_side
x f; the engine picks the mode per side from exactly that comparison (REDUCE_GAIN), and a side whose only consumer runs
in reduce mode is not gathered at all until somebody asks for it (get_factors).

Need-list (sparse) gather of the users.  The item half step reads the row of every user that interacted with one of THIS
rank's items -- with 8 ranks and ~10 items per user that is 1 - (7/8)^10 = 74 % of the users, not all of them, and the
user block is what fills the xGMI links of a strong-scaling run (10 M x 129 floats: 0.66 GB per link and iteration at 8
GPUs, twice the time of the row solves).  When the fraction a rank needs is below SPARSE_MAX_FRACTION on every rank, each
rank therefore receives only those rows: at set-up every rank tells every other which of its local rows it needs (per
chunk), a chunk of freshly solved rows is packed per destination (index_select) and travels by ONE all_to_all_single with
split sizes instead of the all-gather, and lands in a COMPACT gathered matrix -- chunk-major, source-major, ascending local
row: exactly the order of the sorted needed positions, so the item-major CSR is re-indexed once and nothing is scattered
on arrival; the whitening pass runs over the compact matrix (the same fraction fewer rows).  Items stay dense (every rank
needs nearly all of them, and the evaluation reads them by id); get_factors gathers a sparse side densely on demand.

Pipelined gather mode.  When the fixed side does arrive by all-gather in several chunks and the rows being updated are
heavy (cfg2's items at 2 and 4 GPUs: hundreds of entries per row and chunk), the update does not wait for the last
chunk: the gathered matrix is chunk-major, so a row's entries are grouped by chunk; as soon as chunk c has landed and
been whitened, the rows' partial systems over THAT chunk's entries are accumulated into slot c of a
[rows, chunks, partial_row_floats] buffer (wmf_accumulate_rows with slot_stride = chunks), and after the last chunk
wmf_eliminate_rows adds a row's slots in order and solves.  Only the last chunk's accumulation and the elimination
are left once the transfer ends.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class _HeldWork:
    """An asynchronous collective together with the buffer it reads: whoever takes the work out of a pending list and waits
    on it later still holds the buffer until that wait."""

    def __init__(self, work, *buffers, landing=None):
        self.work, self.buffers, self.landing = work, buffers, landing     # landing: (device tensor, host tensor the exchange fills)

    def wait(self):
        self.work.wait()
        if self.landing is not None:
            self.landing[0].copy_(self.landing[1])
            self.landing = None
        self.buffers = ()


class HipKernels:
    """The compute backend: every method is one entry point of libwmf_hip.so (include/wmf_hip.h) on
    torch-owned device buffers and torch's current HIP stream.  This is the only backend the package
    ships; the class exists so the host logic above it (sharding, collectives, control flow) can be
    exercised on CPU ranks in tests with a stand-in that has the same methods."""

    def __init__(self):
        _lib.require_gpu()
        self.lib = _lib.load()

    def ld_for(self, f):
        return int(self.lib.wmf_ld_for(f))

    def whitened_row_floats(self, f, ld, bias):
        return int(self.lib.wmf_whitened_row_floats(f, ld, int(bias)))

    def gram_workspace_bytes(self, f):
        return int(self.lib.wmf_gram_workspace_bytes(f))

    def eval_workspace_bytes(self):
        return int(self.lib.wmf_eval_workspace_bytes())

    def gram(self, Y, m, f, ld, bias, G, ws):
        _lib.check(self.lib.wmf_gram(_ptr(Y), m, f, ld, int(bias), _ptr(G), _ptr(ws), _stream()))

    def factorize(self, G, f, ld, lam, W_white, W_unwhite, info, ws):
        _lib.check(self.lib.wmf_factorize(_ptr(G), f, ld, float(lam), _ptr(W_white), _ptr(W_unwhite), _ptr(info), _ptr(ws),
                                          _stream()))

    def row_transform(self, inp, m, f, ld, W, set_col0_one, out, col0_out):
        _lib.check(self.lib.wmf_row_transform(_ptr(inp), m, f, ld, _ptr(W), int(set_col0_one), _ptr(out), _ptr(col0_out),
                                              _stream()))

    def plan_create(self, indptr_host, n, f, bias):
        handle = ctypes.c_void_p()
        _lib.check(self.lib.wmf_plan_create(indptr_host.ctypes.data_as(ctypes.c_void_p), n, f, int(bool(bias)), ctypes.byref(handle)))
        stats = np.zeros(14, dtype=np.int64)
        _lib.check(self.lib.wmf_plan_stats(handle, stats.ctypes.data_as(ctypes.c_void_p)))
        return handle, stats

    def plan_iter_stats(self, handle):
        """(rows solved by the matrix-free iteration, rows it handed back to the elimination kernels, applications of the row
        operator, rows on the Chebyshev recurrence) since the last call; synchronises the device."""
        out = np.zeros(4, dtype=np.int64)
        _lib.check(self.lib.wmf_plan_iter_stats(handle, out.ctypes.data_as(ctypes.c_void_p)))
        return out

    def plan_destroy(self, handle):
        self.lib.wmf_plan_destroy(handle)

    def solve_rows(self, plan, V, bias_vec, indptr, indices, values, n, f, ld, g, fail, flags=0):
        _lib.check(self.lib.wmf_solve_rows_ex(plan, _ptr(V), _ptr(bias_vec), _ptr(indptr), _ptr(indices), _ptr(values), n, f, ld,
                                              _ptr(g), _ptr(fail), int(flags), _stream()))

    def rolled_layout_supported(self, f, ld):
        return bool(self.lib.wmf_rolled_layout_supported(f, ld))

    def partial_row_floats(self, f):
        return int(self.lib.wmf_partial_row_floats(f))

    def accumulate_rows(self, V, bias_vec, indptr, degrees, indices, values, n, nnz, f, ld, partial, w_eff, slot_stride=1,
                        slot_offset=0):
        _lib.check(self.lib.wmf_accumulate_rows(_ptr(V), _ptr(bias_vec), _ptr(indptr), _ptr(degrees), _ptr(indices), _ptr(values),
                                                n, nnz, f, ld, _ptr(partial), slot_stride, slot_offset, _ptr(w_eff), _stream()))

    def eliminate_rows(self, partial, n, f, ld, g, fail, scratch, slots_per_row=1):
        _lib.check(self.lib.wmf_eliminate_rows(_ptr(partial), n, slots_per_row, f, ld, _ptr(g), _ptr(fail), _ptr(scratch),
                                               _stream()))

    def gather_rows(self, src, rows):
        """src[rows] as a new contiguous tensor (the rows of a solved block packed per destination: wmf_gather_rows)."""
        out = torch.empty(rows.numel(), src.shape[1], dtype=src.dtype, device=src.device)
        _lib.check(self.lib.wmf_gather_rows(_ptr(src), src.shape[1], _ptr(rows), rows.numel(), _ptr(out), _stream()))
        return out

    def spmm_rows(self, V, indptr, indices, values, n, f, ld, g):
        _lib.check(self.lib.wmf_spmm_rows(_ptr(V), _ptr(indptr), _ptr(indices), _ptr(values), n, f, ld, _ptr(g), _stream()))

    def eval_sqerr(self, users, items, f, ld, bias, indptr, indices, values, n, out3, ws):
        _lib.check(self.lib.wmf_eval_sqerr(_ptr(users), _ptr(items), f, ld, int(bias), _ptr(indptr), _ptr(indices),
                                           _ptr(values), n, _ptr(out3), _ptr(ws), _stream()))

    def coo_to_csr(self, rows, cols, vals, n_rows, n_cols):
        """(indptr int64, indices int32, values) of the entries sorted stably by (row, column): wmf_coo_to_csr."""
        nnz = rows.numel()
        dev = rows.device
        indptr = torch.empty(n_rows + 1, dtype=torch.int64, device=dev)
        indices = torch.empty(nnz, dtype=torch.int32, device=dev)
        values = torch.empty(nnz, dtype=torch.float32, device=dev)
        bad = torch.zeros(4, dtype=torch.int32, device=dev)
        ws = torch.empty(int(self.lib.wmf_coo_to_csr_workspace_bytes(nnz, n_rows, n_cols)), dtype=torch.uint8, device=dev)
        _lib.check(self.lib.wmf_coo_to_csr(_ptr(rows), _ptr(cols), _ptr(vals), nnz, n_rows, n_cols, _ptr(indptr), _ptr(indices),
                                           _ptr(values), _ptr(bad), _ptr(ws), ws.numel(), _stream()))
        if int(bad[0]):
            raise IndexError(f"an entry lies outside the {n_rows} x {n_cols} matrix")
        return indptr, indices, values

    def confidence_transform(self, values, alpha, beta, mode):
        if values.dtype == torch.float64:
            _lib.check(self.lib.wmf_confidence_transform_f64(_ptr(values), values.numel(), float(alpha), float(beta), int(mode),
                                                             _stream()))
            return
        _lib.check(self.lib.wmf_confidence_transform(_ptr(values), values.numel(), float(alpha), float(beta), int(mode),
                                                     _stream()))

    def half_step_f64_workspace_bytes(self, f, m, n):
        return int(self.lib.wmf_half_step_f64_workspace_bytes(f, m, n))

    def half_step_f64(self, Y, m, f, bias, indptr, indices, values, n, lam, X, ws, fail):
        """One half step in float64 (the reference's cores > 1 variants, wmf_model.py:242-309): dense float64 [m, f] in,
        dense float64 [n, f] out."""
        _lib.check(self.lib.wmf_half_step_f64(_ptr(Y), m, f, int(bias), _ptr(indptr), _ptr(indices), _ptr(values), n, float(lam),
                                              _ptr(X), _ptr(ws), ws.numel(), _ptr(fail), _stream()))


class Csr:
    """CSR shard resident in HBM (int64 indptr, int32 indices, fp32 values) plus its row plan."""

    def __init__(self, kernels, indptr, indices, values, n_cols, f, bias=False, plan=True):
        """plan=False: a shard that is only read by eval_sums (no row solve): no row plan, none of its device buffers."""
        self.kernels = kernels
        self.indptr = indptr.to(torch.int64).contiguous()
        self.indices = indices.to(torch.int32).contiguous()
        self.values = values.to(torch.float32).contiguous()
        self.n_rows = self.indptr.numel() - 1
        self.n_cols = int(n_cols)
        self.nnz = int(self.indices.numel())
        self.f = f
        if not plan:
            self._plan = None
            return
        host_ptr = np.ascontiguousarray(self.indptr.cpu().numpy(), dtype=np.int64)
        self._plan, stats = kernels.plan_create(host_ptr, self.n_rows, f, bias)
        self.bin_rows, self.bin_nnz = stats[:4].copy(), stats[4:8].copy()
        self.rows8, self.nnz8 = int(stats[8]), int(stats[9])           # rows of bin 0 with at most 8 entries
        self.rows_split, self.nnz_split = int(stats[10]), int(stats[11])   # rows of bin 2 split over several waves
        # rows (of bin 2 / 3) short enough for the matrix-free iteration kernel, csrc/wmf_iter.hip
        self.rows_iter, self.nnz_iter = (int(stats[12]), int(stats[13])) if len(stats) > 13 else (0, 0)

    def iter_stats(self):
        """What the iteration kernel did with its candidates since the last call (see HipKernels.plan_iter_stats); zeros for
        a back end without it."""
        if self._plan is None or not hasattr(self.kernels, "plan_iter_stats"):
            return np.zeros(4, dtype=np.int64)
        return self.kernels.plan_iter_stats(self._plan)

    def __del__(self):
        plan, self._plan = getattr(self, "_plan", None), None
        if plan:
            try:
                self.kernels.plan_destroy(plan)
            except Exception:
                pass


def coo_to_csr(rows, cols, vals, n_rows, kernels=None, n_cols=None):
    """Sort COO entries by (row, col) and build indptr.  Stable: duplicates survive.  On the device this is the library's
    wmf_coo_to_csr (kernels = HipKernels); the torch formulation below serves the CPU rehearsal of the host logic
    (tests/fake_kernels.py has no such method)."""
    if kernels is not None and hasattr(kernels, "coo_to_csr"):
        if n_cols is None:
            n_cols = int(cols.max().item()) + 1 if cols.numel() else 1
        return kernels.coo_to_csr(rows.to(torch.int64).contiguous(), cols.to(torch.int64).contiguous(),
                                  vals.to(torch.float32).contiguous(), int(n_rows), int(n_cols))
    n_cols_bound = int(cols.max().item()) + 1 if cols.numel() else 1
    key = rows.to(torch.int64) * n_cols_bound + cols.to(torch.int64)
    order = torch.argsort(key, stable=True)
    rows_s = rows[order]
    counts = torch.bincount(rows_s, minlength=n_rows)
    if counts.numel() != n_rows:          # a row id >= n_rows: cumsum(out=) below would silently resize indptr
        raise IndexError(f"row index {int(rows_s.max().item())} is out of bounds for {n_rows} rows")
    indptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(counts, 0, out=indptr[1:])
    return indptr, cols[order].to(torch.int32), vals[order]


MIN_CHUNK_ROWS = 32768          # default chunking never makes chunks smaller than this
REDUCE_GAIN = 0.8               # reduce mode when its bytes per link are below this fraction of the all-gather's
PIPE_MIN_ENTRIES = 48           # pipelined gather mode only when a row has at least this many entries per arriving chunk
SPARSE_MAX_FRACTION = 0.85      # need-list gather of the users when no rank needs more than this fraction of them


class Sharding:
    """Which rank owns an id of one side, and at which local row.  ``owner`` / ``local`` are None for the closed-form
    round-robin deal (id i on rank i % W at local row i // W) or device tensors [n] for an explicit assignment
    (``balanced_assignment``)."""

    def __init__(self, n, world, rank, owner=None, local=None, counts=None):
        self.n, self.world, self.rank = int(n), int(world), int(rank)
        self.owner, self.local = owner, local
        if owner is None:
            self.counts = [len(range(r, self.n, self.world)) for r in range(self.world)]
        else:
            self.counts = [int(c) for c in counts]
        self.n_local = self.counts[self.rank]
        self.rpr = max(1, max(self.counts)) if self.n else 1          # rows per rank, padded to the largest block

    def owner_of(self, ids):
        return ids % self.world if self.owner is None else self.owner[ids].to(torch.int64)

    def local_of(self, ids):
        return ids // self.world if self.owner is None else self.local[ids].to(torch.int64)

    def my_ids(self, device):
        """ids of this rank's rows in local-row order."""
        if self.owner is None:
            return torch.arange(self.rank, self.n, self.world, device=device)
        mine = torch.nonzero(self.owner.to(device) == self.rank).flatten()
        return mine[torch.argsort(self.local.to(device)[mine])]


def balanced_assignment(cost, world, heavy_per_rank=64):
    """Deal rows over ``world`` ranks so that the summed cost per rank is equal -- SURVEY.md 8(e): rows cost
    a * nnz * f^2 + b * f^3, and with power-law degrees equal row COUNTS are far from equal work.  Deterministic (every
    rank computes the same deal from the same cost vector):
      1. the heaviest rows (64 per rank) go, in descending cost, to the rank with the least cost so far (LPT);
      2. of the remaining rows, taken in descending cost, each rank first receives a run that lifts it to the level of the
         fullest rank, and the rest is dealt in serpentine order (0 .. W-1, W-1 .. 0), which keeps both the cost and the
         row count of the ranks equal -- equal blocks are what all_gather_into_tensor moves without padding.
    Returns (owner int32 [n], local int32 [n], counts [world]); local rows follow the ids within a rank."""
    n = cost.numel()
    dev = cost.device
    c = cost.to(torch.float64)
    order = torch.argsort(c, descending=True, stable=True)
    cs = c[order]
    owner_sorted = torch.empty(n, dtype=torch.int64, device=dev)
    k = min(n, heavy_per_rank * world)
    load = [0.0] * world
    head = cs[:k].cpu().tolist()
    head_owner = []
    for v in head:                                           # LPT on the host: k is a few hundred
        r = min(range(world), key=lambda i: (load[i], i))
        load[r] += v
        head_owner.append(r)
    owner_sorted[:k] = torch.tensor(head_owner, dtype=torch.int64, device=dev)
    if n > k:
        tail = cs[k:]
        cum = torch.cumsum(tail, 0)
        need = [max(load) - x for x in load]                 # cost that lifts a rank to the fullest one
        bounds, acc_need = [0], 0.0
        for r in range(world):
            acc_need += need[r]
            bounds.append(int(torch.searchsorted(cum, torch.tensor(acc_need, dtype=torch.float64, device=dev), right=True)))
        bounds = [min(b, n - k) for b in bounds]
        t_owner = torch.empty(n - k, dtype=torch.int64, device=dev)
        for r in range(world):
            t_owner[bounds[r]: bounds[r + 1]] = r
        rest = n - k - bounds[-1]
        if rest > 0:
            i = torch.arange(rest, device=dev)
            lap, pos = i // world, i % world
            t_owner[bounds[-1]:] = torch.where(lap % 2 == 0, pos, world - 1 - pos)
        owner_sorted[k:] = t_owner
    owner = torch.empty(n, dtype=torch.int64, device=dev)
    owner[order] = owner_sorted
    counts = torch.bincount(owner, minlength=world)
    by_rank = torch.argsort(owner, stable=True)              # ids grouped by rank, ascending id inside a rank
    start = torch.cumsum(counts, 0) - counts
    local = torch.empty(n, dtype=torch.int64, device=dev)
    local[by_rank] = torch.arange(n, device=dev) - start[owner[by_rank]]
    return owner.to(torch.int32), local.to(torch.int32), counts.cpu().tolist()


def row_cost(degrees, f):
    """Work of one row update, SURVEY.md 8(e): nnz f^2 (accumulating its system) + f^3 (eliminating it)."""
    return degrees.to(torch.float64) * float(f * f) + float(f) ** 3


def gathered_positions(ids, world, rows_per_rank, chunk_len, owner=None, local=None):
    """Row of id ``ids`` in a chunk-major gathered matrix (module docstring).  Works on ints and tensors; ``owner`` /
    ``local`` give the rank and local row of every id (default: the round-robin deal)."""
    j = ids // world if local is None else local
    r = ids % world if owner is None else owner
    s_c = (j // chunk_len) * chunk_len
    rest = rows_per_rank - s_c
    len_c = torch.clamp(rest, max=chunk_len) if torch.is_tensor(rest) else min(chunk_len, rest)
    return world * s_c + r * len_c + (j - s_c)


class AlsEngine:
    """Weighted-ALS state of one rank: factor blocks, whitened gathers, CSR shards."""

    def __init__(self, n_users, n_items, dim, bias, gamma, device=None, group=None, kernels=None, chunks=None,
                 reduce_mode=None, pipe_mode=None, force_exchange=None, sparse_mode=None):
        self.K = kernels if kernels is not None else HipKernels()     # raises without a GPU / built library
        self.lib = getattr(self.K, "lib", None)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.group = group
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
            self.rank = torch.distributed.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        # the exchange machinery (separate gathered matrices, chunks, collectives) is what more than one rank uses; a
        # single rank can be made to go through it as well (force_exchange / WMF_FORCE_EXCHANGE=1) so that the very
        # RCCL calls of the multi-GPU path -- asynchronous all-gathers and reduce-scatters on RCCL's stream, ordered
        # against the kernels by work.wait() -- can be rehearsed on a one-GPU box
        if force_exchange is None:
            force_exchange = os.environ.get("WMF_FORCE_EXCHANGE") == "1"
        self.exchange = self.world > 1 or (bool(force_exchange) and torch.distributed.is_available()
                                           and torch.distributed.is_initialized())
        self.n = {"users": int(n_users), "items": int(n_items)}
        self.dim, self.bias, self.gamma = int(dim), bool(bias), float(gamma)
        self.f = self.dim + 1 if self.bias else self.dim
        self.ld = self.K.ld_for(self.f)
        self.pr = self.K.partial_row_floats(self.f) if hasattr(self.K, "partial_row_floats") else 0
        self._reduce_arg, self.pipe_mode, self._chunks_arg, self._sparse_arg = reduce_mode, pipe_mode, chunks, sparse_mode
        self.csr = {}          # "users": shard with local user rows, "items": shard with local item rows
        self.csr_chunks = {}   # the same rows as one Csr (own row plan) per chunk
        self._configure({s: Sharding(self.n[s], self.world, self.rank) for s in self.n})

    def _configure(self, shard):
        """Everything that follows from who owns which row: block sizes, exchange mode per side, chunking, buffers.
        Called with the round-robin deal by the constructor and again by set_interactions*(balance=True) once the
        degrees are known (factors set before that are dropped: load them after the interactions)."""
        W = self.world
        self.shard = shard
        self.rpr = {s: shard[s].rpr for s in self.n}                          # rows per rank (padded to the largest block)
        self.n_local = {s: shard[s].n_local for s in self.n}
        pr, chunks = self.pr, self._chunks_arg
        # exchange mode per updated side (module docstring): "reduce" = accumulate everywhere + reduce-scatter, else gather
        forced = os.environ.get("WMF_REDUCE") if self._reduce_arg is None else ("1" if self._reduce_arg else "0")
        self.reduce = {}
        for s in self.n:
            o = "items" if s == "users" else "users"
            gain = W > 1 and pr > 0 and self.rpr[s] * pr < REDUCE_GAIN * self.rpr[o] * self.ld
            # A summed system that is not positive definite cannot be handed to the pivoted kernel by its owner (no CSR
            # there); only bias-adjusted weights can go negative.  Bias models use the mode all the same: a half step that
            # counted such a row is done again through the gather path (_redo_through_gather).
            self.reduce[s] = self.exchange and pr > 0 and (forced == "1" or (forced is None and gain))
        if self.reduce["users"] and self.reduce["items"]:
            # both at once would leave nobody holding a whole side; keep the one that saves more
            keep = "items" if self.rpr["items"] <= self.rpr["users"] else "users"
            self.reduce = {s: s == keep for s in self.n}
        auto_chunks = chunks is None and "WMF_CHUNKS" not in os.environ
        if chunks is None:
            chunks = int(os.environ.get("WMF_CHUNKS", "4")) if self.exchange else 1
        # chunk c of a side = local rows [c * chunk_len, min((c + 1) * chunk_len, rows_per_rank)).  A side whose block is
        # small is not cut when the chunk count is the default: its gather is cheap, and a few thousand heavy rows per
        # launch would leave most of the 3000 resident waves of the row kernels idle in the last round.
        want = {s: max(1, int(chunks)) for s in self.n}
        if auto_chunks:
            # a side updated in reduce mode is accumulated over ALL its rows on every rank: W times the rows per chunk
            want = {s: max(1, min(want[s], (self.rpr[s] * (W if self.reduce[s] else 1)) // MIN_CHUNK_ROWS)) for s in self.n}
            for s in self.n:                         # nobody gathers a side whose only consumer runs in reduce mode:
                if self.reduce["items" if s == "users" else "users"]:      # chunking it would only cost launches
                    want[s] = 1
        self.chunk_len = {s: max(1, (self.rpr[s] + want[s] - 1) // want[s]) for s in self.n}
        self.chunk_bounds = {s: [(lo, min(self.chunk_len[s], self.rpr[s] - lo))
                                 for lo in range(0, max(self.rpr[s], 1), self.chunk_len[s]) if lo < self.rpr[s]]
                             for s in self.n}
        # rows [start, stop) of chunk c in the GATHERED matrix of a side (dense gather: W x the chunk; a need-list gather
        # replaces them by the compact ranges, _setup_sparse)
        self.chunk_range = {s: [(W * lo, W * (lo + ln)) for lo, ln in self.chunk_bounds[s]] for s in self.n}
        self.sparse = {s: False for s in self.n}
        self.need = {}
        dev, f32 = self.device, torch.float32
        z = lambda *shape, dtype=f32: torch.zeros(*shape, dtype=dtype, device=dev)  # noqa: E731
        # local factor blocks [rows_per_rank, ld]; rows >= n_local are padding and stay zero
        self.factors = {s: z(self.rpr[s], self.ld) for s in self.n}
        self.g = {s: z(self.rpr[s], self.ld) for s in self.n}
        # gathered factors of all ranks (chunk-major positions); with one rank the local block itself
        self.X = {s: (z(W * self.rpr[s], self.ld) if self.exchange else self.factors[s]) for s in self.n}
        self._pending = {s: [] for s in self.n}                            # all-gathers in flight
        # whitened gathered factors / bias of the fixed side [W * rows_per_rank, ld] -- or, where the library writes its
        # split layout (include/wmf_hip.h, wmf_row_transform: k = 16 m with biases), a packed body [.., f - 1] and the
        # {last feature, bias} pairs [.., 2]; either way row ranges of the two tensors are what the kernels are given
        self.ldv = self.K.whitened_row_floats(self.f, self.ld, self.bias)
        self.split = self.ldv != self.ld
        # rolled whitened coordinates with the bias in the body's last mantissa bits (include/wmf_hip.h, wmf_row_transform modes
        # 3 / 4): the iteration kernels then fetch nothing from the pairs.  WMF_ROLLED=0 keeps the plain split layout.
        self.rolled = bool(self.split and self.bias and os.environ.get("WMF_ROLLED", "1") != "0"
                           and self.K.rolled_layout_supported(self.f, self.ld))
        self.white_mode = 3 if self.rolled else self.bias           # set_col0_one of the whitening / of the un-whitening
        self.unwhite_mode = 4 if self.rolled else False
        self.solve_flags = 1 if self.rolled else 0
        self.V = {s: z(W * self.rpr[s], self.ldv) for s in self.n}
        self.bias_vec = {s: (z(W * self.rpr[s], 2) if self.split else z(W * self.rpr[s])) for s in self.n}
        self.G = z(self.f * self.f, dtype=torch.float64)
        self.W_white, self.W_unwhite = z(self.f, self.ld), z(self.f, self.ld)
        self.info = z(4, dtype=torch.int32)
        self.fail = z(4, dtype=torch.int32)
        self.ws = torch.empty(self.K.gram_workspace_bytes(self.f), dtype=torch.uint8, device=dev)
        self.eval_ws = torch.empty(self.K.eval_workspace_bytes(), dtype=torch.uint8, device=dev)
        self.eval_out = z(3, dtype=torch.float64)
        self.pipe = {s: False for s in self.n}               # pipelined gather mode, decided in set_interactions (needs the degrees)
        self.csr_pipe = {}     # per chunk of the fixed side: (indptr, degrees, indices, values, w_eff) over this rank's rows
        self.partial_pipe = {}
        self.csr_red = {}      # reduce mode: the side's matrix over ALL its rows x this rank's rows of the fixed side
        self.partial_all = {s: (torch.empty(W * self.rpr[s], pr, dtype=f32, device=dev) if self.reduce[s] else None) for s in self.n}
        self.partial_mine = {s: (torch.empty(self.rpr[s], pr, dtype=f32, device=dev) if self.reduce[s] else None) for s in self.n}
        self.scratch_rows = torch.zeros(max(self.rpr.values()), dtype=torch.int32, device=dev)
        self._stale = {s: False for s in self.n}                              # X[s] not gathered since the last update
        self.csr, self.csr_chunks = {}, {}
        self.has_factors = {"users": False, "items": False}

    # ---------------------------------------------------------------- id <-> position maps
    def positions(self, side, ids):
        sh = self.shard[side]
        if sh.owner is None:
            return gathered_positions(ids, self.world, self.rpr[side], self.chunk_len[side])
        return gathered_positions(ids, self.world, self.rpr[side], self.chunk_len[side], sh.owner_of(ids), sh.local_of(ids))

    def _other(self, side):
        return "items" if side == "users" else "users"

    # ---------------------------------------------------------------- data
    def _check_csr(self, indptr, cols, what):
        """The reference indexes the fixed factors with the stored column ids (wmf_model.py:233 ``Y[idx, :]``) and raises
        IndexError for an id outside them; here the ids become gather positions of device kernels, so they are checked once,
        when the matrix is handed over."""
        if indptr.numel() != self.n["users"] + 1:
            raise ValueError(f"{what}: {indptr.numel() - 1} rows, but the model has num_users = {self.n['users']}")
        if int(indptr[-1]) != cols.numel() or (indptr.numel() > 1 and bool((indptr[1:] < indptr[:-1]).any())):
            raise ValueError(f"{what}: indptr is not a monotone row pointer over {cols.numel()} stored entries")
        if cols.numel():
            lo, hi = int(cols.min()), int(cols.max())
            if lo < 0 or hi >= self.n["items"]:
                raise IndexError(f"{what}: column index {hi if hi >= self.n['items'] else lo} is out of bounds for "
                                 f"num_items = {self.n['items']}")

    def set_interactions(self, indptr, indices, values, balance=False):
        """Full user-major confidence matrix (device tensors, CSR as stored), the same on every rank.  Builds this
        rank's user-row shard and item-row shard (the transpose is taken here, on the device; wmf_model.py:128
        ``count_mat.T.tocsr()``).  balance=True deals the rows of both sides by cost (balanced_assignment) instead of
        round-robin; factors loaded before are dropped then."""
        dev = self.device
        indptr = indptr.to(dev, torch.int64)
        cols = indices.to(dev, torch.int64)
        vals = values.to(dev, torch.float32)
        self._check_csr(indptr, cols, "count_mat")
        counts = indptr[1:] - indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(self.n["users"], device=dev), counts)
        if balance and self.world > 1:
            self._balance(counts, torch.bincount(cols, minlength=self.n["items"]))
        mine = {}
        for side, (r_, c_) in (("users", (rows, cols)), ("items", (cols, rows))):
            if self.world > 1:
                m = self.shard[side].owner_of(r_) == self.rank
                mine[side] = (r_[m], c_[m], vals[m])
            else:
                mine[side] = (r_, c_, vals)
        self._build_shards(mine)

    def set_interactions_distributed(self, first_user, indptr, indices, values, balance=True):
        """The same from a matrix that is itself distributed: this rank holds the user rows
        [first_user, first_user + len(indptr) - 1) of the user-major matrix and nothing else (a partition of a file, a
        generator's block).  Every stored entry travels once to the owner of its user and once to the owner of its item
        (all_to_all over the ranks); no rank ever sees the whole matrix.  The row degrees -- one number per row -- are
        shared so that every rank computes the same cost-balanced deal."""
        dev = self.device
        indptr = indptr.to(dev, torch.int64)
        cols = indices.to(dev, torch.int64)
        vals = values.to(dev, torch.float32)
        n_rows = indptr.numel() - 1
        if int(indptr[-1]) != cols.numel() or (n_rows > 0 and bool((indptr[1:] < indptr[:-1]).any())):
            raise ValueError("count_mat block: indptr is not a monotone row pointer over the stored entries")
        if first_user < 0 or first_user + n_rows > self.n["users"]:
            raise ValueError(f"count_mat block: user rows [{first_user}, {first_user + n_rows}) outside num_users = {self.n['users']}")
        if cols.numel() and (int(cols.min()) < 0 or int(cols.max()) >= self.n["items"]):
            raise IndexError(f"count_mat block: a column index is out of bounds for num_items = {self.n['items']}")
        counts = indptr[1:] - indptr[:-1]
        rows = first_user + torch.repeat_interleave(torch.arange(n_rows, device=dev), counts)
        if balance and self.world > 1:
            deg_u = torch.zeros(self.n["users"], dtype=torch.int64, device=dev)
            deg_u[first_user: first_user + n_rows] = counts
            deg_i = torch.bincount(cols, minlength=self.n["items"])
            if self.exchange:
                torch.distributed.all_reduce(deg_u, group=self.group)
                torch.distributed.all_reduce(deg_i, group=self.group)
            self._balance(deg_u, deg_i)
        mine = {}
        for side, (r_, c_) in (("users", (rows, cols)), ("items", (cols, rows))):
            dest = self.shard[side].owner_of(r_)
            mine[side] = self._exchange(dest, (r_, c_, vals)) if self.world > 1 else (r_, c_, vals)
        self._build_shards(mine)

    def _balance(self, deg_users, deg_items):
        shard = {}
        for side, deg in (("users", deg_users), ("items", deg_items)):
            owner, local, counts = balanced_assignment(row_cost(deg, self.f), self.world)
            shard[side] = Sharding(self.n[side], self.world, self.rank, owner.to(self.device), local.to(self.device), counts)
        self._configure(shard)

    def _exchange(self, dest, tensors):
        """Send every entry to rank dest[entry]; returns what this rank receives (same tensor layout).  RCCL: one
        all_to_all_single per tensor with the counts exchanged first; gloo (CPU rehearsal) has no all_to_all: every rank
        gathers the padded buffers and keeps its part."""
        W = self.world
        order = torch.argsort(dest, stable=True)
        send_counts = torch.bincount(dest, minlength=W)
        backend = torch.distributed.get_backend(self.group)
        if backend == "nccl":
            recv_counts = torch.empty_like(send_counts)
            torch.distributed.all_to_all_single(recv_counts, send_counts, group=self.group)
            sc, rc = send_counts.tolist(), recv_counts.tolist()
            out = []
            for t in tensors:
                buf = torch.empty(sum(rc), dtype=t.dtype, device=t.device)
                torch.distributed.all_to_all_single(buf, t[order].contiguous(), rc, sc, group=self.group)
                out.append(buf)
            return tuple(out)
        counts_all = [torch.empty_like(send_counts) for _ in range(W)]
        torch.distributed.all_gather(counts_all, send_counts, group=self.group)
        cap = int(torch.stack(counts_all).sum(1).max())
        out = []
        for t in tensors:
            pad = torch.zeros(cap, dtype=t.dtype, device=t.device)
            pad[: t.numel()] = t[order]
            gathered = [torch.empty_like(pad) for _ in range(W)]
            torch.distributed.all_gather(gathered, pad, group=self.group)
            parts = []
            for src in range(W):
                c = counts_all[src]
                lo = int(c[: self.rank].sum())
                parts.append(gathered[src][lo: lo + int(c[self.rank])])
            out.append(torch.cat(parts))
        return tuple(out)

    def _build_shards(self, mine):
        """mine[side] = (row ids, column ids, values) of the entries whose ``side`` row this rank owns."""
        for side in ("users", "items"):
            r_, c_, v_ = mine[side]
            self.csr[side] = self._shard(side, r_, c_, v_)
        for side in ("users", "items"):
            if self.reduce[side]:
                self.csr_red[side] = self._shard_reduce(side)          # (built on the dense positions of ``side``)
        self._setup_sparse()                                           # may re-index csr["items"] into a compact gather of the users
        for side in ("users", "items"):
            full = self.csr[side]
            bounds = self.chunk_bounds[side]
            if len(bounds) == 1:
                self.csr_chunks[side] = [full]
            else:
                edges = full.indptr[[lo for lo, _ in bounds] + [self.rpr[side]]].tolist()
                self.csr_chunks[side] = [
                    Csr(self.K, full.indptr[lo: lo + ln + 1] - edges[c], full.indices[edges[c]: edges[c + 1]],
                        full.values[edges[c]: edges[c + 1]], full.n_cols, self.f, self.bias)
                    for c, (lo, ln) in enumerate(bounds)]
        for side in ("users", "items"):
            if not self.reduce[side]:
                self._setup_pipe(side)

    def _setup_sparse(self):
        """Need-list gather of the users (module docstring), decided jointly: every rank must take the same path."""
        S, consumer = "users", "items"
        forced = os.environ.get("WMF_SPARSE") if self._sparse_arg is None else ("1" if self._sparse_arg else "0")
        if not self.exchange or self.reduce[consumer] or forced == "0":
            return
        W, dev = self.world, self.device
        shard = self.csr[consumer]
        idx = shard.indices.to(torch.int64)
        P = torch.unique(idx)                                          # needed dense positions, ascending = the compact order
        total = W * self.rpr[S]
        frac = torch.tensor([P.numel() / max(1, total)], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(frac, op=torch.distributed.ReduceOp.MAX, group=self.group)
        if forced != "1" and float(frac.item()) >= SPARSE_MAX_FRACTION:
            return
        bounds = self.chunk_bounds[S]
        starts = torch.tensor([W * lo for lo, _ in bounds], dtype=torch.int64, device=dev)
        los = torch.tensor([lo for lo, _ in bounds], dtype=torch.int64, device=dev)
        lns = torch.tensor([ln for _, ln in bounds], dtype=torch.int64, device=dev)
        c = torch.bucketize(P, starts, right=True) - 1                 # chunk of every needed position
        rel = P - starts[c]
        src = rel // lns[c]                                            # the rank that owns the row ...
        row = los[c] + rel % lns[c]                                    # ... and its local row there
        C = len(bounds)
        recv_counts = torch.bincount(c * W + src, minlength=C * W).reshape(C, W)
        coff = torch.zeros(C + 1, dtype=torch.int64)
        coff[1:] = torch.cumsum(recv_counts.sum(1).cpu(), 0)
        # tell every owner which of its rows this rank needs (one exchange at set-up): (requester, chunk, local row)
        me = torch.full_like(row, self.rank)
        if W > 1:
            req, cc, rr = self._exchange(src, (me, c, row))
        else:
            req, cc, rr = me, c, row
        order = torch.argsort(cc * W + req, stable=True)               # chunk-major, requester-major; rows stay ascending
        req, cc, rr = req[order], cc[order], rr[order]
        send_counts = torch.bincount(cc * W + req, minlength=C * W).reshape(C, W)
        soff = torch.zeros(C + 1, dtype=torch.int64)
        soff[1:] = torch.cumsum(send_counts.sum(1).cpu(), 0)
        self.need[S] = {
            "send_idx": [rr[int(soff[k]): int(soff[k + 1])].contiguous() for k in range(C)],
            "send_counts": [[int(x) for x in send_counts[k].tolist()] for k in range(C)],
            "recv_counts": [[int(x) for x in recv_counts[k].tolist()] for k in range(C)],
            "needed_fraction": P.numel() / max(1, total),
        }
        shard.indices.copy_(torch.searchsorted(P, idx).to(torch.int32))        # the item-major CSR now points into the compact matrix
        n_compact = max(1, int(P.numel()))
        shard.n_cols = n_compact
        self.sparse[S] = True
        self.chunk_range[S] = [(int(coff[k]), int(coff[k + 1])) for k in range(C)]
        f32 = torch.float32
        self._wait(S)                           # gathers of factors loaded before the interactions still write the old matrix
        self.X[S] = torch.zeros(n_compact, self.ld, dtype=f32, device=dev)
        self.V[S] = torch.zeros(n_compact, self.ldv, dtype=f32, device=dev)
        self.bias_vec[S] = torch.zeros(n_compact, 2, dtype=f32, device=dev) if self.split else torch.zeros(n_compact, dtype=f32, device=dev)
        if self.has_factors[S]:                                        # factors loaded before the interactions: publish them again
            for k in range(C):
                self._publish(S, k)

    def exchange_bytes_sent(self, side):
        """Bytes this rank SENDS to other ranks when its freshly solved block of ``side`` is exchanged once (one half step):
        the dense all-gather sends the whole block to each of the W - 1 peers, the need-list gather only the rows each peer
        asked for; 0 when the other side runs in reduce mode (the block then stays where it is)."""
        if not self.exchange or self.reduce[self._other(side)]:
            return 0
        row_bytes = self.ld * 4
        if self.sparse[side]:
            sc = self.need[side]["send_counts"]
            return sum(n for per_chunk in sc for d, n in enumerate(per_chunk) if d != self.rank) * row_bytes
        return (self.world - 1) * self.rpr[side] * row_bytes

    def _setup_pipe(self, side):
        """Pipelined gather mode for ``side`` (module docstring) if its rows are heavy enough per arriving chunk."""
        fixed = self._other(side)
        bounds = self.chunk_bounds[fixed]
        full = self.csr[side]
        forced = os.environ.get("WMF_PIPE") if self.pipe_mode is None else ("1" if self.pipe_mode else "0")
        heavy = full.nnz >= PIPE_MIN_ENTRIES * max(1, self.n_local[side]) * len(bounds)
        self.pipe[side] = (self.exchange and len(bounds) > 1 and self.pr > 0 and not self.reduce[fixed]
                           and (forced == "1" or (forced is None and heavy)))
        if not self.pipe[side]:
            return
        W, n = self.world, self.rpr[side]
        idx = full.indices.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(n, device=self.device), full.indptr[1:] - full.indptr[:-1])
        subs = []
        for start, stop in self.chunk_range[fixed]:
            m = (idx >= start) & (idx < stop)
            cnt = torch.bincount(rows[m], minlength=n)
            ptr = torch.zeros(n + 1, dtype=torch.int64, device=self.device)
            torch.cumsum(cnt, 0, out=ptr[1:])
            v = full.values[m].contiguous()
            subs.append((ptr, cnt.to(torch.int32).contiguous(), full.indices[m].contiguous(), v,
                         torch.empty_like(v) if self.bias else None))
        self.csr_pipe[side] = subs
        self.partial_pipe[side] = torch.empty(n * len(bounds), self.pr, dtype=torch.float32, device=self.device)

    def _shard_reduce(self, side):
        """Reduce mode: every row of ``side`` (at its gathered position) x this rank's rows of the other side (local
        rows) -- the transpose of this rank's own shard of the other side, whose column indices ARE the gathered positions
        of ``side``.  Returns (indptr, degrees int32, indices int32, values, w_eff workspace)."""
        o = self.csr[self._other(side)]
        local_rows = torch.repeat_interleave(torch.arange(o.n_rows, device=self.device), o.indptr[1:] - o.indptr[:-1])
        indptr, idx, v = coo_to_csr(o.indices.to(torch.int64), local_rows, o.values, self.world * self.rpr[side], self.K, o.n_rows)
        deg = (indptr[1:] - indptr[:-1]).to(torch.int32).contiguous()
        w_eff = torch.empty_like(v) if self.bias and not self.split else None    # (split layout: the kernels take the bias with the row)
        return indptr.contiguous(), deg, idx.contiguous(), v.contiguous(), w_eff

    def _shard(self, side, rows, cols, vals, plan=True):
        """CSR of this rank's rows of ``side`` (all of them its own already): local row x gathered position of the column."""
        other = self._other(side)
        local = self.shard[side].local_of(rows)
        indptr, idx, v = coo_to_csr(local, self.positions(other, cols), vals, self.rpr[side], self.K, self.world * self.rpr[other])
        return Csr(self.K, indptr, idx, v, self.world * self.rpr[other], self.f, self.bias, plan=plan)

    def set_factors(self, side, full):
        """Load a full host/device [n, f] factor matrix (reference layout) into this rank's block."""
        full = torch.as_tensor(full, dtype=torch.float32)
        sh = self.shard[side]
        mine = (full[self.rank::self.world] if sh.owner is None else full[sh.my_ids(full.device)]).to(self.device)
        blk = self.factors[side]
        blk.zero_()
        blk[: mine.shape[0], : self.f] = mine
        self.has_factors[side] = True
        for c in range(len(self.chunk_bounds[side])):
            self._publish(side, c)

    def get_factors(self, side):
        """Full [n, f] factor matrix on the host, in id order."""
        ids = torch.arange(self.n[side], device=self.device)
        if self.sparse[side]:                        # the compact matrix holds only what this rank's items need: gather densely, once
            W = self.world
            self._wait(side)
            dense = torch.zeros(W * self.rpr[side], self.ld, dtype=torch.float32, device=self.device)
            for lo, ln in self.chunk_bounds[side]:
                torch.distributed.all_gather_into_tensor(dense[W * lo: W * (lo + ln)], self.factors[side][lo: lo + ln].contiguous(),
                                                         group=self.group)
            return dense[self.positions(side, ids)][:, : self.f].cpu().numpy()
        self._ensure_gathered(side)
        return self.X[side][self.positions(side, ids)][:, : self.f].cpu().numpy()

    # ---------------------------------------------------------------- exchange
    def _publish(self, side, c, force=False):
        """Start the all-gather of chunk ``c`` of this rank's freshly written block.  It is ordered after the
        kernels already enqueued on the current stream and runs beside whatever is enqueued next."""
        if not self.exchange:
            return
        if self.reduce[self._other(side)] and not force:
            self._stale[side] = True                 # the other side reads only this rank's block: gather on demand
            return
        lo, ln = self.chunk_bounds[side][c]
        W = self.world
        if self.sparse[side]:
            need = self.need[side]
            start, stop = self.chunk_range[side][c]
            # packed per destination, rows ascending (the library's copy kernel; the CPU stand-in of the tests indexes)
            rows = need["send_idx"][c]
            send = self.K.gather_rows(self.factors[side], rows) if hasattr(self.K, "gather_rows") else self.factors[side].index_select(0, rows)
            self._pending[side].append(self._all_to_all_rows(self.X[side][start:stop], send, need["recv_counts"][c],
                                                             need["send_counts"][c]))
            return
        self._pending[side].append(torch.distributed.all_gather_into_tensor(
            self.X[side][W * lo: W * (lo + ln)], self.factors[side][lo: lo + ln], group=self.group, async_op=True))

    def _all_to_all_rows(self, out, send, recv_counts, send_counts):
        """out = the rows every rank packed for this one (source-major), send = this rank's rows packed per destination:
        ONE asynchronous all_to_all_single with split sizes, whatever the backend -- RCCL on device tensors; gloo (the CPU
        rehearsal of the tests, and ranks sharing a GPU) on host tensors, device tensors staged through the host.  The same
        split-size bookkeeping therefore runs under every test that 8 GPUs will run.  Returns something with .wait()."""
        if torch.distributed.get_backend(self.group) == "nccl" or not out.is_cuda:
            work = torch.distributed.all_to_all_single(out, send, list(recv_counts), list(send_counts), group=self.group, async_op=True)
            return _HeldWork(work, send)             # the send buffer lives exactly as long as its exchange is in flight
        torch.cuda.current_stream().synchronize()    # (gloo reads the host copy: the rows must have been written)
        send_h, out_h = send.cpu(), torch.empty(out.shape, dtype=out.dtype)
        work = torch.distributed.all_to_all_single(out_h, send_h, list(recv_counts), list(send_counts), group=self.group, async_op=True)
        return _HeldWork(work, send_h, landing=(out, out_h))

    def _ensure_gathered(self, side):
        """X[side] complete on this rank (reduce mode leaves it un-gathered until somebody needs it)."""
        if self._stale[side]:
            self._stale[side] = False
            for c in range(len(self.chunk_bounds[side])):
                self._publish(side, c, force=True)
        self._wait(side)

    def _wait(self, side):
        """Make the current stream wait for every all-gather of ``side`` still in flight."""
        for work in self._pending[side]:
            if work is not None:
                work.wait()
        self._pending[side] = []

    # ---------------------------------------------------------------- one half step
    def prepare(self, fixed):
        """Gramian of the fixed side, its Cholesky factor, and the gathered whitened factors.
        wmf_model.py:215 / :328-332."""
        K = self.K
        blk = self.factors[fixed]
        K.gram(blk, self.n_local[fixed], self.f, self.ld, self.bias, self.G, self.ws)
        if self.exchange:
            torch.distributed.all_reduce(self.G, group=self.group)
        K.factorize(self.G, self.f, self.ld, self.gamma, self.W_white, self.W_unwhite, self.info, self.ws)
        V, bvec = self.V[fixed], self.bias_vec[fixed]
        if not self.exchange:
            K.row_transform(self.X[fixed], self.n_local[fixed], self.f, self.ld, self.W_white, self.white_mode, V,
                            bvec if self.bias else None)
            return
        # whiten chunk by chunk, each as soon as ITS all-gather has landed: the passes over the early chunks run while
        # the later chunks are still on the wire (in-flight gathers complete in issue order)
        if self._stale[fixed]:
            self._ensure_gathered(fixed)                             # it was last updated while nobody needed all of it
        pending, self._pending[fixed] = self._pending[fixed], []
        done = len(self.chunk_bounds[fixed]) - len(pending)          # chunks whose gather was waited for earlier
        for c, (start, stop) in enumerate(self.chunk_range[fixed]):
            if c >= done and pending[c - done] is not None:
                pending[c - done].wait()
            rows = slice(start, stop)
            if stop > start:
                K.row_transform(self.X[fixed][rows], stop - start, self.f, self.ld, self.W_white, self.white_mode, V[rows],
                                bvec[rows] if self.bias else None)

    def update(self, side):
        """Solve every local row of ``side`` against the prepared fixed side.  wmf_model.py:220-239."""
        K = self.K
        fixed = self._other(side)
        bvec = self.bias_vec[fixed] if self.bias else None
        self._wait(side)                             # nobody may still be reading the block that is about to be rewritten
        for c, ((lo, ln), csr) in enumerate(zip(self.chunk_bounds[side], self.csr_chunks[side])):
            g = self.g[side][lo: lo + ln]
            K.solve_rows(csr._plan, self.V[fixed], bvec, csr.indptr, csr.indices, csr.values, ln, self.f, self.ld, g, self.fail,
                         self.solve_flags)
            real = max(0, min(ln, self.n_local[side] - lo))
            K.row_transform(g, real, self.f, self.ld, self.W_unwhite, self.unwhite_mode, self.factors[side][lo: lo + ln], None)
            self._publish(side, c)
        self.has_factors[side] = True

    def half_step(self, side):
        """``side`` <- recompute_factors(other side, C or C^T, gamma)."""
        if self.reduce[side]:
            return self._half_step_reduce(side)
        if self.pipe[side]:
            return self._half_step_pipelined(side)
        self.prepare(self._other(side))
        self.update(side)

    def _half_step_pipelined(self, side):
        """Pipelined gather mode (module docstring): accumulate over each chunk of the fixed side as it lands."""
        K, W = self.K, self.world
        fixed = self._other(side)
        K.gram(self.factors[fixed], self.n_local[fixed], self.f, self.ld, self.bias, self.G, self.ws)
        torch.distributed.all_reduce(self.G, group=self.group)
        K.factorize(self.G, self.f, self.ld, self.gamma, self.W_white, self.W_unwhite, self.info, self.ws)
        if self._stale[fixed]:
            self._ensure_gathered(fixed)
        pending, self._pending[fixed] = self._pending[fixed], []
        bounds = self.chunk_bounds[fixed]
        done = len(bounds) - len(pending)
        V, bvec = self.V[fixed], self.bias_vec[fixed]
        self._wait(side)
        n, C = self.rpr[side], len(bounds)
        for c, ((start, stop), (ptr, deg, idx, val, w_eff)) in enumerate(zip(self.chunk_range[fixed], self.csr_pipe[side])):
            if c >= done and pending[c - done] is not None:
                pending[c - done].wait()
            rows = slice(start, stop)
            if stop > start:
                K.row_transform(self.X[fixed][rows], stop - start, self.f, self.ld, self.W_white, self.white_mode, V[rows],
                                bvec[rows] if self.bias else None)
            K.accumulate_rows(V, bvec if self.bias else None, ptr, deg, idx, val, n, idx.numel(), self.f, self.ld,
                              self.partial_pipe[side], w_eff, slot_stride=C, slot_offset=c)
        before = self._fail_snapshot()
        K.eliminate_rows(self.partial_pipe[side], n, self.f, self.ld, self.g[side], self.fail, self.scratch_rows, slots_per_row=C)
        if self._summed_system_failed(before):
            return self._redo_through_gather(side)
        K.row_transform(self.g[side], self.n_local[side], self.f, self.ld, self.W_unwhite, self.unwhite_mode, self.factors[side], None)
        self.has_factors[side] = True
        for c in range(len(self.chunk_bounds[side])):
            self._publish(side, c)

    def _half_step_reduce(self, side):
        """Reduce mode (module docstring): the fixed side stays where it is; the rows' partial systems travel."""
        K, W = self.K, self.world
        fixed = self._other(side)
        blk = self.factors[fixed]
        K.gram(blk, self.n_local[fixed], self.f, self.ld, self.bias, self.G, self.ws)
        torch.distributed.all_reduce(self.G, group=self.group)
        K.factorize(self.G, self.f, self.ld, self.gamma, self.W_white, self.W_unwhite, self.info, self.ws)
        v_loc, b_loc = self.V[fixed][: self.rpr[fixed]], self.bias_vec[fixed][: self.rpr[fixed]]     # this rank's block only
        K.row_transform(blk, self.n_local[fixed], self.f, self.ld, self.W_white, self.white_mode, v_loc, b_loc if self.bias else None)
        indptr, deg, idx, vals, w_eff = self.csr_red[side]
        self._wait(side)
        works = []
        for c, (lo, ln) in enumerate(self.chunk_bounds[side]):
            r0, r1 = W * lo, W * (lo + ln)
            part = self.partial_all[side][r0:r1]
            if self.bias and (c == 0 or self.split):  # values - bias[indices] for ALL entries, once per half step (split layout:
                K.accumulate_rows(v_loc, b_loc, indptr[r0: r1 + 1], deg[r0:r1], idx, vals, r1 - r0, idx.numel(), self.f, self.ld,
                                  part, w_eff)         # no such pass, the kernels take the bias with the gathered row)
            else:
                K.accumulate_rows(v_loc, None, indptr[r0: r1 + 1], deg[r0:r1], idx, w_eff if self.bias else vals, r1 - r0,
                                  idx.numel(), self.f, self.ld, part, None)
            works.append(self._reduce_scatter(self.partial_mine[side][lo: lo + ln], part))
        for w in works:
            if w is not None:
                w.wait()
        before = self._fail_snapshot()
        K.eliminate_rows(self.partial_mine[side], self.rpr[side], self.f, self.ld, self.g[side], self.fail, self.scratch_rows)
        if self._summed_system_failed(before):
            return self._redo_through_gather(side)
        K.row_transform(self.g[side], self.n_local[side], self.f, self.ld, self.W_unwhite, self.unwhite_mode, self.factors[side], None)
        self.has_factors[side] = True
        for c in range(len(self.chunk_bounds[side])):
            self._publish(side, c)

    def _fail_snapshot(self):
        """The sticky failure counter as it stands before an eliminate pass (bias models only; None otherwise)."""
        return self.fail[:1].clone() if self.bias else None

    def _summed_system_failed(self, before):
        """Bias models in reduce / pipelined mode: did any rank meet a row whose summed system is not positive definite
        (negative bias-adjusted weights) IN THE ELIMINATE PASS JUST ENQUEUED?  ``before`` is the counter's value in front of
        that pass: the counter is sticky and also holds what the previous half step left (a singular row of the gather path's
        pivoted kernel is only looked at once per iteration, in check_numerics) -- such a count is neither taken for a failure
        of this pass nor lost: only the delta is cleared.  One small all-reduce and one host sync per half step -- the price of
        letting bias models use the two modes; models without biases never ask (their systems are positive definite by
        construction)."""
        if not self.bias:
            return False
        flag = self.fail[:1] - before
        if self.exchange:
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX, group=self.group)
        if int(flag.item()) == 0:
            return False
        self.fail[:1].copy_(before)                  # those rows are solved again below, by kernels that can pivot
        return True

    def _redo_through_gather(self, side):
        """The half step again through the all-gather path: its row kernels see the rows' own entries and hand a row with
        negative weights to the pivoted kernel (the reference's np.linalg.solve, wmf_model.py:350).  The Gramian and its
        factor are still those of this half step; the fixed side is gathered (it may not have been: reduce mode) and
        whitened as a whole."""
        K, W = self.K, self.world
        fixed = self._other(side)
        self._ensure_gathered(fixed)
        K.row_transform(self.X[fixed], self.X[fixed].shape[0], self.f, self.ld, self.W_white, self.white_mode, self.V[fixed],
                        self.bias_vec[fixed] if self.bias else None)
        self.update(side)

    def _reduce_scatter(self, out, inp):
        """out = this rank's slice of the sum over ranks of inp (rank-major blocks).  Backends without a
        reduce-scatter (gloo) get an all-reduce and a copy."""
        if torch.distributed.get_backend(self.group) == "nccl":
            return torch.distributed.reduce_scatter_tensor(out, inp, group=self.group, async_op=True)
        torch.distributed.all_reduce(inp, group=self.group)
        n = out.shape[0]
        out.copy_(inp[self.rank * n: (self.rank + 1) * n])
        return None

    def half_step_unweighted(self, side):
        """Closed form of the un-weighted branch, wmf_model.py:85,88:
        side = R . Y . (Y^T Y + gamma I)^-1  =  (R . V) . L^-1  with V = Y L^-T."""
        K = self.K
        fixed = self._other(side)
        self.prepare(fixed)
        c = self.csr[side]
        K.spmm_rows(self.V[fixed], c.indptr, c.indices, c.values, c.n_rows, self.f, self.ld, self.g[side])
        K.row_transform(self.g[side], self.n_local[side], self.f, self.ld, self.W_unwhite, self.unwhite_mode, self.factors[side], None)
        self.has_factors[side] = True
        for c in range(len(self.chunk_bounds[side])):
            self._publish(side, c)

    iter_on = True      # the matrix-free iteration kernel (csrc/wmf_iter.hip) is part of the solve; tools that switch it off with
                        # wmf_debug_set_flags(268435456) set this to False for bench.py's byte model

    def iter_stats(self, side):
        """(rows solved by the matrix-free iteration, rows it handed back to the elimination kernels, applications of the row
        operator, rows on the Chebyshev recurrence) over the half steps of ``side`` since the last call, summed over the
        chunks; synchronises the device."""
        out = np.zeros(4, dtype=np.int64)
        for c in self.csr_chunks.get(side, []):
            out += c.iter_stats()
        return out

    def check_numerics(self):
        """Host sync: raise if a Gramian was not positive definite or a row system was singular (the reference's
        np.linalg.solve raises LinAlgError there).  The two flags are sticky on the device, so one check per iteration
        sees a failure of either half step; with several ranks they are max-reduced first, so that every rank raises
        together instead of one leaving the others in the next collective."""
        flags = torch.stack((self.info[0], self.fail[0])).to(torch.int32)
        if self.exchange:
            torch.distributed.all_reduce(flags, op=torch.distributed.ReduceOp.MAX, group=self.group)
        info, fail = (int(x) for x in flags.cpu())
        if info or fail:
            self.info.zero_()
            self.fail.zero_()
        if info:
            raise _lib.WmfNumericError(f"Gramian + gamma*I is not positive definite (leading minor {info})")
        if fail:
            raise _lib.WmfNumericError(f"{fail} row systems were singular (largest count over the ranks)")

    # ---------------------------------------------------------------- evaluation
    def make_eval_shard(self, indptr, indices, values):
        """User-row shard of a utility matrix for eval_sums (columns are item positions)."""
        dev = self.device
        indptr = indptr.to(dev, torch.int64)
        cols = indices.to(dev, torch.int64)
        self._check_csr(indptr, cols, "utility_mat / eval_mat")
        counts = indptr[1:] - indptr[:-1]
        rows = torch.repeat_interleave(torch.arange(self.n["users"], device=dev), counts)
        vals = values.to(dev, torch.float32)
        if self.world > 1:
            m = self.shard["users"].owner_of(rows) == self.rank
            rows, cols, vals = rows[m], cols[m], vals[m]
        return self._shard("users", rows, cols, vals, plan=False)

    def make_eval_shard_distributed(self, first_user, indptr, indices, values):
        """make_eval_shard from this rank's block of user rows (set_interactions_distributed): every entry travels to the
        owner of its user."""
        dev = self.device
        indptr = indptr.to(dev, torch.int64)
        cols = indices.to(dev, torch.int64)
        vals = values.to(dev, torch.float32)
        counts = indptr[1:] - indptr[:-1]
        rows = first_user + torch.repeat_interleave(torch.arange(indptr.numel() - 1, device=dev), counts)
        if self.world > 1:
            rows, cols, vals = self._exchange(self.shard["users"].owner_of(rows), (rows, cols, vals))
        return self._shard("users", rows, cols, vals, plan=False)

    def eval_sums(self, shard):
        """(sum of squared errors, sum of absolute errors, count) over the stored non-zero entries
        of ``shard``; all-reduced over ranks.  base_model.py:163-176."""
        self._ensure_gathered("items")
        self.K.eval_sqerr(self.factors["users"], self.X["items"], self.f, self.ld, self.bias, shard.indptr, shard.indices,
                          shard.values, shard.n_rows, self.eval_out, self.eval_ws)
        if self.exchange:
            torch.distributed.all_reduce(self.eval_out, group=self.group)
        return tuple(float(x) for x in self.eval_out.cpu())

    # ---------------------------------------------------------------- roofline bookkeeping
    def algorithmic_bytes_half(self, side):
        """z*(4f+8) + n*(4f+4) + 4f^2 for this rank's rows, plus 4*m*f for
        the Gramian of the fixed side (this rank's share of it)."""
        c = self.csr[side]
        f = self.f
        return c.nnz * (4 * f + 8) + self.n_local[side] * (4 * f + 4) + 4 * f * f + 4 * self.n_local[self._other(side)] * f
