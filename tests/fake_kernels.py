"""CPU stand-in for recmodel_amd.engine.HipKernels -- TEST INFRASTRUCTURE ONLY.

Same method surface, NumPy arithmetic on CPU torch tensors, so that the host logic of AlsEngine
(round-robin sharding, position maps, the Gramian all-reduce and the all-gather of the whitened
block, padding rows, eval reduction) can run on gloo ranks without a GPU.  It is never imported by
the package; the HIP kernels themselves are tested on the GPU (tests/test_gpu_parity.py)."""
import numpy as np


class NumpyKernels:
    def ld_for(self, f):
        return (f + 3) & ~3

    def whitened_row_floats(self, f, ld, bias):
        """The device library's split layout (include/wmf_hip.h, wmf_row_transform): packed body + {last feature, bias}
        pairs for bias models at f = 16 m + 1 <= 144 (m + 1 not a multiple of 4), ld = f + 3."""
        split = bias and 16 < f <= 144 and f % 16 == 1 and (f // 16) % 4 != 3 and ld == f + 3
        return f - 1 if split else ld

    def _whitened(self, V, bias_vec, f, ld):
        """(rows x f whitened factors, bias or None) from either layout."""
        if bias_vec is not None and self.whitened_row_floats(f, ld, True) == f - 1:
            side = bias_vec.numpy().reshape(-1, 2).astype(np.float64)
            body = V.numpy().reshape(-1, f - 1).astype(np.float64)
            return np.concatenate([body, side[:, :1]], axis=1), side[:, 1]
        v = V.numpy().astype(np.float64)[:, :f]
        return v, (bias_vec.numpy().astype(np.float64) if bias_vec is not None else None)

    def gram_workspace_bytes(self, f):
        return 16

    def eval_workspace_bytes(self):
        return 16

    def gram(self, Y, m, f, ld, bias, G, ws):
        y = Y.numpy()[:m, :f].astype(np.float64).copy()
        if bias:
            y[:, 0] = 1.0
        G.numpy()[:] = (y.T @ y).reshape(-1)

    def factorize(self, G, f, ld, lam, W_white, W_unwhite, info, ws):
        A = G.numpy().reshape(f, f) + lam * np.eye(f)
        W_white.numpy()[:] = 0
        W_unwhite.numpy()[:] = 0
        try:
            Linv = np.linalg.inv(np.linalg.cholesky(A))
        except np.linalg.LinAlgError:
            info.numpy()[0] = 1             # sticky, like the device kernel: only the caller resets it
            return
        W_white.numpy()[:, :f] = Linv.T
        W_unwhite.numpy()[:, :f] = Linv

    def rolled_layout_supported(self, f, ld):
        """wmf_rolled_layout_supported: bias models with a 128-float body."""
        return f == 129 and self.whitened_row_floats(f, ld, True) == f - 1

    def row_transform(self, inp, m, f, ld, W, set_col0_one, out, col0_out):
        x = inp.numpy()[:m, :f].astype(np.float64).copy()
        mode = int(set_col0_one)
        if mode == 4:                                   # the input is in rolled coordinates: position c holds feature c + 1 (mod f)
            x = np.roll(x, 1, axis=1)
            set_col0_one = False
        split = set_col0_one and self.whitened_row_floats(f, ld, True) == f - 1
        if set_col0_one:
            if split:
                col0_out.numpy().reshape(-1, 2)[:m, 1] = inp.numpy()[:m, 0]
            elif col0_out is not None:
                col0_out.numpy()[:m] = inp.numpy()[:m, 0]
            x[:, 0] = 1.0
        y = x @ W.numpy()[:, :f].astype(np.float64)
        if mode == 3:                                   # written in rolled coordinates (the stand-in leaves the mantissa bits alone:
            y = np.roll(y, -1, axis=1)                  # its solve_rows reads the bias from the pairs)
        if split:
            out.numpy().reshape(-1, f - 1)[:m] = y[:, : f - 1]
            col0_out.numpy().reshape(-1, 2)[:m, 0] = y[:, f - 1]
            return
        o = out.numpy()
        o[:m] = 0
        o[:m, :f] = y

    def plan_create(self, indptr_host, n, f, bias=False):
        deg = np.diff(indptr_host)
        stats = np.zeros(14, dtype=np.int64)
        stats[0], stats[4] = n, deg.sum()
        return ("plan", n, f), stats

    def plan_destroy(self, handle):
        pass

    def solve_rows(self, plan, V, bias_vec, indptr, indices, values, n, f, ld, g, fail, flags=0):
        v, b = self._whitened(V, bias_vec, f, ld)
        ip, ix, w = indptr.numpy(), indices.numpy(), values.numpy().astype(np.float64)
        out = g.numpy()
        out[:n] = 0
        for u in range(n):
            lo, hi = ip[u], ip[u + 1]
            if hi == lo:
                continue
            idx = ix[lo:hi]
            wu = w[lo:hi] - (b[idx] if b is not None else 0.0)
            vu = v[idx, :f]
            out[u, :f] = np.linalg.solve(np.eye(f) + vu.T @ (vu * wu[:, None]), (wu + 1.0) @ vu)

    # ---- reduce mode: per-row partial systems [f*f (A) | f (y)] in float32, summed over ranks by the engine
    def partial_row_floats(self, f):
        return f * f + f

    def accumulate_rows(self, V, bias_vec, indptr, degrees, indices, values, n, nnz, f, ld, partial, w_eff, slot_stride=1,
                        slot_offset=0):
        ip = indptr.numpy()
        idx = indices.numpy()
        w = values.numpy().astype(np.float64)
        Vn, b = self._whitened(V, bias_vec, f, ld)
        if b is not None:
            w = w - b[idx]
            if w_eff is not None:
                w_eff.numpy()[:] = w
        out = partial.numpy().reshape(-1, f * f + f)[slot_offset::slot_stride]
        assert (np.diff(ip) == degrees.numpy()).all()
        for i in range(n):
            cols = idx[ip[i]: ip[i + 1]]
            U, wi = Vn[cols], w[ip[i]: ip[i + 1]]
            out[i, : f * f] = (U.T @ (U * wi[:, None])).reshape(-1)
            out[i, f * f:] = (wi + 1.0) @ U

    def eliminate_rows(self, partial, n, f, ld, g, fail, scratch, slots_per_row=1):
        P = partial.numpy().astype(np.float64).reshape(-1, slots_per_row, f * f + f)[:n].sum(axis=1)
        out = g.numpy()
        out[:n] = 0
        for i in range(n):
            A = np.eye(f) + P[i, : f * f].reshape(f, f)
            try:                                       # like the device kernel: no pivoting here, a system that is not
                np.linalg.cholesky(A)                  # positive definite is counted and left for the caller
            except np.linalg.LinAlgError:
                fail.numpy()[0] += 1
                continue
            out[i, :f] = np.linalg.solve(A, P[i, f * f:])

    def spmm_rows(self, V, indptr, indices, values, n, f, ld, g):
        v = V.numpy().astype(np.float64)
        ip, ix, w = indptr.numpy(), indices.numpy(), values.numpy().astype(np.float64)
        out = g.numpy()
        out[:n] = 0
        for u in range(n):
            lo, hi = ip[u], ip[u + 1]
            out[u, :f] = w[lo:hi] @ v[ix[lo:hi], :f]

    def eval_sqerr(self, users, items, f, ld, bias, indptr, indices, values, n, out3, ws):
        x, y = users.numpy().astype(np.float64), items.numpy().astype(np.float64)
        ip, ix, val = indptr.numpy(), indices.numpy(), values.numpy().astype(np.float64)
        rows = np.repeat(np.arange(n), np.diff(ip))
        keep = val != 0
        rows, cols, val = rows[keep], ix[keep], val[keep]
        if bias:
            pred = (x[rows, 1:f] * y[cols, 1:f]).sum(1) + x[rows, 0] + y[cols, 0]
        else:
            pred = (x[rows, :f] * y[cols, :f]).sum(1)
        e = val - pred
        out3.numpy()[:] = [np.sum(e * e), np.sum(np.abs(e)), float(len(e))]

    def confidence_transform(self, values, alpha, beta, mode):
        v = values.numpy()
        v[:] = alpha * np.log(1 + beta * v) if mode == 0 else alpha * v
