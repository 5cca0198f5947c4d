"""wmf_coo_to_csr (include/wmf_hip.h): the device transpose / re-sorting of stored entries, against SciPy and against the
stable (row, column) order the reference's ``count_mat.T.tocsr()`` (RecModel/wmf_model.py:128) produces."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

pytestmark = pytest.mark.gpu


def _kernels():
    from recmodel_amd.engine import HipKernels
    return HipKernels()


@pytest.mark.parametrize("n_rows,n_cols,nnz", [(1, 1, 1), (7, 5, 40), (1000, 37, 20000), (300000, 100000, 3000000)])
def test_coo_to_csr_is_the_stable_row_column_sort(n_rows, n_cols, nnz):
    rng = np.random.default_rng(n_rows + nnz)
    rows = rng.integers(0, max(1, n_rows - 2), nnz)            # the last rows stay empty
    cols = rng.integers(0, n_cols, nnz)
    if nnz > 10:
        rows[5], cols[5] = rows[3], cols[3]                    # a duplicate entry: both survive, in stored order
    vals = rng.standard_normal(nnz).astype(np.float32)
    K = _kernels()
    indptr, indices, values = K.coo_to_csr(torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda(),
                                           torch.from_numpy(vals).cuda(), n_rows, n_cols)
    order = np.lexsort((np.arange(nnz), cols, rows))           # stable in (row, col)
    np.testing.assert_array_equal(indices.cpu().numpy(), cols[order].astype(np.int32))
    np.testing.assert_array_equal(values.cpu().numpy(), vals[order])
    want_ptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n_rows))])
    np.testing.assert_array_equal(indptr.cpu().numpy(), want_ptr)
    assert indices.dtype == torch.int32 and indptr.dtype == torch.int64


def test_transpose_matches_scipy():
    """(rows, cols) swapped = the transpose: what AlsEngine.set_interactions does for the item side."""
    from recmodel_amd import synth
    indptr, indices, counts = synth.make_counts(5000, 700, 12, seed=3)
    C = synth.to_scipy(indptr, indices, counts, (5000, 700))
    CT = C.T.tocsr()
    CT.sort_indices()
    rows = torch.repeat_interleave(torch.arange(5000), indptr[1:] - indptr[:-1])
    K = _kernels()
    ptr, idx, val = K.coo_to_csr(indices.to(torch.int64).cuda(), rows.cuda(), counts.to(torch.float32).cuda(), 700, 5000)
    np.testing.assert_array_equal(ptr.cpu().numpy(), CT.indptr)
    np.testing.assert_array_equal(idx.cpu().numpy(), CT.indices)
    np.testing.assert_array_equal(val.cpu().numpy(), CT.data.astype(np.float32))


def test_entries_outside_the_matrix_are_reported_not_written():
    K = _kernels()
    rows = torch.tensor([0, 4, 2], dtype=torch.int64).cuda()
    cols = torch.tensor([0, 0, 1], dtype=torch.int64).cuda()
    with pytest.raises(IndexError):
        K.coo_to_csr(rows, cols, torch.ones(3).cuda(), 4, 2)
    with pytest.raises(IndexError):
        K.coo_to_csr(cols, rows, torch.ones(3).cuda(), 2, 4)
    ptr, idx, val = K.coo_to_csr(rows[:0], cols[:0], torch.ones(0).cuda(), 3, 2)      # nothing stored
    assert ptr.tolist() == [0, 0, 0, 0] and idx.numel() == 0
