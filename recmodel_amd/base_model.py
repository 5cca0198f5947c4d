"""``RecModel`` base class: the evaluation surface every model shares.

Host-side mirror of RecModel/base_model.py:34-179 for the WMF path.  ``eval_prec`` (the call
``WMF.train`` makes every iteration, wmf_model.py:163) is answered by the model's device kernel
through ``_eval_sums``; ``eval_topn`` / ``compute_hit`` keep the reference's NumPy RNG call
sequence so a seeded run samples the same negatives.
"""
from functools import partial
from multiprocessing import Pool

import numpy as np


def iter_rows_two_matrices(A, B):
    """Per row: (row, A data, A indices, B data, B indices).  base_model.py:10-20."""
    for row in range(A.shape[0]):
        a0, a1 = A.indptr[row], A.indptr[row + 1]
        b0, b1 = B.indptr[row], B.indptr[row + 1]
        yield row, A.data[a0:a1], A.indices[a0:a1], B.data[b0:b1], B.indices[b0:b1]


def iter_rows_mat(A):
    """Per row: (row, data, indices).  base_model.py:23-31."""
    for row in range(A.shape[0]):
        a0, a1 = A.indptr[row], A.indptr[row + 1]
        yield row, A.data[a0:a1], A.indices[a0:a1]


class RecModel:
    """Common evaluation scheme; subclasses implement train / predict / rank (base_model.py:34-49)."""

    def train(self):
        pass

    def predict(self, user_item):
        pass

    def rank(self, items, user, topn=None):
        pass

    # -------------------------------------------------------------- accuracy metrics
    def _eval_sums(self, utility_mat):
        """(sum sq err, sum abs err, count) over stored non-zero entries; subclasses provide it."""
        raise NotImplementedError

    def eval_prec(self, utility_mat, metric='mse'):
        """MSE / RMSE / MAE of predict() on the non-zero entries.  base_model.py:150-179."""
        metric = metric.upper()
        if metric not in ('MSE', 'RMSE', 'MAE'):
            raise ValueError("Metric {metric} is not implemented.")
        sq, ab, cnt = self._eval_sums(utility_mat)
        if cnt == 0:
            return float('nan')          # np.mean of an empty selection
        if metric == 'RMSE':
            return np.sqrt(sq / cnt)
        if metric == 'MSE':
            return sq / cnt
        return ab / cnt

    # -------------------------------------------------------------- sampled Recall@N
    def compute_hit(self, elem, rand_sampled, topn, dtype="float32"):
        """Hits of one user's test items among ``rand_sampled`` random candidates.
        base_model.py:51-98 (same np.random call order: candidates, then the slot)."""
        user, _, _, test_dat, test_idx = elem
        if len(test_dat) == 0:
            return np.zeros(topn.shape, dtype=dtype)
        candidates = np.random.randint(0, self.num_items, size=(rand_sampled + 1))
        slot = np.random.randint(0, rand_sampled - (2 * topn.max()))
        hits = np.zeros(topn.shape, dtype=dtype)
        for item in test_idx:
            candidates[slot] = item
            best = self.rank(items=candidates, users=user, topn=topn.max())
            for pos in range(len(topn)):
                if item in best[:topn[pos]]:
                    hits[pos] += 1
        return hits

    def _hit_counts(self, pair_user, pair_item, pair_row, candidates, slot, topn):
        """Hook for a device implementation of compute_hit over all test entries at once: entry p is
        (pair_user[p], pair_item[p]) with candidate row ``candidates[pair_row[p]]`` whose position
        ``slot[pair_row[p]]`` holds the test item.  Returns hits per topn, or None when not provided."""
        return None

    def eval_topn(self, test_mat, train_mat=None, eval_mat=None, topn=[10], rand_sampled=1000, cores=1,
                  random_state=None, dtype='float32'):
        """Recall@N with sampled negatives.  base_model.py:100-148.

        The random candidates are drawn on the host exactly as compute_hit draws them (users in row order,
        only users with test entries; candidates first, then the slot), so a seeded call samples what the
        reference samples.  Models that provide ``_hit_counts`` then rank every test entry in one device
        launch (``cores`` has no meaning there); others go through ``rank`` entry by entry like the reference."""
        super_mat = test_mat
        if train_mat is not None:
            super_mat += train_mat
        if eval_mat is not None:
            super_mat += eval_mat
        if random_state is not None:
            np.random.seed(random_state)
        if not isinstance(topn, np.ndarray):
            raise ValueError("Topn has to be a np.array")
        n_test = len(test_mat.nonzero()[0])
        if type(self)._hit_counts is RecModel._hit_counts:
            hits = np.zeros(topn.shape, dtype=dtype)
            if cores == 1:
                for elem in iter_rows_two_matrices(super_mat, test_mat):
                    hits += self.compute_hit(elem, rand_sampled=rand_sampled, topn=topn)
            else:
                with Pool(cores) as pool:
                    fn = partial(self.compute_hit, rand_sampled=rand_sampled, topn=topn)
                    hits = np.stack(pool.map(fn, iter_rows_two_matrices(super_mat, test_mat))).sum(axis=0)
        else:
            per_row = np.diff(test_mat.indptr)
            rows = np.flatnonzero(per_row)
            cand = np.empty((len(rows), rand_sampled + 1), dtype=np.int32)
            slot = np.empty(len(rows), dtype=np.int32)
            for k in range(len(rows)):                            # the draws of compute_hit, in its order
                cand[k] = np.random.randint(0, self.num_items, size=(rand_sampled + 1))
                slot[k] = np.random.randint(0, rand_sampled - (2 * topn.max()))
            pair_user = np.repeat(np.arange(test_mat.shape[0]), per_row)
            pair_row = np.repeat(np.arange(len(rows)), per_row[rows])
            hits = np.asarray(self._hit_counts(pair_user, test_mat.indices, pair_row, cand, slot, topn)).astype(dtype)
        recall = hits / n_test
        return {f"Recall@{topn[pos]}": recall[pos] for pos in range(len(topn))}
