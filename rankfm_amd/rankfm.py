"""RankFM on the MI355X engine: the host-side mirror of the reference's model class.

Same public surface as rankfm.RankFM (rankfm/rankfm.py:11-454) -- constructor arguments and their validation,
fit / fit_partial / predict / recommend / similar_items / similar_users, and the attribute names, dtypes and
layouts of the data plumbing products (`interactions`, `sample_weight`, `user_items`, `x_uf`, `x_if`) and of the
model weights (`w_i, w_if, v_u, v_i, v_uf, v_if`, C-contiguous float32 numpy arrays) -- so a fitted model is
interchangeable with the reference's.  Training, scoring and ranking run through the private operator
boundary `_fit/_predict/_recommend` (rankfm_amd/_rankfm.py -> include/rankfm_hip.h), exactly where the
reference calls its Cython module (rankfm/rankfm.py:304-324, 347-357, 381-394).

Written for scale: id->index mapping, the per-user item lists (CSR, not a dict of arrays) and the
`fit_partial` merge are vectorised; the reference's per-user Python loops (rankfm/rankfm.py:170-174,
rankfm/_rankfm.pyx:201-212) are O(N) interpreter work.
"""
import numpy as np
import pandas as pd

from ._rankfm import _similar, DEFAULT_ENGINE, EngineOptions, UserItemsCSR, _fit, _predict, _recommend
from .utils import get_data


class RankFM():
    """Factorization Machines for Ranking Problems with Implicit Feedback Data (MI355X engine)"""

    def __init__(self, factors=10, loss='bpr', max_samples=10, alpha=0.01, beta=0.1, sigma=0.1, learning_rate=0.1,
                 learning_schedule='constant', learning_exponent=0.25, engine=None):
        """store hyperparameters and initialize internal model state

        The nine modelling arguments, their defaults and their validation are the reference's
        (rankfm/rankfm.py:14-38).  `engine` (rankfm_amd.EngineOptions, optional) selects how the SGD loop runs on
        the device; the default is the Hogwild production mode, `rankfm_amd.REFERENCE_ENGINE` reproduces the
        reference's sequential arithmetic on one wavefront.
        """
        assert isinstance(factors, int) and factors >= 1, "[factors] must be a positive integer"
        assert isinstance(loss, str) and loss in ('bpr', 'warp'), "[loss] must be in ('bpr', 'warp')"
        assert isinstance(max_samples, int) and max_samples > 0, "[max_samples] must be a positive integer"
        assert isinstance(alpha, float) and alpha > 0.0, "[alpha] must be a positive float"
        assert isinstance(beta, float) and beta > 0.0, "[beta] must be a positive float"
        assert isinstance(sigma, float) and sigma > 0.0, "[sigma] must be a positive float"
        assert isinstance(learning_rate, float) and learning_rate > 0.0, "[learning_rate] must be a positive float"
        assert isinstance(learning_schedule, str) and learning_schedule in ('constant', 'invscaling'), \
            "[learning_schedule] must be in ('constant', 'invscaling')"
        assert isinstance(learning_exponent, float) and learning_exponent > 0.0, "[learning_exponent] must be a positive float"
        assert engine is None or isinstance(engine, EngineOptions), "[engine] must be a rankfm_amd.EngineOptions"

        self.factors = factors
        self.loss = loss
        self.max_samples = max_samples
        self.alpha = alpha
        self.beta = beta
        self.sigma = sigma
        self.learning_rate = learning_rate
        self.learning_schedule = learning_schedule
        self.learning_exponent = learning_exponent
        self.engine = engine if engine is not None else DEFAULT_ENGINE
        self._reset_state()

    # ------------------------------------------------------------------ state

    def _reset_state(self):
        """initialize or reset internal model state (attribute set of rankfm/rankfm.py:60-97)"""
        self.user_id = self.item_id = None
        self.user_idx = self.item_idx = None
        self.index_to_user = self.index_to_item = None
        self.user_to_index = self.item_to_index = None
        self.interactions = None
        self.sample_weight = None
        self.user_items = None
        self.x_uf = self.x_if = None
        self.w_i = self.w_if = None
        self.v_u = self.v_i = self.v_uf = self.v_if = None
        self.is_fit = False
        self.epochs_trained = 0          # absolute epoch counter (keys the engine's counter RNG across fit_partial calls)
        self.last_fit_report = None

    def _init_all(self, interactions, user_features=None, item_features=None, sample_weight=None):
        """index the interaction data and user/item features and initialize model weights (rankfm/rankfm.py:100-137)"""
        assert isinstance(interactions, (np.ndarray, pd.DataFrame)), "[interactions] must be np.ndarray or pd.dataframe"
        assert interactions.shape[1] == 2, "[interactions] should be: [user_id, item_id]"

        self._init_ids(interactions)
        self._init_interactions(interactions, sample_weight)
        self._init_features(user_features, item_features)
        self._init_weights(user_features, item_features)

    def _init_ids(self, interactions):
        """the identifier <-> index maps: sorted unique identifiers, zero-based index = rank of the identifier
        (rankfm/rankfm.py:113-127)"""
        data = get_data(interactions)
        self._set_ids(np.sort(pd.unique(data[:, 0])), np.sort(pd.unique(data[:, 1])))

    def _set_ids(self, users_sorted, items_sorted):
        """install the identifier <-> index maps from the sorted unique identifiers (fit_distributed builds those rank-locally)"""
        self.user_id = pd.Series(users_sorted)
        self.item_id = pd.Series(items_sorted)
        self.index_to_user = self.user_id
        self.index_to_item = self.item_id
        self.user_to_index = pd.Series(data=self.index_to_user.index, index=self.index_to_user.values)
        self.item_to_index = pd.Series(data=self.index_to_item.index, index=self.index_to_item.values)
        self.user_idx = np.arange(len(self.user_id), dtype=np.int32)
        self.item_idx = np.arange(len(self.item_id), dtype=np.int32)

    def _lookup(self, values, which):
        """identifier -> index, -1 where unknown (vectorised counterpart of Series.map(user_to_index))"""
        index = pd.Index(self.index_to_user.values if which == 'user' else self.index_to_item.values)
        return index.get_indexer(pd.Index(values))

    def _index_pairs(self, interactions, sample_weight):
        """identifiers -> int32 [N,2] index pairs (and self.sample_weight); no per-user item lists yet"""
        data = get_data(interactions)
        u = self._lookup(data[:, 0], 'user')
        i = self._lookup(data[:, 1], 'item')
        if (u < 0).any() or (i < 0).any():
            # the reference casts the mapped (NaN-holding) column to int32 before its dropna(), which raises
            raise ValueError("[interactions] contains users/items not present in the data the model was first fit on")
        pairs = np.empty((len(u), 2), dtype=np.int32)
        pairs[:, 0] = u
        pairs[:, 1] = i
        if sample_weight is not None:
            assert isinstance(sample_weight, (np.ndarray, pd.Series)), "[sample_weight] must be np.ndarray or pd.series"
            assert sample_weight.ndim == 1, "[sample_weight] must a vector (ndim=1)"
            assert len(sample_weight) == len(interactions), "[sample_weight] must have the same length as [interactions]"
            self.sample_weight = np.ascontiguousarray(get_data(sample_weight), dtype=np.float32)
        else:
            self.sample_weight = np.ones(len(pairs), dtype=np.float32)
        return pairs

    def _init_interactions(self, interactions, sample_weight):
        """map new interaction data to existing internal user/item indexes (rankfm/rankfm.py:140-177)"""
        assert isinstance(interactions, (np.ndarray, pd.DataFrame)), "[interactions] must be np.ndarray or pd.dataframe"
        assert interactions.shape[1] == 2, "[interactions] should be: [user_id, item_id]"

        pairs = self._index_pairs(interactions, sample_weight)
        n_users = len(self.user_idx)
        if self.is_fit:
            # extend each user's item set with the new observations (set union, so duplicates collapse);
            # users without new rows keep their list (the reference raises KeyError for them, rankfm/rankfm.py:172)
            old = self.user_items
            old_u = np.repeat(np.arange(n_users, dtype=np.int64), np.diff(old.offsets))
            both = np.unique(np.concatenate([old_u * len(self.item_idx) + old.items,
                                             pairs[:, 0].astype(np.int64) * len(self.item_idx) + pairs[:, 1]]))
            self.user_items = UserItemsCSR.from_pairs(both // len(self.item_idx), both % len(self.item_idx), n_users)
        else:
            # first fit: every observed row is kept, sorted by item within user (duplicates stay, like the reference)
            self.user_items = UserItemsCSR.from_pairs(pairs[:, 0], pairs[:, 1], n_users)
        self.interactions = np.ascontiguousarray(pairs, dtype=np.int32)

    def _index_features(self, features, to_index, idx, who):
        frame = pd.DataFrame(features.copy())
        frame = frame.set_index(frame.columns[0])
        frame.index = frame.index.map(to_index)
        if np.array_equal(sorted(frame.index.values), idx):
            return np.ascontiguousarray(frame.sort_index(), dtype=np.float32)
        raise KeyError('the %ss in [%s_features] do not match the %ss in [interactions]' % (who, who, who))

    def _init_features(self, user_features=None, item_features=None):
        """dense float32 feature matrices row-ordered by index, or the all-zero [U,1] / [I,1] placeholders
        (rankfm/rankfm.py:181-211)"""
        if user_features is not None:
            self.x_uf = self._index_features(user_features, self.user_to_index, self.user_idx, 'user')
        else:
            self.x_uf = np.zeros([len(self.user_idx), 1], dtype=np.float32)
        if item_features is not None:
            self.x_if = self._index_features(item_features, self.item_to_index, self.item_idx, 'item')
        else:
            self.x_if = np.zeros([len(self.item_idx), 1], dtype=np.float32)

    def _init_weights(self, user_features=None, item_features=None):
        """zero scalar weights, N(0, sigma) factors, N(0, (alpha/beta) sigma) feature factors when features are given
        (rankfm/rankfm.py:214-244).  Draw order v_u, v_i, v_uf, v_if from numpy's global RNG, like the reference,
        so np.random.seed(s) gives the reference's initial weights."""
        self.w_i = np.zeros(len(self.item_idx)).astype(np.float32)
        self.w_if = np.zeros(self.x_if.shape[1]).astype(np.float32)
        self.v_u = np.random.normal(loc=0, scale=self.sigma, size=(len(self.user_idx), self.factors)).astype(np.float32)
        self.v_i = np.random.normal(loc=0, scale=self.sigma, size=(len(self.item_idx), self.factors)).astype(np.float32)
        scale = (self.alpha / self.beta) * self.sigma
        if user_features is not None:
            self.v_uf = np.random.normal(loc=0, scale=scale, size=[self.x_uf.shape[1], self.factors]).astype(np.float32)
        else:
            self.v_uf = np.zeros([self.x_uf.shape[1], self.factors], dtype=np.float32)
        if item_features is not None:
            self.v_if = np.random.normal(loc=0, scale=scale, size=[self.x_if.shape[1], self.factors]).astype(np.float32)
        else:
            self.v_if = np.zeros([self.x_if.shape[1], self.factors], dtype=np.float32)

    # ------------------------------------------------------------------ public API

    def fit(self, interactions, user_features=None, item_features=None, sample_weight=None, epochs=1, verbose=False):
        """clear previous model state and learn new model weights using the input data (rankfm/rankfm.py:252-266)"""
        self._reset_state()
        self.fit_partial(interactions, user_features, item_features, sample_weight, epochs, verbose)
        return self

    def fit_partial(self, interactions, user_features=None, item_features=None, sample_weight=None, epochs=1, verbose=False):
        """learn or update model weights resuming from the current model state (rankfm/rankfm.py:269-327)"""
        assert isinstance(epochs, int) and epochs >= 1, "[epochs] must be a positive integer"
        assert isinstance(verbose, bool), "[verbose] must be a boolean value"

        if self.is_fit:
            self._init_interactions(interactions, sample_weight)
            self._init_features(user_features, item_features)
        else:
            self._init_all(interactions, user_features, item_features, sample_weight)

        # 'bpr' is the one-draw case of the WARP loop (rankfm/rankfm.py:294-299)
        if self.loss == 'bpr':
            max_samples = 1
        elif self.loss == 'warp':
            max_samples = self.max_samples
        else:
            raise ValueError('[loss] function not recognized')

        report = {}
        # the reference's schedule restarts at epoch 0 on every call (rankfm/_rankfm.pyx:218-223): keep that for eta
        # (epoch_begin stays 0), but key the counter RNG and the keyed order by the absolute epoch so that resumed
        # training with a fixed engine seed does not replay the first call's order and draws
        _fit(self.interactions, self.sample_weight, self.user_items, self.x_uf, self.x_if,
             self.w_i, self.w_if, self.v_u, self.v_i, self.v_uf, self.v_if,
             self.alpha, self.beta, self.learning_rate, self.learning_schedule, self.learning_exponent,
             max_samples, epochs, verbose, engine=self.engine, rng_epoch_offset=self.epochs_trained, report=report)
        self.last_fit_report = report
        self.epochs_trained += epochs
        self.is_fit = True
        return self

    def predict(self, pairs, cold_start='nan'):
        """predicted pointwise utilities for all (user, item) pairs (rankfm/rankfm.py:330-364)"""
        assert isinstance(pairs, (np.ndarray, pd.DataFrame)), "[pairs] must be np.ndarray or pd.dataframe"
        assert pairs.shape[1] == 2, "[pairs] should be: [user_id, item_id]"
        assert self.is_fit, "you must fit the model prior to generating predictions"

        data = get_data(pairs)
        idx = np.empty((len(data), 2), dtype=np.float32)
        u = self._lookup(data[:, 0], 'user')
        i = self._lookup(data[:, 1], 'item')
        idx[:, 0] = np.where(u < 0, np.nan, u)
        idx[:, 1] = np.where(i < 0, np.nan, i)
        scores = _predict(idx, self.x_uf, self.x_if, self.w_i, self.w_if, self.v_u, self.v_i, self.v_uf, self.v_if,
                          device=self.engine.device)
        if cold_start == 'nan':
            return scores
        elif cold_start == 'drop':
            return scores[~np.isnan(scores)]
        else:
            raise ValueError("param [cold_start] must be set to either 'nan' or 'drop'")

    def recommend(self, users, n_items=10, filter_previous=False, cold_start='nan'):
        """topN items for each user as a DataFrame indexed by user (rankfm/rankfm.py:367-402)"""
        assert getattr(users, '__iter__', False), "[users] must be an iterable (e.g. list, array, series)"
        assert self.is_fit, "you must fit the model prior to generating recommendations"

        users = list(users) if not isinstance(users, (np.ndarray, pd.Series, pd.Index, list)) else users
        u = self._lookup(np.asarray(pd.Series(users).values), 'user')
        user_idx = np.ascontiguousarray(np.where(u < 0, np.nan, u), dtype=np.float32)
        rec = _recommend(user_idx, self.user_items, n_items, filter_previous, self.x_uf, self.x_if, self.w_i, self.w_if,
                         self.v_u, self.v_i, self.v_uf, self.v_if, device=self.engine.device)
        # index -> identifier, NaN rows stay NaN
        ids = self.index_to_item.values
        known = ~np.isnan(rec)
        out = np.empty(rec.shape, dtype=object if ids.dtype == object else np.float64 if ids.dtype.kind in 'iuf' else object)
        out[:] = np.nan
        out[known] = ids[rec[known].astype(np.int64)]
        rec_items = pd.DataFrame(out, index=users)
        if ids.dtype.kind in 'iu' and known.all():
            rec_items = rec_items.astype(ids.dtype)
        if cold_start == 'nan':
            return rec_items
        elif cold_start == 'drop':
            return rec_items.dropna(how='any')
        else:
            raise ValueError("param [cold_start] must be set to either 'nan' or 'drop'")

    # ------------------------------------------------------------------ persistence

    _WEIGHTS = ("w_i", "w_if", "v_u", "v_i", "v_uf", "v_if")

    def save(self, path):
        """write the fitted model to one .npz: hyper-parameters, id maps, the reference's weight layout (float32, C order),
        features and the per-user item lists.  The reference has no serialisation; its models are plain attributes, and the
        arrays written here are exactly those attributes, so they can be assigned onto a reference `RankFM` as well."""
        assert self.is_fit, "you must fit the model prior to saving it"
        hyper = dict(factors=self.factors, loss=self.loss, max_samples=self.max_samples, alpha=self.alpha, beta=self.beta,
                     sigma=self.sigma, learning_rate=self.learning_rate, learning_schedule=self.learning_schedule,
                     learning_exponent=self.learning_exponent)
        def plain(ids):          # object arrays (mixed frames) hold either python ints or strings
            if ids.dtype != object:
                return ids
            return ids.astype(np.int64) if all(isinstance(x, (int, np.integer)) for x in ids) else ids.astype("U")
        ids_u, ids_i = plain(self.user_id.values), plain(self.item_id.values)
        np.savez_compressed(
            path, format=np.array("rankfm_amd/1"), hyper=np.array(repr(hyper)), epochs_trained=np.int64(self.epochs_trained),
            user_id=ids_u, item_id=ids_i,
            x_uf=self.x_uf, x_if=self.x_if, csr_offsets=self.user_items.offsets, csr_items=self.user_items.items,
            **{k: getattr(self, k) for k in self._WEIGHTS})

    @classmethod
    def load(cls, path, engine=None):
        """rebuild a fitted model from `save()` output; predict / recommend / fit_partial work on it immediately"""
        import ast
        z = np.load(path if str(path).endswith(".npz") else str(path) + ".npz", allow_pickle=False)
        assert str(z["format"]) == "rankfm_amd/1", "not a rankfm_amd model file"
        m = cls(engine=engine, **ast.literal_eval(str(z["hyper"])))
        m.user_id, m.item_id = pd.Series(z["user_id"]), pd.Series(z["item_id"])
        if m.user_id.dtype.kind == "U":
            m.user_id = m.user_id.astype(object)
        if m.item_id.dtype.kind == "U":
            m.item_id = m.item_id.astype(object)
        m.index_to_user, m.index_to_item = m.user_id, m.item_id
        m.user_to_index = pd.Series(data=m.index_to_user.index, index=m.index_to_user.values)
        m.item_to_index = pd.Series(data=m.index_to_item.index, index=m.index_to_item.values)
        m.user_idx = np.arange(len(m.user_id), dtype=np.int32)
        m.item_idx = np.arange(len(m.item_id), dtype=np.int32)
        m.x_uf, m.x_if = np.ascontiguousarray(z["x_uf"]), np.ascontiguousarray(z["x_if"])
        m.user_items = UserItemsCSR(z["csr_offsets"], z["csr_items"])
        for k in cls._WEIGHTS:
            setattr(m, k, np.ascontiguousarray(z[k], dtype=np.float32))
        m.epochs_trained = int(z["epochs_trained"])
        m.is_fit = True
        return m

    def similar_items(self, item_id, n_items=10):
        """most similar items wrt latent factor space representation (rankfm/rankfm.py:405-428), ranked on the device"""
        assert item_id in self.item_id.values, "you must select an [item_id] present in the training data"
        assert self.is_fit, "you must fit the model prior to generating similarities"
        idx = int(self.item_to_index.loc[item_id])
        n = min(int(n_items), len(self.item_idx) - 1)
        order = _similar(0, idx, n, self.x_uf, self.x_if, self.w_i, self.w_if, self.v_u, self.v_i, self.v_uf, self.v_if,
                         device=self.engine.device).astype(np.int64)
        return self.index_to_item.values[order]

    def similar_users(self, user_id, n_users=10):
        """most similar users wrt latent factor space representation (rankfm/rankfm.py:431-454), ranked on the device"""
        assert user_id in self.user_id.values, "you must select an [user_id] present in the training data"
        assert self.is_fit, "you must fit the model prior to generating similarities"
        idx = int(self.user_to_index.loc[user_id])
        n = min(int(n_users), len(self.user_idx) - 1)
        order = _similar(1, idx, n, self.x_uf, self.x_if, self.w_i, self.w_if, self.v_u, self.v_i, self.v_uf, self.v_if,
                         device=self.engine.device).astype(np.int64)
        return self.index_to_user.values[order]
