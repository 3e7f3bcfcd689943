"""Hold-out ranking metrics with the reference's definitions (rankfm/evaluation.py:9-175).

Each metric ranks with `model.recommend(test_users, n_items=k, filter_previous=..., cold_start='drop')` -- i.e. on the
device through `_recommend` -- and then scores the top-k lists against each user's held-out item set.  The set
logic is vectorised over one (user, rank) membership matrix instead of per-user Python set operations; the
numbers are the reference's (pinned by tests/golden/api_*.npz).
"""
import numpy as np
import pandas as pd

from .utils import get_data


def _relevance(model, test_interactions, k, filter_previous):
    """(hits [n_users, k] bool, n_test_items [n_users], recs DataFrame) for test users also seen in training"""
    assert model.is_fit, "you must fit the model prior to evaluating hold-out metrics"
    test = pd.DataFrame(get_data(test_interactions), columns=['user_id', 'item_id'])
    # first-appearance order of users, as dict(groupby(...)) keys come out sorted: the means below are order-free
    test_users = pd.unique(test['user_id'])
    recs = model.recommend(users=test_users, n_items=k, filter_previous=filter_previous, cold_start='drop')
    users = recs.index.values
    if len(users) == 0:
        return np.zeros((0, k), dtype=bool), np.zeros(0), recs
    test = test[test['user_id'].isin(users)].drop_duplicates()
    pos = pd.Series(np.arange(len(users)), index=users)
    long = pd.DataFrame({'row': np.repeat(np.arange(len(users)), recs.shape[1]),
                         'rank': np.tile(np.arange(recs.shape[1]), len(users)),
                         'item_id': recs.values.reshape(-1)})
    held = pd.DataFrame({'row': pos.loc[test['user_id'].values].values, 'item_id': test['item_id'].values, 'hit': True})
    merged = long.merge(held, on=['row', 'item_id'], how='left')
    hits = merged['hit'].notna().values.reshape(len(users), recs.shape[1])
    n_test = held.groupby('row')['item_id'].nunique().reindex(np.arange(len(users)), fill_value=0).values
    return hits, n_test, recs


def hit_rate(model, test_interactions, k=10, filter_previous=False):
    """proportion of test users with any relevant item in their top-k (rankfm/evaluation.py:9-33)"""
    hits, _, _ = _relevance(model, test_interactions, k, filter_previous)
    return float(np.mean(hits.any(axis=1)))


def reciprocal_rank(model, test_interactions, k=10, filter_previous=False):
    """mean inverse rank of the first relevant recommendation, 0 when none (rankfm/evaluation.py:36-61)"""
    hits, _, _ = _relevance(model, test_interactions, k, filter_previous)
    first = np.where(hits.any(axis=1), hits.argmax(axis=1) + 1, 0)
    return float(np.mean(np.where(first > 0, 1.0 / np.maximum(first, 1), 0.0)))


def discounted_cumulative_gain(model, test_interactions, k=10, filter_previous=False):
    """mean sum over relevant ranks r (0-based) of 1/log2(r+2) (rankfm/evaluation.py:64-89)"""
    hits, _, _ = _relevance(model, test_interactions, k, filter_previous)
    gains = 1.0 / np.log2(np.arange(hits.shape[1]) + 2)
    return float(np.mean((hits * gains).sum(axis=1)))


def precision(model, test_interactions, k=10, filter_previous=False):
    """mean fraction of the top-k that is relevant (rankfm/evaluation.py:92-116)"""
    hits, _, recs = _relevance(model, test_interactions, k, filter_previous)
    # the reference divides the size of the SET intersection by k: repeated recommendations cannot occur
    return float(np.mean(hits.sum(axis=1) / recs.shape[1]))


def recall(model, test_interactions, k=10, filter_previous=False):
    """mean fraction of each user's held-out items found in the top-k (rankfm/evaluation.py:119-143)"""
    hits, n_test, _ = _relevance(model, test_interactions, k, filter_previous)
    return float(np.mean(hits.sum(axis=1) / n_test))


def diversity(model, test_interactions, k=10, filter_previous=False):
    """count / share of test users that get each training item recommended (rankfm/evaluation.py:146-175)"""
    assert model.is_fit, "you must fit the model prior to evaluating hold-out metrics"
    test = pd.DataFrame(get_data(test_interactions), columns=['user_id', 'item_id'])
    recs = model.recommend(users=test['user_id'].unique(), n_items=k, filter_previous=filter_previous, cold_start='drop')
    counts = pd.Series(recs.values.reshape(-1)).value_counts()
    out = pd.DataFrame({'item_id': model.item_id.values})
    out['cnt_users'] = counts.reindex(out['item_id'].values, fill_value=0).values
    out = out.sort_values('cnt_users', ascending=False, kind='stable').reset_index(drop=True)
    out['pct_users'] = out['cnt_users'] / len(recs)
    return out
