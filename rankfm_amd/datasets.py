"""Loaders for the data BASELINE.json's config 1 names.  MovieLens-1M is not shipped (no network in the build or GPU boxes): when a
`ratings.dat` of the GroupLens ml-1m archive is supplied, `load_movielens_1m` reads it exactly as the reference's example
notebook does (examples/movielens.ipynb: `UserID::MovieID::Rating::Timestamp`, every rating an implicit interaction, 75 / 25 random
split); otherwise callers fall back to the planted MovieLens-1M-shaped surrogate (rankfm_amd.synthetic.make_planted)."""
import os

import numpy as np
import pandas as pd

ML1M_ENV = "RANKFM_ML1M"          # path of ratings.dat (or of the directory holding it)


def find_movielens_1m(path=None):
    """path of a usable ratings.dat, or None: the argument, $RANKFM_ML1M, ./data/ml-1m/ratings.dat, ./ml-1m/ratings.dat"""
    cands = [path, os.environ.get(ML1M_ENV), os.path.join("data", "ml-1m", "ratings.dat"), os.path.join("ml-1m", "ratings.dat")]
    for c in cands:
        if not c:
            continue
        if os.path.isdir(c):
            c = os.path.join(c, "ratings.dat")
        if os.path.isfile(c):
            return c
    return None


def load_movielens_1m(path=None, holdout=0.25, seed=1492):
    """dict(train, test: DataFrames [user_id, item_id]; n_users, n_items) from ratings.dat -- None when no file is found"""
    f = find_movielens_1m(path)
    if f is None:
        return None
    raw = pd.read_csv(f, sep="::", engine="python", header=None, names=["user_id", "item_id", "rating", "timestamp"],
                      usecols=[0, 1], dtype=np.int64)
    raw = raw.drop_duplicates()
    rng = np.random.default_rng(seed)
    mask = rng.random(len(raw)) < holdout
    train, test = raw[~mask].reset_index(drop=True), raw[mask].reset_index(drop=True)
    # evaluation needs users and items the model has seen (the reference's metrics skip the others the same way)
    test = test[test.user_id.isin(train.user_id) & test.item_id.isin(train.item_id)].reset_index(drop=True)
    return dict(train=train, test=test, n_users=int(raw.user_id.nunique()), n_items=int(raw.item_id.nunique()), path=f)
