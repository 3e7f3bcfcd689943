"""small host-side helpers (counterpart of rankfm/utils.py:5-18)"""


def get_data(obj):
    """the ndarray underneath a DataFrame / Series, an ndarray unchanged, TypeError for anything else"""
    kind = obj.__class__.__name__
    if kind in ('DataFrame', 'Series'):
        return obj.values
    if kind == 'ndarray':
        return obj
    raise TypeError("input data must be in either pd.dataframe/pd.series or np.ndarray format")
