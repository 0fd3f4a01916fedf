from functools import partial


def multi_apply(func, *args, **kwargs):
    """mmdet.core.utils.misc.multi_apply: map func over the per-level lists, transpose the result tuples into lists."""
    pfunc = partial(func, **kwargs) if kwargs else func
    map_results = map(pfunc, *args)
    return tuple(map(list, zip(*map_results)))
