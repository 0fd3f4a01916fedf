"""TEST INFRASTRUCTURE.  Stand-in for the un-vendored ``torch_scatter`` dependency of the reference's point_pillar.py
(README.md:63, built for torch 1.11+cu102; absent from this image and from /root/reference) so that the reference module can
be imported unmodified for oracle pinning.  Restates the published semantics of the two functions the reference calls
(point_pillar.py:6,32,61) - parity for torch_scatter ITSELF is therefore unpinned:

* ``scatter_mean(src, index, dim=0)``: out[i] = sum of src rows with index == i, divided by max(count, 1); size = index.max()+1.
* ``scatter_max(src, index, dim=0)``: (out, arg): out[i] = max over the rows (0 for an empty segment), arg = a row attaining it
  (torch_scatter's CPU kernel updates on strict '>' in row order, i.e. the first such row; ``src.size(dim)`` for empty segments).
"""
import torch


def _size(index):
    return int(index.max()) + 1 if index.numel() else 0


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    n = dim_size if dim_size is not None else _size(index)
    tot = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).index_add_(0, index, src)
    cnt = torch.zeros(n, dtype=src.dtype, device=src.device).index_add_(0, index, torch.ones_like(index, dtype=src.dtype)).clamp_(min=1)
    return tot / cnt.view((n,) + (1,) * (src.dim() - 1))


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0 and out is None
    n = dim_size if dim_size is not None else _size(index)
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    res = torch.zeros((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device).scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    rows = torch.arange(src.shape[0], device=src.device).view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    cand = torch.where(src == res[index], rows, torch.full_like(rows, src.shape[0]))
    arg = torch.full(res.shape, src.shape[0], dtype=torch.long, device=src.device).scatter_reduce(0, idx, cand, reduce="amin", include_self=True)
    return res, arg
