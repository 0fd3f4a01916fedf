"""Functional wrappers over the C ABI (include/transfuser_hip.h).

No autograd here: every function launches HIP kernels on torch's current stream and returns
tensors; ``transfuser_amd/functions.py`` composes them into a handful of block-level
``torch.autograd.Function``s.  Layout conventions: feature maps are explicit NHWC tensors
``(B, H, W, C)``, conv weights are ``(Cout, Cin/g, kh, kw)`` parameters in channels_last memory
format (physically ``(Cout, kh, kw, Cin/g)``), linear weights are ``(out, in)``.
"""
import ctypes

import os
import sys

import torch

from . import _lib
from ._lib import ConvGeom, GemmDesc, c_p, check, ptr, stream_of

byref = ctypes.byref

_ws_cache = {}


def L():
    return _lib.load()


def workspace(device):
    """Reduction scratch: one per (device, stream) - kernels on one stream serialise, so each stream shares one buffer."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if getattr(device, "type", str(device)[:4]) == "cuda" else 0)
    ws = _ws_cache.get(key)
    if ws is None:
        L().tf_workspace_bytes.restype = ctypes.c_long
        # zero-initialised ONCE: the arrival counters of the fused reduce + finalize kernels live at its end and reset themselves
        ws = torch.zeros(L().tf_workspace_bytes() // 4, dtype=torch.float32, device=device)
        _ws_cache[key] = ws
    return ws


# ---- pair launch (csrc/gemm_pair.cpp): the weight gradient and the input gradient of a layer both read dy and write disjoint outputs; inside
# ``gemm_pair()`` the (up to two) eligible GEMMs are held back by the library and issued as ONE grid when the block closes, so the second
# problem's workgroups start as the first one's retire (no drain / dispatch gap / ramp between the two launches, no cross-queue edge).
# Weight gradients on a side stream - the other way to overlap them - were measured at +3.7 ms/step in rounds 1 and 4 (~10 us per cross-queue
# dependency) and are gone.
GEMM_PAIR = os.environ.get("TF_GEMM_PAIR", "1") != "0"      # A/B switch of the round-5 pair launch
_pair_open = [False]
_pair_shapes = []      # (m, n, k, batch, flops) of the calls inside the open bracket (census only)
_AB_MERGE_HEADS = os.environ.get("TF_AB_MERGE_HEADS", "1") != "0"      # same-lease A/B switches of round 5 (tools/final_evidence_r05.sh: "round-4 behaviour of the same library")
_AB_WSUM = os.environ.get("TF_AB_WSUM", "1") != "0"


class gemm_pair:
    """with gemm_pair(t): linear_wgrad(...); dx = linear_dgrad(...)   (t: any tensor on the device / stream the calls use)"""

    def __init__(self, t):
        self.t = t
        self.on = False

    def __enter__(self):
        self.on = GEMM_PAIR and not _CHECK and not _pair_open[0] and (self.t.is_cuda or _lib.is_test_backend())
        if self.on:
            self._e = _census_begin()          # the census times the bracket as ONE entry (the held calls launch when it closes)
            self._n0 = gemm_pair_count() if self._e is not None else 0
            _pair_shapes.clear()
            check(L().tf_gemm_pair_begin(), "tf_gemm_pair_begin")
            _pair_open[0] = True
        return self

    def __exit__(self, et, ev, tb):
        if self.on:
            _pair_open[0] = False
            rc = L().tf_gemm_pair_end(stream_of(self.t))
            if et is None:
                check(rc, "tf_gemm_pair_end")
                if self._e is not None and _pair_shapes:
                    joint = gemm_pair_count() > self._n0
                    _census_end(self._e, "gemm pair wgrad+dgrad (one grid)" if joint else "gemm wgrad+dgrad (two launches)", _pair_shapes[-1][:4], sum(v[4] for v in _pair_shapes))
        return False


def gemm_pair_count(singles=False):
    L().tf_gemm_pair_count.restype = ctypes.c_long
    return L().tf_gemm_pair_count(int(singles))


def autotune(enable):
    """cudnn.benchmark-style tiling search for the MFMA engine (eager warm-up only: it synchronises)."""
    check(L().tf_autotune(int(bool(enable))), "tf_autotune")


def set_precision(mode):
    """Compute precision of every MFMA-engine contraction (accumulation, epilogues, master weights, AdamW stay fp32 in all modes):
    "fp32"  exact fp32 MFMA (v_mfma_f32_32x32x2_f32; default - the reference's arithmetic);
    "f32x3" bf16x3 split: every fp32 operand is split exactly into three bf16 terms in registers and the six leading partial products run on
            the bf16 MFMA (16x the fp32 MFMA rate on gfx950) - fp32-ACCURATE (dropped terms <= 2^-26 |xy|), not a reduced-precision mode;
    "bf16"  bf16 operands (BASELINE configs[2]); "fp16" IEEE-half operands (configs[4]; use a loss scale: train.Engine does).
    In the two 16-bit modes the linear layers of the GPT fusion stages additionally run on 16-bit STORED operands (``lowp_storage()``;
    TF_STORE16=0 keeps every operand fp32 in memory and rounds in registers only): activations / gradients / weights are written once as
    bf16 / half matrices (``cast16``) and multiplied by ``gemm16_nt`` - half the operand bytes through L2 / LDS per MFMA flop."""
    m = {"fp32": 0, "f32": 0, 0: 0, "bf16": 1, 1: 1, "f32x3": 2, "fp32x3": 2, "bf16x3": 2, 2: 2, "fp16": 3, "f16": 3, "half": 3, 3: 3}[mode]
    check(L().tf_set_precision(m), "tf_set_precision")
    _lowp["dtype"] = {1: 1, 3: 2}.get(m, 0) if _STORE16 else 0
    _lowp["w"].clear()
    _lowp.pop("table", None)


def get_precision():
    return ("fp32", "bf16", "f32x3", "fp16")[L().tf_get_precision()]


# ---- 16-bit operand storage (csrc/cast16.cpp, tf_gemm16_nt_f32)
_STORE16 = os.environ.get("TF_STORE16", "1") != "0"
_lowp = {"dtype": 0, "managed": False, "w": {}}


def lowp_storage():
    """0: operands are fp32 in memory; 1: bf16 / 2: fp16 copies feed the packed-16 GEMMs of the GPT linear layers."""
    return _lowp["dtype"]


def _t16():
    return torch.bfloat16 if _lowp["dtype"] == 1 else torch.float16


def cast16(x, want=True, want_t=True, out=None, out_t=None):
    """x (rows, cols) fp32 (row stride may exceed cols) -> (x16 (rows, cols), x16t (cols, rows8)): the 16-bit copy and its TRANSPOSE (rows
    zero-padded to a multiple of 8), one pass.  Either may be skipped (None)."""
    rows, cols = x.shape
    assert x.stride(1) == 1
    rows8 = (rows + 7) // 8 * 8
    y = out if out is not None else (torch.empty(rows, cols, dtype=_t16(), device=x.device) if want else None)
    yt = out_t if out_t is not None else (torch.empty(cols, rows8, dtype=_t16(), device=x.device) if want_t else None)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)
    check(L().tf_cast16_f32(ptr(x), rows, cols, x.stride(0), vp(y), y.stride(0) if y is not None else 0, vp(yt), yt.stride(0) if yt is not None else 0,
                            _lowp["dtype"], stream_of(x)), "tf_cast16_f32")
    return y, yt


def gemm16_nt(a16, b16, out, bias=None, res=None, relu=False, accumulate=False, mask=None, alpha=1.0, k=None, kind=0):
    """out (m, n) fp32 (op)= alpha * a16 (m, k) @ b16 (n, k).T (+bias) (+res) (relu) (mask); 16-bit operands as cast16 writes them
    (k = the common, 8-aligned contraction length when the buffers are the zero-padded transposed copies).  kind (tests / tuning): pin an
    LDS-DMA tile configuration 1..8 (0 = the library's choice)."""
    m, n = out.shape
    k = k if k is not None else a16.shape[1]
    assert a16.shape[0] == m and b16.shape[0] == n and a16.shape[1] >= k and b16.shape[1] >= k and k % 8 == 0, (a16.shape, b16.shape, out.shape, k)
    _e = _census_begin()
    check(L().tf_gemm16_nt_f32(ctypes.c_void_p(a16.data_ptr()), ctypes.c_void_p(b16.data_ptr()), ptr(out), m, n, k, a16.stride(0), b16.stride(0), out.stride(0),
                               ptr(bias), ptr(res), res.stride(0) if res is not None else 0, ctypes.c_float(alpha), int(relu), int(accumulate), ptr(mask),
                               mask.stride(0) if mask is not None else 0, _lowp["dtype"] + 16 * int(kind), stream_of(out)), "tf_gemm16_nt_f32")
    _census_end(_e, "gemm16 nt", (m, n, k, 1), 2.0 * m * n * k)
    return out


def gemm16_nt_colstat(a16, b16, out, want_stat=True):
    """out = a16 @ b16.T (plain store) + the BatchNorm statistics of out from the epilogue (ColStat or None)."""
    m, n = out.shape
    k = a16.shape[1]
    cs = ColStat(m, n, out.device) if (want_stat and want_colstat(m)) else None
    if cs is None:
        return gemm16_nt(a16, b16, out), None
    _e = _census_begin()
    check(L().tf_gemm16_nt_colstat_f32(ctypes.c_void_p(a16.data_ptr()), ctypes.c_void_p(b16.data_ptr()), ptr(out), m, n, k, a16.stride(0), b16.stride(0), out.stride(0),
                                       _lowp["dtype"], ptr(cs.buf), byref(cs.nparts), stream_of(out)), "tf_gemm16_nt_colstat_f32")
    _census_end(_e, "gemm16 nt", (m, n, k, 1), 2.0 * m * n * k)
    return out, (cs if cs else None)


STORE16_CONV = os.environ.get("TF_STORE16_CONV", "1") != "0"      # round 5 (second session): the bottlenecks' 1x1 convolutions on 16-bit STORED operands whose copies their producers write


def lowp_conv():
    """16-bit storage mode of the RegNetY bottlenecks' 1x1 convolutions (functions.YBlockFn): 0 off, else the storage dtype code."""
    return _lowp["dtype"] if STORE16_CONV else 0


def _pair16(rows, C, dev, want=True):
    rows8 = (rows + 7) // 8 * 8
    return (torch.empty(rows, C, dtype=_t16(), device=dev) if want else None), torch.empty(C, rows8, dtype=_t16(), device=dev), rows8


_vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def bn_apply16(x, coef, res=None, relu=False, want_f32=True):
    """y = x * scale + shift (+ res) (relu) from coef = [scale | shift] (bn_finalize_parts) -> (y fp32 or None, y16 (rows, C), y16t (C, rows8))."""
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x) if want_f32 else None
    y16, y16t, rows8 = _pair16(rows, C, x.device)
    check(L().tf_bn_apply16_f32(ptr(_c(x)), ptr(coef), ptr(res), int(relu), ptr(y), rows, C, _vp(y16), _vp(y16t), rows8, _lowp["dtype"], stream_of(x)), "tf_bn_apply16_f32")
    return y, y16, y16t


def se_scale_bn16(y, coef, gate):
    """relu(y * scale + shift) * sigmoid(gate[b, c]) as 16-bit copies only -> (z16 (rows, C), z16t (C, rows8))."""
    B, H, W, C = y.shape
    z16, z16t, rows8 = _pair16(B * H * W, C, y.device)
    check(L().tf_se_scale_bn16_f32(ptr(_c(y)), ptr(coef), ptr(_c(gate)), B, H * W, C, _vp(z16), _vp(z16t), rows8, _lowp["dtype"], stream_of(y)), "tf_se_scale_bn16_f32")
    return z16, z16t


def bn_bwd16(dz, z, x, gamma, sm, si, dgamma, dbeta, want_dres=False, want_f32=False):
    """bn_bwd whose dx leaves as 16-bit copies -> (dx fp32 or None, dx16, dx16t, dres or None)."""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x) if want_f32 else None
    dres = torch.empty_like(x) if want_dres else None
    d16, d16t, rows8 = _pair16(rows, C, x.device)
    check(L().tf_bn_bwd16_f32(ptr(_c(dz)), ptr(z), ptr(_c(x)), rows, C, ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta),
                              ptr(workspace(x.device)), _vp(d16), _vp(d16t), rows8, _lowp["dtype"], stream_of(x)), "tf_bn_bwd16_f32")
    return dx, d16, d16t, dres


def bn_bwd_remask16(dz, x, coef, gamma, sm, si, dgamma, dbeta, want_f32=False):
    """bn_bwd_remask whose dx leaves as 16-bit copies -> (dx fp32 or None, dx16, dx16t)."""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x) if want_f32 else None
    d16, d16t, rows8 = _pair16(rows, C, x.device)
    check(L().tf_bn_bwd_remask16_f32(ptr(_c(dz)), ptr(_c(x)), ptr(coef), rows, C, ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dgamma), ptr(dbeta),
                                     ptr(workspace(x.device)), _vp(d16), _vp(d16t), rows8, _lowp["dtype"], stream_of(x)), "tf_bn_bwd_remask16_f32")
    return dx, d16, d16t


def lowp_weight(w):
    """(w16 (N, K), w16t (K, N8)) of a linear weight (N, K).  Inside train.Engine the copies are cached and rewritten once per step right after
    AdamW (``lowp_refresh_weights``); anywhere else they are re-made at every use (the parameter may have changed)."""
    key = (w.data_ptr(), tuple(w.shape))
    ent = _lowp["w"].get(key)
    if ent is None:
        y, yt = cast16(w.detach())
        _lowp["w"][key] = (w.detach(), y, yt)
        return y, yt
    if not _lowp["managed"]:
        cast16(ent[0], out=ent[1], out_t=ent[2])
    return ent[1], ent[2]


class _Cast16Item(ctypes.Structure):      # include/transfuser_hip.h: tf_cast16_item
    _fields_ = [("x", ctypes.c_void_p), ("y16", ctypes.c_void_p), ("y16t", ctypes.c_void_p), ("rows", ctypes.c_int), ("cols", ctypes.c_int),
                ("ldx", ctypes.c_int), ("ldy", ctypes.c_int), ("ldyt", ctypes.c_int), ("tile0", ctypes.c_int)]


CAST16_MULTI = os.environ.get("TF_CAST16_MULTI", "1") != "0"      # A/B switch of round 5: the weight copies in one launch


def _cast16_table():
    """Device-resident tf_cast16_item table over the cached weights (rebuilt only when the set of cached weights changes; built OUTSIDE graph
    capture - the Engine's warm-up steps run lowp_refresh_weights eagerly before anything is captured)."""
    ents = list(_lowp["w"].values())
    key = tuple((src.data_ptr(), y.data_ptr(), yt.data_ptr()) for src, y, yt in ents)
    tab = _lowp.get("table")
    if tab is not None and tab[0] == key:
        return tab
    assert not (ents[0][0].is_cuda and torch.cuda.is_current_stream_capturing()), "the 16-bit weight table must exist before graph capture (run a warm-up step)"
    items = (_Cast16Item * len(ents))()
    tile = 0
    for it, (src, y, yt) in zip(items, ents):
        rows, cols = src.shape
        assert src.stride(1) == 1 and y.stride(0) >= cols and yt.stride(0) % 8 == 0 and yt.stride(0) >= (rows + 7) // 8 * 8 and yt.data_ptr() % 16 == 0
        it.x, it.y16, it.y16t = src.data_ptr(), y.data_ptr(), yt.data_ptr()
        it.rows, it.cols, it.ldx, it.ldy, it.ldyt, it.tile0 = rows, cols, src.stride(0), y.stride(0), yt.stride(0), tile
        tile += ((rows + 63) // 64) * ((cols + 63) // 64)
    host = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
    dev = host.to(ents[0][0].device)
    tab = (key, dev, len(ents), tile)
    _lowp["table"] = tab
    return tab


def lowp_refresh_weights():
    ents = _lowp["w"]
    if not ents:
        return
    if not CAST16_MULTI or len(ents) == 1:
        for src, y, yt in ents.values():
            cast16(src, out=y, out_t=yt)
        return
    _, dev, n, tiles = _cast16_table()
    src0 = next(iter(ents.values()))[0]
    check(L().tf_cast16_multi_f32(ctypes.c_void_p(dev.data_ptr()), n, tiles, _lowp["dtype"], stream_of(src0)), "tf_cast16_multi_f32")


class lowp_managed:
    """train.Engine: the cached 16-bit weight copies are valid between two AdamW steps."""

    def __enter__(self):
        self.prev, _lowp["managed"] = _lowp["managed"], True

    def __exit__(self, *a):
        _lowp["managed"] = self.prev


def force_plan(bm=0, bn=0, bk=0, splitk=1):
    """Tests: pin one engine tiling for every call (bm=0 restores planning)."""
    check(L().tf_force_plan(bm, bn, bk, splitk), "tf_force_plan")


def force_dma(kind, splitk=1):
    """tests: pin LDS-DMA GEMM configuration ``kind`` (1..5); calls that are not eligible (unaligned / non-plain operands) keep the heuristic."""
    check(L().tf_force_dma(kind, splitk), "tf_force_dma")


def plans_save(path):
    check(L().tf_plans_save(path.encode()), "tf_plans_save")


def plans_load(path):
    n = L().tf_plans_load(path.encode())
    if n < 0:
        check(n, "tf_plans_load")
    return n


def _c(t):
    assert t.is_contiguous(), "expected a contiguous tensor, got strides %s for shape %s" % (t.stride(), tuple(t.shape))
    return t


def wptr(w):
    """Pointer to a conv weight stored channels_last (or any 2-D / 1x1 weight)."""
    if w.dim() == 4:
        assert w.permute(0, 2, 3, 1).is_contiguous(), "conv weights must be channels_last"
    else:
        assert w.is_contiguous()
    return ptr(w)



# ------------------------------------------------------------------------------------------ GEMM
hbm_census = None   # set to a list to record (name, algorithmic bytes, start_event, end_event) of the bandwidth-bound calls north_star names (bench.py roofline_hbm)


def _hbm_begin():
    if hbm_census is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _hbm_end(e0, name, nbytes):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        hbm_census.append((name, nbytes, e0, e1))


census = None   # set to a list to record (kind, shape, flops, start_event, end_event) of every MFMA-engine call (tools/census.py)


def _census_begin():
    if census is None or _pair_open[0]:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    return e


def _census_end(e0, kind, shape, flops):
    if e0 is not None:
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record()
        census.append((kind, shape, flops, e0, e1))


_CHECK = bool(int(__import__("os").environ.get("TF_CHECK", "0")))  # debug aid: verify every GEMM against fp64 ATen (slow)


def _gemm_reference(a, b, c_old, m, n, k, lda, ldb, ldc, a_trans, b_trans, bias, res, ldres, alpha, relu, accumulate, batch, inner, sa, sb, sc):
    outer = batch // inner
    A = torch.as_strided(a, (outer, inner, m, k), (sa[0], sa[1], 1 if a_trans else lda, lda if a_trans else 1)).double()
    B = torch.as_strided(b, (outer, inner, k, n), (sb[0], sb[1], ldb if b_trans else 1, 1 if b_trans else ldb)).double()
    R = alpha * (A @ B)
    if bias is not None:
        R = R + bias.double().view(1, 1, 1, n)
    if res is not None:
        R = R + torch.as_strided(res, (outer, inner, m, n), (sc[0], sc[1], ldres, 1)).double()
    if relu:
        R = R.clamp_min(0)
    if accumulate:
        R = R + c_old
    return R


TWO_PASS_SPLITK = os.environ.get("TF_TWO_PASS_SPLITK", "1") != "0"
_TRACE_GEMM = os.environ.get("TF_TRACE_GEMM", "0") == "1"
_TWO_PASS_MAX_ELEMS = 4300000      # <= ~256 tiles of 128 x 128


STREAM_K = os.environ.get("TF_STREAM_K", "0") != "0"      # opt-in: stream-K plans (measured no faster than the data-parallel launches, profiles/r04_sk_lab_*; no shipped plan uses one).
# Off, no call hands the library hand-over flags, so the tuner cannot pick a stream-K plan on timing noise and the two-pass scratch query keeps its output-size cap
_skf_cache = {}


def _sk_flags(device):
    """Hand-over flags of the stream-K GEMM plans: zero between launches (the kernels reset what they set), one buffer per (device, stream)."""
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if getattr(device, "type", str(device)[:4]) == "cuda" else 0)
    f = _skf_cache.get(key)
    if f is None:
        f = _skf_cache[key] = torch.zeros(2048, dtype=torch.int32, device=device)
    return f


_wsq = None


def _ws_query():
    global _wsq
    if _wsq is None:
        _wsq = L().tf_gemm_splitk_ws_floats
        _wsq.restype = ctypes.c_long
    return _wsq


def gemm(a, b, c, m, n, k, lda, ldb, ldc, a_trans=False, b_trans=False, bias=None, res=None, ldres=0, alpha=1.0, relu=False,
         accumulate=False, batch=1, inner=1, sa=(0, 0), sb=(0, 0), sc=(0, 0), mask=None, colstat=None, drop=None):
    """colstat: a ColStat request (see linear_fwd(..., colstat=True)): the epilogue also writes the BatchNorm statistics of the output.
    drop = (seed tensor, site, p): nn.Dropout on the product before ``res`` is added (tf_gemm_desc.drop_seed; the mask of ops.dropout(seed, site, p))."""
    if _CHECK and c.is_cuda:
        cview = lambda: torch.as_strided(c, (batch // inner, inner, m, n), (sc[0], sc[1], ldc, 1))
        old = cview().double().clone() if accumulate else None
        ref = _gemm_reference(a, b, old, m, n, k, lda, ldb, ldc, a_trans, b_trans, bias, res, ldres, alpha, relu, accumulate, batch, inner, sa, sb, sc)
    # outputs with too few tiles for the 256 CUs (e.g. the GPT-4 MLP's [1740 x 6048] . [6048 x 1512]: 168 tiles) may run as a deterministic
    # two-pass split-K: the scratch for <= 4 k-slices comes from the caching allocator (stream-ordered, reused inside a captured graph)
    d = GemmDesc(a=ptr(a), b=ptr(b), c=ptr(c), bias=ptr(bias), res=ptr(res), m=m, n=n, k=k, a_trans=int(a_trans), b_trans=int(b_trans),
                 lda=lda, ldb=ldb, ldc=ldc, ldres=ldres, batch=batch, inner=inner, sa_outer=sa[0], sa_inner=sa[1], sb_outer=sb[0],
                 sb_inner=sb[1], sc_outer=sc[0], sc_inner=sc[1], alpha=alpha, relu=int(relu), accumulate=int(accumulate), mask=ptr(mask),
                 ldmask=mask.stride(0) if mask is not None else 0, splitk_ws=c_p(0), splitk_ws_floats=0)
    if colstat is not None:
        d.colstat, d.colstat_nparts = ptr(colstat.buf), ctypes.pointer(colstat.nparts)
    if drop is not None:
        assert ldc == n and batch == 1 and not accumulate, "gemm: dropout epilogue needs a contiguous (m, n) plain store"
        d.drop_seed, d.drop_site, d.drop_p = ptr(drop[0]), int(drop[1]), float(drop[2])
    skws = None
    if TWO_PASS_SPLITK and batch == 1 and k >= 256 and 128 * 128 <= m * n and (m * n <= _TWO_PASS_MAX_ELEMS or STREAM_K):
        need = _ws_query()(byref(d))          # > 0 only when the plan of THIS shape is a two-pass / stream-K plan (or the autotuner wants to try one)
        if need > 0:
            skws = torch.empty(need, dtype=torch.float32, device=c.device)
            d.splitk_ws, d.splitk_ws_floats = ptr(skws), need
            if STREAM_K:
                d.sk_flags = ptr(_sk_flags(c.device))
    _e = _census_begin()
    if _pair_open[0] and census is not None:
        _pair_shapes.append((m, n, k, batch, 2.0 * m * n * k * batch))
    if _TRACE_GEMM:      # debugging aid: name every call before it runs and wait for it (TF_TRACE_GEMM=1)
        print("[gemm] m=%d n=%d k=%d a_trans=%d b_trans=%d lda=%d ldb=%d ldc=%d batch=%d acc=%d relu=%d bias=%d res=%d mask=%d ws=%s" % (
            m, n, k, a_trans, b_trans, lda, ldb, ldc, batch, accumulate, relu, bias is not None, res is not None, mask is not None,
            skws.numel() if skws is not None else None), file=sys.stderr, flush=True)
    check(L().tf_gemm_f32(byref(d), stream_of(c)), "tf_gemm_f32")
    if _TRACE_GEMM and c.is_cuda:
        torch.cuda.synchronize()
    _census_end(_e, "gemm a%db%d" % (a_trans, b_trans), (m, n, k, batch), 2.0 * m * n * k * batch)
    if _CHECK and c.is_cuda:
        got = cview().double()
        err = (got - ref).abs().max().item()
        rms = ref.pow(2).mean().sqrt().item() + 1e-30
        if not err <= 1e-4 * rms:
            bad = ((got - ref).abs() > 1e-4 * rms)
            idx = bad.nonzero()
            raise RuntimeError("TF_CHECK gemm mismatch: m=%d n=%d k=%d at=%d bt=%d lda=%d ldb=%d ldc=%d batch=%d inner=%d alpha=%g relu=%d acc=%d bias=%s res=%s "
                               "maxerr=%.3e rms=%.3e nbad=%d first_bad=%s rows=%s cols=%s" %
                               (m, n, k, a_trans, b_trans, lda, ldb, ldc, batch, inner, alpha, relu, accumulate, bias is not None, res is not None, err, rms,
                                int(bad.sum()), idx[:6].tolist(), sorted(set(idx[:, 2].tolist()))[:24], sorted(set(idx[:, 3].tolist()))[:24]))
    return c


# ---- BatchNorm statistics fused into the PRODUCING convolution (verdict r2 item 1).  The layers below a few ten-thousand rows (stages 2-4 of
# both trunks: 120 of the 136 BatchNorm layers) are latency-, not bandwidth-bound: their separate moments pass is 6-12 us of launch + a
# re-read of the conv output.  The epilogue of the producing GEMM / grouped convolution writes per-part Welford triples instead; larger maps
# keep the streaming reduction (thousands of parts per channel would make the merge the slow part).
FUSE_BN_STATS = os.environ.get("TF_FUSE_BN_STATS", "1") != "0"
_COLSTAT_MAX_ROWS = 40000


class ColStat:
    """Per-part Welford triples [part][{count, mean, M2}][C] written by a producer's epilogue; ``nparts`` is filled in (on the host, at
    launch time) by the C library: 0 = this launch could not produce them."""

    def __init__(self, rows, C, device, max_parts=None):
        self.nparts = ctypes.c_int(0)
        self.rows, self.C = rows, C
        self.buf = torch.empty(3 * C * (max_parts or (rows + 31) // 32), dtype=torch.float32, device=device)

    def __bool__(self):
        return self.nparts.value > 0


def want_colstat(rows):
    return FUSE_BN_STATS and rows <= _COLSTAT_MAX_ROWS


def linear_fwd(x, w, bias=None, relu=False, res=None, out=None, colstat=False, drop=None):
    """y = x @ w.T + bias (+res) (relu); x (M, K) row-major (row stride may exceed K), w (N, K).
    colstat=True: returns (y, ColStat) - the epilogue also gathers the BatchNorm statistics of y (None when not worthwhile / not possible).
    drop = (seed, site, p): y = res + dropout(x @ w.T + bias) in the same launch (the Block's residual branches, transfuser.py:543-549)."""
    M, K = x.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
    cs = ColStat(M, N, x.device) if (colstat and want_colstat(M) and res is None and not relu) else None
    y = gemm(x, w, out, M, N, K, x.stride(0), w.stride(0), out.stride(0), bias=bias, res=res,
             ldres=res.stride(0) if res is not None else 0, relu=relu, colstat=cs, drop=drop)
    return (y, cs if cs else None) if colstat else y


def linear_dgrad(dy, w, out=None, accumulate=False, res=None, mask=None):
    """dx = dy @ w (+res); dy (M, N), w (N, K).  mask (M, K): dx is zeroed where mask <= 0 (fused ReLU backward)."""
    M, N = dy.shape
    K = w.shape[1]
    if out is None:
        out = torch.empty(M, K, dtype=torch.float32, device=dy.device)
    return gemm(dy, w, out, M, K, N, dy.stride(0), w.stride(0), out.stride(0), b_trans=True, accumulate=accumulate, res=res,
                ldres=res.stride(0) if res is not None else 0, mask=mask)


def linear_wgrad(dy, x, dw, accumulate=True):
    """dw (+)= dy.T @ x; dy (M, N), x (M, K), dw (N, K)."""
    M, N = dy.shape
    K = x.shape[1]
    return gemm(dy, x, dw, N, K, M, dy.stride(0), x.stride(0), dw.stride(0), a_trans=True, b_trans=True, accumulate=accumulate)


# ------------------------------------------------------------------------------------------ conv
def conv_geom(x_shape, cout, ksize, stride, pad, groups):
    B, Hi, Wi, Cin = x_shape
    Ho = (Hi + 2 * pad - ksize) // stride + 1
    Wo = (Wi + 2 * pad - ksize) // stride + 1
    return ConvGeom(B, Hi, Wi, Cin, Ho, Wo, cout, ksize, stride, pad, groups)


def _gshape(g):
    return (g.B, g.Hi, g.Wi, g.Cin, g.Cout, g.ksize, g.stride, g.groups)


def _gflops(g):
    return 2.0 * g.B * g.Ho * g.Wo * g.Cout * (g.Cin // g.groups) * g.ksize * g.ksize


_DIRECT_MIN_PIXELS = 1 << 16   # below this the engine's im2col path is not L2-bound and wins


def _direct_ok(shape, cout, cin, ks, stride, pad, groups):
    """Large-resolution 3x3/s1/p1 conv with <= 32 channels on both sides -> the LDS-tiled direct kernels (csrc/conv_direct.cpp)."""
    B, H, W, _ = shape
    return _DIRECT and ks == 3 and stride == 1 and pad == 1 and groups == 1 and cin <= 32 and cout <= 32 and B * H * W >= _DIRECT_MIN_PIXELS


_DIRECT = bool(int(__import__("os").environ.get("TF_DIRECT_CONV", "1")))
_THIN = bool(int(__import__("os").environ.get("TF_THIN_CONV", "1")))
_S2_GEMM = bool(int(__import__("os").environ.get("TF_S2_GEMM", "1")))
_S2_SUBPIX = bool(int(__import__("os").environ.get("TF_S2_SUBPIX", "1")))


def _thin_ok(shape, cout, cin, ks, stride, pad, groups):
    """The decoders' last layer (32 -> 7 / 32 -> 1 at full resolution): taps folded into the GEMM dimensions (csrc/conv_thin.cpp)."""
    return _THIN and _direct_ok(shape, cout, cin, ks, stride, pad, groups) and cin == 32 and cout <= 7
_GROUPED = bool(int(__import__("os").environ.get("TF_GROUPED_CONV", "1")))
_gws_cache = {}


def _grouped_ok(shape, cout, cin, ks, stride, pad, groups):
    """RegNetY bottleneck convolution: 3x3 / s1 / p1, group width 24 on both sides -> the per-group direct kernels (csrc/conv_grouped.cpp).
    The stride-2 first block of every stage stays on the implicit-GEMM engine."""
    return _GROUPED and ks == 3 and stride == 1 and pad == 1 and groups > 1 and cin == cout == groups * 24 and shape[1] >= 4 and shape[2] >= 8


_GROUPED_S2 = bool(int(__import__("os").environ.get("TF_GROUPED_S2", "1")))      # round 5: default on (GPU suite green with it; same-lease A/B 49.42-49.62 vs 49.62-49.74 ms/step together with the im2col form, gpurun_out/r05_call1.log)


def _grouped_s2_ok(shape, cout, cin, ks, stride, pad, groups):
    """The stride-2 grouped 3x3 convolution of the first block of a RegNetY stage -> csrc/conv_grouped.cpp:conv3x3_grouped_s2_{fwd,wgrad}_kernel."""
    return (_GROUPED_S2 and ks == 3 and stride == 2 and pad == 1 and groups > 1 and cin == cout == groups * 24 and (shape[1] - 1) // 2 + 1 >= 4 and
            (shape[2] - 1) // 2 + 1 >= 8)


_IM2COL_GEMM = bool(int(__import__("os").environ.get("TF_IM2COL_GEMM", "1")))    # round 5: default on (parity at the decoders' own shapes on the MI355X: test_im2col_gemm_form_of_few_row_deep_k_convolutions)


def _im2col_gemm_ok(g, ks, stride, pad, groups):
    """Dense 3x3 / s1 / p1 convolution with FEW output rows and a DEEP contraction (the first layer of the Seg / Depth decoders, 512 -> 128 at 8 x 22:
    56 output tiles, K = 4608): materialise the im2col matrix (tf_im2col3x3_f32) and run a plain GEMM, whose split-K plans fill the chip - through
    the implicit GEMM each tile is one 144-step k-chain (178 us at 11.6 TFLOP/s, profiles/r04_census_fp32_final.txt)."""
    return (_IM2COL_GEMM and ks == 3 and stride == 1 and pad == 1 and groups == 1 and g.Cin % 4 == 0 and 9 * g.Cin >= 1024 and g.B * g.Ho * g.Wo <= 4096 and
            g.Cout <= 256)


def _grouped_ws(device):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if getattr(device, "type", str(device)[:4]) == "cuda" else 0)
    ws = _gws_cache.get(key)
    if ws is None:
        L().tf_conv3x3_grouped_wgrad_ws_floats.restype = ctypes.c_long
        ws = torch.empty(L().tf_conv3x3_grouped_wgrad_ws_floats(), dtype=torch.float32, device=device)
        _gws_cache[key] = ws
    return ws
_DIRECT_WGRAD_MAX_COUT = 32  # measured (tools/conv_bench.py): 421-425 us for every Cout at 256x704 vs 674-752 us through the engine's split-K path


def conv_fwd(x, w, bias=None, stride=1, pad=None, groups=1, relu=False, colstat=False):
    """colstat=True (bias-free, no ReLU: a conv followed by BatchNorm): returns (y, ColStat or None), see linear_fwd."""
    ks = w.shape[2]
    pad = ks // 2 if pad is None else pad
    g = conv_geom(x.shape, w.shape[0], ks, stride, pad, groups)
    y = torch.empty(g.B, g.Ho, g.Wo, g.Cout, dtype=torch.float32, device=x.device)
    _e = _census_begin()
    if colstat:
        rows = g.B * g.Ho * g.Wo
        cs = None
        if want_colstat(rows) and bias is None and not relu and not _direct_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
            if _grouped_s2_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
                cs = ColStat(rows, g.Cout, x.device, max_parts=L().tf_conv3x3_grouped_colstat_parts())
                check(L().tf_conv3x3_grouped_s2_fwd_f32(ptr(_c(x)), c_p(0), wptr(w), ptr(y), g.B, g.Hi, g.Wi, g.Cin, ptr(cs.buf), byref(cs.nparts), stream_of(x)),
                      "tf_conv3x3_grouped_s2_fwd_f32")
                _census_end(_e, "conv fwd g2", _gshape(g), _gflops(g))
            elif _grouped_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
                cs = ColStat(rows, g.Cout, x.device, max_parts=L().tf_conv3x3_grouped_colstat_parts())
                check(L().tf_conv3x3_grouped_fwd_colstat_f32(ptr(_c(x)), wptr(w), ptr(y), g.B, g.Hi, g.Wi, g.Cin, ptr(cs.buf), byref(cs.nparts), stream_of(x)),
                      "tf_conv3x3_grouped_fwd_colstat_f32")
                _census_end(_e, "conv fwd g", _gshape(g), _gflops(g))
            else:
                cs = ColStat(rows, g.Cout, x.device)
                check(L().tf_conv2d_fwd_colstat_f32(byref(g), ptr(_c(x)), wptr(w), c_p(0), ptr(y), ptr(cs.buf), byref(cs.nparts), stream_of(x)),
                      "tf_conv2d_fwd_colstat_f32")
                _census_end(_e, "conv fwd", _gshape(g), _gflops(g))
            return y, (cs if cs else None)
        return conv_fwd(x, w, bias, stride, pad, groups, relu), None
    if not relu and _thin_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_thin_fwd_f32(ptr(_c(x)), wptr(w), ptr(bias), ptr(y), g.B, g.Hi, g.Wi, g.Cin, g.Cout, stream_of(x)), "tf_conv3x3_thin_fwd_f32")
        _census_end(_e, "conv fwd t", _gshape(g), _gflops(g))
        return y
    if _direct_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_small_fwd_f32(ptr(_c(x)), wptr(w), ptr(bias), ptr(y), g.B, g.Hi, g.Wi, g.Cin, g.Cout, int(relu), stream_of(x)),
              "tf_conv3x3_small_fwd_f32")
        _census_end(_e, "conv fwd*", _gshape(g), _gflops(g))
        return y
    if _grouped_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_grouped_fwd_f32(ptr(_c(x)), wptr(w), ptr(bias), ptr(y), g.B, g.Hi, g.Wi, g.Cin, int(relu), stream_of(x)), "tf_conv3x3_grouped_fwd_f32")
        _census_end(_e, "conv fwd g", _gshape(g), _gflops(g))
        return y
    if _im2col_gemm_ok(g, ks, stride, pad, groups):
        M = g.B * g.Ho * g.Wo
        cols = torch.empty(M, 9 * g.Cin, dtype=torch.float32, device=x.device)
        check(L().tf_im2col3x3_f32(ptr(_c(x)), ptr(cols), g.B, g.Hi, g.Wi, g.Cin, stream_of(x)), "tf_im2col3x3_f32")
        assert w.permute(0, 2, 3, 1).is_contiguous(), "conv weights must be channels_last"
        linear_fwd(cols, w.permute(0, 2, 3, 1).reshape(g.Cout, 9 * g.Cin), bias, relu=relu, out=y.view(M, g.Cout))      # its split-K plans fill the chip
        return y
    if bias is None and not relu and _grouped_s2_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_grouped_s2_fwd_f32(ptr(_c(x)), c_p(0), wptr(w), ptr(y), g.B, g.Hi, g.Wi, g.Cin, c_p(0), c_p(0), stream_of(x)), "tf_conv3x3_grouped_s2_fwd_f32")
        _census_end(_e, "conv fwd g2", _gshape(g), _gflops(g))
        return y
    check(L().tf_conv2d_fwd_f32(byref(g), ptr(_c(x)), wptr(w), ptr(bias), ptr(y), int(relu), stream_of(x)), "tf_conv2d_fwd_f32")
    _census_end(_e, "conv fwd", _gshape(g), _gflops(g))
    return y


def conv_dgrad(dy, w, x_shape, stride=1, pad=None, groups=1, out=None, accumulate=False, mask=None):
    """mask (optional, shape of dx): the forward output of the ReLU layer that produced this convolution's input - dx is zeroed where it is
    <= 0 (that layer's ReLU backward; fused into the epilogue of the thin-output kernels, one extra launch otherwise)."""
    ks = w.shape[2]
    pad = ks // 2 if pad is None else pad
    g = conv_geom(x_shape, w.shape[0], ks, stride, pad, groups)
    if out is None:
        out = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    _e = _census_begin()
    if _thin_ok(x_shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_thin_dgrad_f32(ptr(_c(dy)), wptr(w), ptr(_c(mask)) if mask is not None else c_p(0), ptr(_c(out)), g.B, g.Hi, g.Wi, g.Cin, g.Cout,
                                            int(accumulate), stream_of(dy)), "tf_conv3x3_thin_dgrad_f32")
        _census_end(_e, "conv dgrad t", _gshape(g), _gflops(g))
        return out
    if mask is not None:
        assert not accumulate, "conv_dgrad: mask + accumulate needs the fused kernel"
        return relu_mask(conv_dgrad(dy, w, x_shape, stride, pad, groups, out=out), mask, out=out)
    if _S2_GEMM and ks == 1 and stride == 2 and pad == 0 and groups == 1 and accumulate and g.Cin % 4 == 0:
        # RegNet downsample branch: dx[b, 2i, 2j, :] += dy[b, i, j, :] W - one plain GEMM over the B Ho Wo output pixels + a scatter-add (the
        # implicit-GEMM path walks all B Hi Wi input pixels, 3/4 of them for nothing, with gathered operands: 190 vs ~40 us at 16 x 44 x 576 -> 1512)
        tmp = linear_dgrad(dy.view(-1, g.Cout), w.view(g.Cout, g.Cin))
        check(L().tf_add_strided2_f32(ptr(tmp), ptr(_c(out)), g.B, g.Ho, g.Wo, g.Cin, g.Hi, g.Wi, stream_of(dy)), "tf_add_strided2_f32")
        return out
    if _direct_ok(x_shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_small_dgrad_f32(ptr(_c(dy)), wptr(w), ptr(_c(out)), g.B, g.Hi, g.Wi, g.Cin, g.Cout, int(accumulate), stream_of(dy)),
              "tf_conv3x3_small_dgrad_f32")
        _census_end(_e, "conv dgrad*", _gshape(g), _gflops(g))
        return out
    if _GROUPED and _S2_SUBPIX and ks == 3 and stride == 2 and pad == 1 and groups > 1 and g.Cin == g.Cout == groups * 24:
        check(L().tf_conv3x3_grouped_s2_dgrad_f32(ptr(_c(dy)), wptr(w), ptr(_c(out)), g.B, g.Hi, g.Wi, g.Cin, int(accumulate), stream_of(dy)),
              "tf_conv3x3_grouped_s2_dgrad_f32")
        _census_end(_e, "conv dgrad g2", _gshape(g), _gflops(g))
        return out
    if _grouped_ok(x_shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_grouped_dgrad_f32(ptr(_c(dy)), wptr(w), ptr(_c(out)), g.B, g.Hi, g.Wi, g.Cin, int(accumulate), stream_of(dy)), "tf_conv3x3_grouped_dgrad_f32")
        _census_end(_e, "conv dgrad g", _gshape(g), _gflops(g))
        return out
    check(L().tf_conv2d_dgrad_f32(byref(g), ptr(_c(dy)), wptr(w), ptr(_c(out)), int(accumulate), stream_of(dy)), "tf_conv2d_dgrad_f32")
    _census_end(_e, "conv dgrad", _gshape(g), _gflops(g))
    return out


_thin_ws_cache = {}


def _thin_ws(device):
    key = (str(device), torch.cuda.current_stream(device).cuda_stream if getattr(device, "type", str(device)[:4]) == "cuda" else 0)
    ws = _thin_ws_cache.get(key)
    if ws is None:
        L().tf_conv3x3_thin_wgrad_ws_floats.restype = ctypes.c_long
        ws = _thin_ws_cache[key] = torch.empty(L().tf_conv3x3_thin_wgrad_ws_floats(), dtype=torch.float32, device=device)
    return ws


def conv_wgrad(dy, x, dw, stride=1, pad=None, groups=1, accumulate=True, dbias=None):
    """dbias (optional, (Cout,) accumulator): the bias gradient sum(dy) is added to it by the same launch where the kernel can (thin-output
    layers); returns dw - callers test ``conv_wgrad_takes_bias`` to know whether dbias was consumed."""
    return _conv_wgrad(dy, x, dw, stride, pad, groups, accumulate, dbias)


def _conv_wgrad(dy, x, dw, stride, pad, groups, accumulate, dbias):
    ks = dw.shape[2]
    pad = ks // 2 if pad is None else pad
    g = conv_geom(x.shape, dw.shape[0], ks, stride, pad, groups)
    _e = _census_begin()
    if _thin_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_thin_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), wptr(dw), ptr(dbias), g.B, g.Hi, g.Wi, g.Cin, g.Cout, int(accumulate), ptr(_thin_ws(x.device)),
                                            stream_of(dy)), "tf_conv3x3_thin_wgrad_f32")
        _census_end(_e, "conv wgrad t", _gshape(g), _gflops(g))
        return dw
    assert dbias is None, "conv_wgrad: dbias is only fused by the thin-output kernels (check conv_wgrad_takes_bias first)"
    if _direct_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups) and (g.Cout <= _DIRECT_WGRAD_MAX_COUT or _DIRECT_MIN_PIXELS == 0):
        check(L().tf_conv3x3_small_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), wptr(dw), g.B, g.Hi, g.Wi, g.Cin, g.Cout, int(accumulate), ptr(workspace(x.device)),
                                             stream_of(dy)), "tf_conv3x3_small_wgrad_f32")
        _census_end(_e, "conv wgrad*", _gshape(g), _gflops(g))
        return dw
    if _grouped_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_grouped_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), wptr(dw), g.B, g.Hi, g.Wi, g.Cin, int(accumulate), ptr(_grouped_ws(x.device)),
                                               stream_of(dy)), "tf_conv3x3_grouped_wgrad_f32")
        _census_end(_e, "conv wgrad g", _gshape(g), _gflops(g))
        return dw
    if _grouped_s2_ok(x.shape, g.Cout, g.Cin, ks, stride, pad, groups):
        check(L().tf_conv3x3_grouped_s2_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), c_p(0), wptr(dw), g.B, g.Hi, g.Wi, g.Cin, int(accumulate), ptr(_grouped_ws(x.device)),
                                                  stream_of(dy)), "tf_conv3x3_grouped_s2_wgrad_f32")
        _census_end(_e, "conv wgrad g2", _gshape(g), _gflops(g))
        return dw
    check(L().tf_conv2d_wgrad_f32(byref(g), ptr(_c(dy)), ptr(_c(x)), wptr(dw), int(accumulate), stream_of(dy)), "tf_conv2d_wgrad_f32")
    _census_end(_e, "conv wgrad", _gshape(g), _gflops(g))
    return dw


def conv_wgrad_takes_bias(x_shape, cout, ks, stride=1, pad=None, groups=1):
    return _thin_ok(x_shape, cout, x_shape[3], ks, stride, ks // 2 if pad is None else pad, groups)


def stem_conv_fwd(s0, s1, w, normalize, stride=2, pad=None):
    """k x k conv (3x3 / s2 for RegNet, 7x7 / s2 / p3 for ResNet) on NCHW inputs (s1 optional extra channels) -> NHWC."""
    B, C0, H, W = s0.shape
    C1 = s1.shape[1] if s1 is not None else 0
    ks = w.shape[2]
    g = conv_geom((B, H, W, C0 + C1), w.shape[0], ks, stride, ks // 2 if pad is None else pad, 1)
    y = torch.empty(B, g.Ho, g.Wo, g.Cout, dtype=torch.float32, device=s0.device)
    check(L().tf_stem_conv_fwd_f32(byref(g), ptr(_c(s0)), C0, ptr(_c(s1)) if s1 is not None else c_p(0), C1, int(normalize), wptr(w), ptr(y),
                                   stream_of(s0)), "tf_stem_conv_fwd_f32")
    return y


def stem_conv_wgrad(dy, s0, s1, dw, normalize, accumulate=True, stride=2, pad=None):
    B, C0, H, W = s0.shape
    C1 = s1.shape[1] if s1 is not None else 0
    ks = dw.shape[2]
    g = conv_geom((B, H, W, C0 + C1), dw.shape[0], ks, stride, ks // 2 if pad is None else pad, 1)
    L().tf_stem_conv_wgrad_ws_floats.restype = ctypes.c_long
    need = L().tf_stem_conv_wgrad_ws_floats(byref(g), C0, C1)     # > 0: the direct kernels want scratch for their partial panels (RegNet stems)
    ws = torch.empty(need, dtype=torch.float32, device=dy.device) if need > 0 else None
    check(L().tf_stem_conv_wgrad_ws_f32(byref(g), ptr(_c(dy)), ptr(_c(s0)), C0, ptr(_c(s1)) if s1 is not None else c_p(0), C1, int(normalize),
                                        wptr(dw), int(accumulate), ptr(ws), ctypes.c_long(need), stream_of(dy)), "tf_stem_conv_wgrad_ws_f32")
    return dw


def maxpool3x3s2_fwd(x):
    """nn.MaxPool2d(3, 2, 1) on NHWC (timm ResNet stem) -> (y, idx uint8 winning taps)."""
    B, H, W, C = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
    idx = torch.empty(B, Ho, Wo, C, dtype=torch.uint8, device=x.device)
    check(L().tf_maxpool3x3s2_fwd_f32(ptr(_c(x)), ptr(y), ctypes.c_void_p(idx.data_ptr()), B, H, W, C, stream_of(x)), "tf_maxpool3x3s2_fwd_f32")
    return y, idx


def maxpool3x3s2_bwd(dy, idx, x_shape):
    B, H, W, C = x_shape
    dx = torch.empty(tuple(x_shape), dtype=torch.float32, device=dy.device)
    check(L().tf_maxpool3x3s2_bwd_f32(ptr(_c(dy)), ctypes.c_void_p(idx.data_ptr()), ptr(dx), B, H, W, C, stream_of(dy)), "tf_maxpool3x3s2_bwd_f32")
    return dx


# ------------------------------------------------------------------------------------------ ConvNeXt pieces (csrc/convnext.cpp)
def dwconv7(x, w, bias=None, flip=False, out=None, accumulate=False):
    """Depthwise 7x7 / pad 3 on NHWC; w = the (C, 1, 7, 7) parameter; flip=True: mirrored taps (= the input gradient of the forward)."""
    B, H, W, C = x.shape
    assert tuple(w.shape[-2:]) == (7, 7) and w.shape[0] == C and w.numel() == C * 49 and w.is_contiguous()
    if out is None:
        out = torch.empty_like(x)
    check(L().tf_dwconv7_fwd_f32(ptr(_c(x)), ptr(w), ptr(bias), ptr(out), B, H, W, C, int(flip), int(accumulate), stream_of(x)), "tf_dwconv7_fwd_f32")
    return out


def dwconv7_wgrad(dy, x, dw, dbias=None):
    """dw (C, 1, 7, 7) += ..., dbias (C,) += ... (accumulating)."""
    B, H, W, C = x.shape
    assert dw.is_contiguous()
    check(L().tf_dwconv7_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), ptr(dw), ptr(dbias), B, H, W, C, stream_of(x)), "tf_dwconv7_wgrad_f32")


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(L().tf_gelu_fwd_f32(ptr(_c(x)), ptr(y), ctypes.c_int64(x.numel()), stream_of(x)), "tf_gelu_fwd_f32")
    return y


def gelu_bwd(dy, x, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(L().tf_gelu_bwd_f32(ptr(_c(dy)), ptr(_c(x)), ptr(out), ctypes.c_int64(x.numel()), stream_of(x)), "tf_gelu_bwd_f32")
    return out


def colscale_add(x, gamma=None, beta=None, res=None, out=None):
    """res + gamma[c] * x + beta[c] over the last dimension (each optional)."""
    C = x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    check(L().tf_colscale_add_f32(ptr(_c(x)), ptr(gamma), ptr(beta), ptr(_c(res)) if res is not None else c_p(0), ptr(out), ctypes.c_int64(x.numel() // C), C,
                                  stream_of(x)), "tf_colscale_add_f32")
    return out


def colsum_mul(a, b, out, accumulate=True):
    """out (C,) (+)= sum over rows of a * b."""
    C = a.shape[-1]
    check(L().tf_colsum_mul_f32(ptr(_c(a)), ptr(_c(b)), a.numel() // C, C, ptr(out), int(accumulate), ptr(workspace(a.device)), stream_of(a)), "tf_colsum_mul_f32")
    return out


# ------------------------------------------------------------------------------------------ norms
def layernorm_fwd(x, gamma, beta, eps=1e-5):
    rows, C = x.shape
    y = torch.empty_like(x)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    _h = _hbm_begin()
    check(L().tf_layernorm_fwd_f32(ptr(_c(x)), ptr(gamma), ptr(beta), ptr(y), ptr(mean), ptr(rstd), rows, C, ctypes.c_float(eps), stream_of(x)),
          "tf_layernorm_fwd_f32")
    _hbm_end(_h, "layernorm forward (row in registers: x read, y written)", 8 * x.numel())
    return y, mean, rstd


def layernorm_fwd16(x, gamma, beta, eps=1e-5):
    """LayerNorm whose outputs are the 16-bit operand copies only (lowp_storage() modes): (y16 (rows, C), y16t (C, rows8), mean, rstd) ==
    cast16(layernorm_fwd(x)[0]) bitwise, without the fp32 tensor in between (tf_layernorm_fwd16_f32)."""
    rows, C = x.shape
    rows8 = (rows + 7) // 8 * 8
    y = torch.empty(rows, C, dtype=_t16(), device=x.device)
    yt = torch.empty(C, rows8, dtype=_t16(), device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty_like(mean)
    check(L().tf_layernorm_fwd16_f32(ptr(_c(x)), ptr(gamma), ptr(beta), ctypes.c_void_p(y.data_ptr()), C, ctypes.c_void_p(yt.data_ptr()), rows8, ptr(mean), ptr(rstd),
                                     rows, C, ctypes.c_float(eps), _lowp["dtype"], stream_of(x)), "tf_layernorm_fwd16_f32")
    return y, yt, mean, rstd


def layernorm_fwd16_ok(x, gamma, beta):
    C = x.shape[1]
    return C % 4 == 0 and C <= 2048 and x.is_contiguous() and all(t.data_ptr() % 16 == 0 for t in (x, gamma, beta))


def layernorm_bwd(dy, x, gamma, mean, rstd, dgamma=None, dbeta=None, dx=None, accumulate=False, drop=None):
    """drop = (seed, site, p): also returns nn.Dropout(p)(dx) of the finished gradient (mask of ops.dropout(seed, site)) -> (dx, dropped)."""
    rows, C = x.shape
    if dx is None:
        dx = torch.empty_like(x)
    if drop is not None:
        dropped = torch.empty_like(dx)
        assert dx.is_contiguous()
        check(L().tf_layernorm_bwd_drop_f32(ptr(_c(dy)), ptr(_c(x)), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), int(accumulate), ptr(dgamma), ptr(dbeta),
                                            rows, C, ptr(dropped), ptr(drop[0]), ctypes.c_uint32(int(drop[1])), ctypes.c_float(drop[2]), stream_of(x)),
              "tf_layernorm_bwd_drop_f32")
        return dx, dropped
    check(L().tf_layernorm_bwd_f32(ptr(_c(dy)), ptr(_c(x)), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), int(accumulate), ptr(dgamma), ptr(dbeta),
                                   rows, C, stream_of(x)), "tf_layernorm_bwd_f32")
    return dx


def softmax_fwd_(s, rows, n, ld):
    check(L().tf_softmax_fwd_f32(ptr(s), rows, n, ld, stream_of(s)), "tf_softmax_fwd_f32")
    return s


def softmax_dropout_fwd_(s, rows, n, ld, seed, site, p):
    """softmax in place + attn_drop into a new tensor, one launch; returns the dropped probabilities."""
    sd = torch.empty_like(s)
    check(L().tf_softmax_dropout_fwd_f32(ptr(s), ptr(sd), rows, n, ld, ptr(seed), ctypes.c_uint32(site), ctypes.c_float(p), stream_of(s)),
          "tf_softmax_dropout_fwd_f32")
    return sd


def softmax_dropout_bwd_(p, dp, rows, n, ld, seed, site, pdrop):
    check(L().tf_softmax_dropout_bwd_f32(ptr(p), ptr(dp), rows, n, ld, ptr(seed), ctypes.c_uint32(site), ctypes.c_float(pdrop), stream_of(p)),
          "tf_softmax_dropout_bwd_f32")
    return dp


def softmax_bwd_(p, dp, rows, n, ld):
    check(L().tf_softmax_bwd_f32(ptr(p), ptr(dp), rows, n, ld, stream_of(p)), "tf_softmax_bwd_f32")
    return dp


# ---- fused attention (csrc/attention.cpp)
FUSED_ATTENTION = os.environ.get("TF_FUSED_ATTENTION", "1") != "0"


def attention_supported(T, C, nh):
    return FUSED_ATTENTION and bool(L().tf_attention_supported(T, C, nh))


def attention_fwd(qkv, B, T, C, nh, drop=None):
    """qkv (B*T, 3C) = [key | query | value] -> (y (B*T, C), lse (B*nh*T)); drop = (seed, site, p) or None."""
    y = torch.empty(B * T, C, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B * nh * T, dtype=torch.float32, device=qkv.device)
    seed, site, p = drop if drop is not None else (None, 0, 0.0)
    check(L().tf_attention_fwd_f32(ptr(_c(qkv)), ptr(y), ptr(lse), B, T, C, nh, ptr(seed), ctypes.c_uint32(site), ctypes.c_float(p), stream_of(qkv)),
          "tf_attention_fwd_f32")
    return y, lse


ATT_BWD_ONE = os.environ.get("TF_ATT_BWD_ONE", "1") != "0"      # 0: the two dependent launches (queries, then keys) of rounds 3-5


def attention_bwd(qkv, dy, lse, B, T, C, nh, drop=None, y=None):
    """-> dqkv (B*T, 3C): gradients of [key | query | value]; the probabilities are recomputed from qkv and lse.  y = attention_fwd's output: the row sums
    D_i = dY_i . Y_i come from it and the key / query halves run as ONE grid (tf_attention_bwd_y_f32)."""
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    seed, site, p = drop if drop is not None else (None, 0, 0.0)
    if y is not None and ATT_BWD_ONE:
        check(L().tf_attention_bwd_y_f32(ptr(_c(qkv)), ptr(_c(dy)), ptr(_c(y)), ptr(lse), ptr(dqkv), ptr(dsum), B, T, C, nh, ptr(seed), ctypes.c_uint32(site),
                                         ctypes.c_float(p), stream_of(qkv)), "tf_attention_bwd_y_f32")
        return dqkv
    check(L().tf_attention_bwd_f32(ptr(_c(qkv)), ptr(_c(dy)), ptr(lse), ptr(dqkv), ptr(dsum), B, T, C, nh, ptr(seed), ctypes.c_uint32(site),
                                   ctypes.c_float(p), stream_of(qkv)), "tf_attention_bwd_f32")
    return dqkv


# ---- zero arena: a device buffer cleared ONCE per training step (zero_scratch_reset, called by LidarCenterNet.forward) from which
# kernels that accumulate with atomics (BatchNorm statistics, SE squeeze) take pre-zeroed slices - instead of one memset / finalize
# launch each.  Slices stay valid until the next reset (the next forward); when the pool runs dry a fresh torch.zeros is returned.
_zpool = {}
# Measured on the MI355X: accumulating the BatchNorm statistics with atomics (tf_bn_*_f32 zacc != NULL; 2 launches per BN instead of 3)
# is SLOWER than partials + a finalize kernel (64.6 vs 60.0 ms/step): same-address atomics serialise.  Kept selectable for experiments.
_DBG_LEGACY_BN = not bool(int(__import__('os').environ.get('TF_ATOMIC_BN', '0')))
_DBG_LEGACY_BNB = _DBG_LEGACY_BN
_ZPOOL_FLOATS = 8 << 20


def uses_zero_arena():
    return not _DBG_LEGACY_BN


def zero_scratch_reset(device):
    key = str(device)
    st = _zpool.get(key)
    if st is None:
        st = _zpool[key] = [torch.zeros(_ZPOOL_FLOATS, dtype=torch.float32, device=device), 0]
    else:
        st[0].zero_()
        st[1] = 0


def zero_scratch(n, device):
    st = _zpool.get(str(device))
    n4 = (n + 3) // 4 * 4
    if st is None or st[1] + n4 > _ZPOOL_FLOATS:
        return torch.zeros(n, dtype=torch.float32, device=device)
    out = st[0][st[1]:st[1] + n]
    st[1] += n4
    return out


def bn_fwd(x, gamma, beta, rmean, rvar, res=None, relu=False, training=True, momentum=0.1, eps=1e-5):
    """x: (..., C) NHWC; returns (y, save_mean, save_invstd)."""
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty_like(sm)
    zacc = zero_scratch(32 * C, x.device) if training and not _DBG_LEGACY_BN else None   # tf_bn_zacc_floats(C)
    check(L().tf_bn_fwd_f32(ptr(_c(x)), rows, C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ctypes.c_float(momentum), ctypes.c_float(eps),
                            ptr(res), int(relu), ptr(y), ptr(sm), ptr(si), ptr(workspace(x.device)), int(training), ptr(zacc), stream_of(x)),
          "tf_bn_fwd_f32")
    return y, sm, si


def bn_fwd_parts(x, cs, gamma, beta, rmean, rvar, res=None, relu=False, momentum=0.1, eps=1e-5):
    """Train-mode BatchNorm forward from the statistics the producer of x gathered (ColStat): finalize + apply, no moments pass over x."""
    C = x.shape[-1]
    rows = x.numel() // C
    assert cs and cs.C == C and cs.rows == rows
    y = torch.empty_like(x)
    sm = torch.empty(C, dtype=torch.float32, device=x.device)
    si = torch.empty_like(sm)
    _h = _hbm_begin()
    check(L().tf_bn_fwd_parts_f32(ptr(_c(x)), rows, C, ptr(cs.buf), cs.nparts.value, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ctypes.c_float(momentum),
                                  ctypes.c_float(eps), ptr(res), int(relu), ptr(y), ptr(sm), ptr(si), ptr(workspace(x.device)), stream_of(x)), "tf_bn_fwd_parts_f32")
    _hbm_end(_h, "batchnorm forward (statistics from the producer's epilogue: finalize + apply)", 4 * x.numel() * (2 + (res is not None)))
    return y, sm, si


# ---- BatchNorm apply folded into the consumers (YBlockFn's conv2 -> BN -> ReLU -> SE segment; csrc/reduce.cpp, csrc/se.cpp)
FUSE_BN_SE = os.environ.get("TF_FUSE_BN_SE", "1") != "0"
SE_FUSED_MAX_C = 3072      # csrc/se.cpp: the fused excitation kernels hold a sample's squeezed vector in LDS (wider blocks take the generic linear path)


def bn_finalize_parts(cs, gamma, beta, rmean, rvar, momentum=0.1, eps=1e-5):
    """The finalize half of bn_fwd_parts: (coef = [scale | shift] (2C,), save_mean, save_invstd); running statistics updated."""
    C = cs.C
    coef = torch.empty(2 * C, dtype=torch.float32, device=cs.buf.device)
    sm = torch.empty(C, dtype=torch.float32, device=cs.buf.device)
    si = torch.empty_like(sm)
    check(L().tf_bn_finalize_parts_f32(ptr(cs.buf), cs.nparts.value, cs.rows, C, ptr(gamma), ptr(beta), ptr(rmean), ptr(rvar), ctypes.c_float(momentum),
                                       ctypes.c_float(eps), ptr(sm), ptr(si), ptr(coef), stream_of(coef)), "tf_bn_finalize_parts_f32")
    return coef, sm, si


FUSE_BN_CONV = os.environ.get("TF_FUSE_BN_CONV", "1") != "0"


def grouped_bnrelu_ok(x_shape, C, groups, stride):
    """conv1 -> BatchNorm -> ReLU -> grouped conv2 of a RegNetY bottleneck with the BatchNorm apply folded into conv2 (csrc/conv_grouped.cpp): the
    per-group direct kernels (3x3 / stride 1, group width 24) with output statistics."""
    if stride == 2:      # the direct stride-2 kernels (opt-in) take the same folded coefficients
        rows = x_shape[0] * ((x_shape[1] - 1) // 2 + 1) * ((x_shape[2] - 1) // 2 + 1)
        return FUSE_BN_CONV and _grouped_s2_ok(x_shape, C, C, 3, 2, 1, groups) and want_colstat(rows)
    rows = x_shape[0] * x_shape[1] * x_shape[2]
    return FUSE_BN_CONV and _grouped_ok(x_shape, C, C, 3, stride, 1, groups) and want_colstat(rows) and not _direct_ok(x_shape, C, C, 3, stride, 1, groups)


def grouped_bnrelu_fwd(x, coef, w, stride=1):
    """y = grouped_conv3x3(max(x * scale + shift, 0)) + BatchNorm statistics of y, x = the RAW output of the preceding convolution, coef = [scale | shift]
    of its BatchNorm (bn_finalize_parts): the normalised activation is never written.  Returns (y, ColStat)."""
    B, H, W, C = x.shape
    if stride == 2:
        Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
        y = torch.empty(B, Ho, Wo, C, dtype=torch.float32, device=x.device)
        _e = _census_begin()
        cs = ColStat(B * Ho * Wo, C, x.device, max_parts=L().tf_conv3x3_grouped_colstat_parts())
        check(L().tf_conv3x3_grouped_s2_fwd_f32(ptr(_c(x)), ptr(coef), wptr(w), ptr(y), B, H, W, C, ptr(cs.buf), byref(cs.nparts), stream_of(x)), "tf_conv3x3_grouped_s2_fwd_f32")
        _census_end(_e, "conv fwd g2", (B, H, W, C, C, 3, 2, C // 24), 2.0 * B * Ho * Wo * C * 24 * 9)
        return y, cs
    y = torch.empty_like(x)
    _e = _census_begin()
    cs = ColStat(B * H * W, C, x.device, max_parts=L().tf_conv3x3_grouped_colstat_parts())
    check(L().tf_conv3x3_grouped_bnrelu_fwd_colstat_f32(ptr(_c(x)), ptr(coef), wptr(w), ptr(y), B, H, W, C, ptr(cs.buf), byref(cs.nparts), stream_of(x)),
          "tf_conv3x3_grouped_bnrelu_fwd_colstat_f32")
    _census_end(_e, "conv fwd g", (B, H, W, C, C, 3, 1, C // 24), 2.0 * B * H * W * C * 24 * 9)
    return y, cs


def grouped_bnrelu_wgrad(dy, x, coef, dw, accumulate=True, stride=1):
    """dW (+)= grouped weight gradient against max(x * scale + shift, 0) (the activation grouped_bnrelu_fwd never stored)."""
    def run():
        B, H, W, C = x.shape
        _e = _census_begin()
        if stride == 2:
            check(L().tf_conv3x3_grouped_s2_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), ptr(coef), wptr(dw), B, H, W, C, int(accumulate), ptr(_grouped_ws(x.device)), stream_of(dy)),
                  "tf_conv3x3_grouped_s2_wgrad_f32")
            _census_end(_e, "conv wgrad g2", (B, H, W, C, C, 3, 2, C // 24), 2.0 * dy.shape[0] * dy.shape[1] * dy.shape[2] * C * 24 * 9)
            return
        check(L().tf_conv3x3_grouped_bnrelu_wgrad_f32(ptr(_c(dy)), ptr(_c(x)), ptr(coef), wptr(dw), B, H, W, C, int(accumulate), ptr(_grouped_ws(x.device)),
                                                      stream_of(dy)), "tf_conv3x3_grouped_bnrelu_wgrad_f32")
        _census_end(_e, "conv wgrad g", (B, H, W, C, C, 3, 1, C // 24), 2.0 * B * H * W * C * 24 * 9)
    run()
    return dw


def se_squeeze_excite_bn_fwd(y, coef, w1, b1, w2, b2):
    """SE squeeze + excitation on z = relu(y * scale + shift) WITHOUT materialising z: chunk sums of z per sample (one launch), finished inside
    the excitation kernel.  y (B, H, W, C) -> (s (B, C) squeezed means, g1 (B, Cr), gate (B, C) pre-sigmoid)."""
    B, H, W, C = y.shape
    Cr = w1.shape[0]
    nch = ctypes.c_int(0)
    ws = workspace(y.device)
    check(L().tf_colsum_bnrelu_parts_f32(ptr(_c(y)), ptr(coef), B, H * W, C, ptr(ws), byref(nch), stream_of(y)), "tf_colsum_bnrelu_parts_f32")
    s = torch.empty(B, C, dtype=torch.float32, device=y.device)
    buf = torch.empty(2, B, Cr, dtype=torch.float32, device=y.device)
    g1 = buf[0]
    g1._bwd_scratch = buf[1]
    gate = torch.empty(B, C, dtype=torch.float32, device=y.device)
    check(L().tf_se_excite_fwd_parts_f32(ptr(ws), nch.value, ctypes.c_float(1.0 / (H * W)), wptr(w1), ptr(b1), wptr(w2), ptr(b2), B, C, Cr, ptr(s), ptr(g1),
                                         ptr(gate), ptr(buf[1]), stream_of(y)), "tf_se_excite_fwd_parts_f32")
    return s, g1, gate


def se_scale_bn_fwd(y, coef, gate):
    """relu(y * scale + shift) * sigmoid(gate[b, c]) in one pass."""
    B, H, W, C = y.shape
    out = torch.empty_like(y)
    check(L().tf_se_scale_bn_fwd_f32(ptr(_c(y)), ptr(coef), ptr(_c(gate)), ptr(out), B, H * W, C, stream_of(y)), "tf_se_scale_bn_fwd_f32")
    return out


def se_gate_excite_bn_bwd(dz2s, y, coef, gate, s, g1, w1, w2, dw1, db1, dw2, db2):
    """Backward of the excitation fed by the gate gradient sum_hw dz2s * relu(y * scale + shift) (chunk sums finished inside the fc2 kernel);
    accumulates the four parameter gradients, returns ds (B, C)."""
    B, H, W, C = y.shape
    Cr = w1.shape[0]
    nch = ctypes.c_int(0)
    ws = workspace(y.device)
    check(L().tf_se_gate_grad_parts_f32(ptr(_c(dz2s)), ptr(_c(y)), ptr(coef), B, H * W, C, ptr(ws), byref(nch), stream_of(y)), "tf_se_gate_grad_parts_f32")
    ds = torch.empty(B, C, dtype=torch.float32, device=y.device)
    scratch = getattr(g1, "_bwd_scratch", None)
    zeroed = scratch is not None
    if zeroed:
        g1._bwd_scratch = None
    else:
        scratch = torch.empty(B, Cr, dtype=torch.float32, device=y.device)
    check(L().tf_se_excite_bwd_parts_f32(ptr(ws), nch.value, ptr(_c(gate)), ptr(_c(s)), ptr(_c(g1)), wptr(w1), wptr(w2), B, C, Cr, wptr(dw1), ptr(db1), wptr(dw2),
                                         ptr(db2), ptr(ds), ptr(scratch), int(zeroed), stream_of(y)), "tf_se_excite_bwd_parts_f32")
    return ds


def bn_bwd_remask(dz, x, coef, gamma, sm, si, dgamma, dbeta):
    """BatchNorm backward behind a ReLU whose output was never stored (mask = [x * scale + shift > 0])."""
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    check(L().tf_bn_bwd_remask_f32(ptr(_c(dz)), ptr(_c(x)), ptr(coef), rows, C, ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dgamma), ptr(dbeta),
                                   ptr(workspace(x.device)), stream_of(x)), "tf_bn_bwd_remask_f32")
    return dx


FUSE_SE_BN_BWD = os.environ.get("TF_FUSE_SE_BN_BWD", "1") != "0"


def bn_bwd_remask_se(dy, gate, dmean, x, coef, gamma, sm, si, dgamma, dbeta):
    """se_scale_bwd_x + bn_bwd_remask in one reduction and one apply pass: the BatchNorm's incoming gradient dy * sigmoid(gate) + dmean / HW is
    recomputed by both passes instead of being written.  x (B, H, W, C) raw convolution output, gate / dmean (B, C)."""
    B, H, W, C = x.shape
    dx = torch.empty_like(x)
    _h = _hbm_begin()
    check(L().tf_bn_bwd_remask_se_f32(ptr(_c(dy)), ptr(gate), ptr(_c(dmean)), B, H * W, C, ptr(_c(x)), ptr(coef), ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dgamma),
                                      ptr(dbeta), ptr(workspace(x.device)), stream_of(x)), "tf_bn_bwd_remask_se_f32")
    _hbm_end(_h, "batchnorm backward (reduce + finalize + apply)", 4 * x.numel() * 5)
    return dx


def bn_bwd(dz, z, x, gamma, sm, si, dgamma, dbeta, want_dres=False):
    C = x.shape[-1]
    rows = x.numel() // C
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    zacc = zero_scratch(32 * C, x.device) if not _DBG_LEGACY_BNB else None
    _h = _hbm_begin()
    check(L().tf_bn_bwd_f32(ptr(_c(dz)), ptr(z), ptr(_c(x)), rows, C, ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dres), ptr(dgamma), ptr(dbeta),
                            ptr(workspace(x.device)), ptr(zacc), stream_of(x)), "tf_bn_bwd_f32")
    # reduce pass reads dz, x (+ z); apply pass reads them again and writes dx (+ dres)
    _hbm_end(_h, "batchnorm backward (reduce + finalize + apply)", 4 * x.numel() * ((2 + (z is not None)) * 2 + 1 + (dres is not None)))
    return dx, dres


FUSE_DROPOUT = os.environ.get("TF_FUSE_DROPOUT", "1") != "0"      # A/B switch of round 5: resid_drop + residual add in the producing GEMM's epilogue
LN_FWD16 = os.environ.get("TF_LN_FWD16", "1") != "0"      # A/B switch of round 5: ln1 / ln2 of a 16-bit-storage Block write the operand copies themselves


COLSUM_MULTI = os.environ.get("TF_COLSUM_MULTI", "1") != "0"      # the Block's bias gradients in one launch (A/B switch)


def colsum_multi(pairs):
    """out += x.sum(0) for every (x (rows, C) row-major view, out (C,)) pair in ONE launch (same row count; the bias gradients of a Block).
    Falls back to one colsum per pair when a matrix does not meet the vector layout (C % 4, 16-byte alignment)."""
    rows = pairs[0][0].shape[0]
    ok = 1 <= len(pairs) <= 8 and all(x.dim() == 2 and x.shape[0] == rows and x.stride(1) == 1 and x.shape[1] % 4 == 0 and x.stride(0) % 4 == 0 and
                                      x.data_ptr() % 16 == 0 and o.is_contiguous() and o.numel() == x.shape[1] for x, o in pairs)
    if not ok:
        for x, o in pairs:
            colsum(x.contiguous(), 1, rows, x.shape[1], 1.0, out=o.view(1, -1), accumulate=True)
        return
    n = len(pairs)
    xs = (ctypes.c_void_p * n)(*[x.data_ptr() for x, _ in pairs])
    outs = (ctypes.c_void_p * n)(*[o.data_ptr() for _, o in pairs])
    Cs = (ctypes.c_int * n)(*[x.shape[1] for x, _ in pairs])
    lds = (ctypes.c_long * n)(*[x.stride(0) for x, _ in pairs])
    check(L().tf_colsum_multi_f32(n, xs, Cs, lds, outs, rows, stream_of(pairs[0][0])), "tf_colsum_multi_f32")


def colsum(x, nseg, rows_per_seg, C, scale=1.0, mask=None, out=None, accumulate=False, pooled=False):
    """pooled=True (internal temporaries only): the result lives in a zero-arena slice and is produced by ONE atomically accumulating
    launch; it is valid until the next LidarCenterNet.forward."""
    if out is None:
        if pooled:
            out, accumulate = zero_scratch(nseg * C, x.device).view(nseg, C), True
        else:
            out = torch.empty(nseg, C, dtype=torch.float32, device=x.device)
    check(L().tf_colsum_f32(ptr(_c(x)), ptr(mask), nseg, rows_per_seg, C, ctypes.c_float(scale), ptr(out), int(accumulate),
                            ptr(workspace(x.device)), stream_of(x)), "tf_colsum_f32")
    return out


def se_scale_fwd(x, gate):
    B, H, W, C = x.shape
    y = torch.empty_like(x)
    check(L().tf_se_scale_fwd_f32(ptr(_c(x)), ptr(_c(gate)), ptr(y), B, H * W, C, stream_of(x)), "tf_se_scale_fwd_f32")
    return y


def se_scale_bwd_gate(dy, x, gate):
    B, H, W, C = x.shape
    dgate = torch.empty(B, C, dtype=torch.float32, device=x.device)
    check(L().tf_se_scale_bwd_gate_f32(ptr(_c(dy)), ptr(_c(x)), ptr(_c(gate)), ptr(dgate), B, H * W, C, ptr(workspace(x.device)), stream_of(x)),
          "tf_se_scale_bwd_gate_f32")
    return dgate


def se_scale_bwd_x(dy, gate, dmean, shape, out=None, accumulate=False):
    """dx (+)= dy * sigmoid(gate) + dmean / HW; either part optional (dy=None: global-avg-pool backward)."""
    B, H, W, C = shape
    if out is None:
        out = torch.empty(shape, dtype=torch.float32, device=(dy if dy is not None else dmean).device)
    check(L().tf_se_scale_bwd_x_f32(ptr(dy), ptr(gate), ptr(dmean), ptr(out), B, H * W, C, int(accumulate), stream_of(out)), "tf_se_scale_bwd_x_f32")
    return out


def se_excite_fwd(s, w1, b1, w2, b2):
    """Fused SE excitation: returns (g1 (B, Cr) post-ReLU, gate (B, C) pre-sigmoid)."""
    B, C = s.shape
    Cr = w1.shape[0]
    buf = torch.empty(2, B, Cr, dtype=torch.float32, device=s.device)
    g1 = buf[0]
    g1._bwd_scratch = buf[1]            # (B, Cr) accumulator of the backward, cleared by the forward kernel (saves one launch per SE block)
    gate = torch.empty(B, C, dtype=torch.float32, device=s.device)
    check(L().tf_se_excite_fwd_f32(ptr(_c(s)), wptr(w1), ptr(b1), wptr(w2), ptr(b2), B, C, Cr, ptr(g1), ptr(gate), ptr(buf[1]), stream_of(s)),
          "tf_se_excite_fwd_f32")
    return g1, gate


def se_excite_bwd(dgate, s, g1, w1, w2, dw1, db1, dw2, db2):
    """Accumulates the four parameter gradients, returns ds (B, C)."""
    B, C = s.shape
    Cr = w1.shape[0]
    ds = torch.empty(B, C, dtype=torch.float32, device=s.device)
    scratch = getattr(g1, "_bwd_scratch", None)      # cleared by se_excite_fwd; valid for ONE backward
    zeroed = scratch is not None
    if zeroed:
        g1._bwd_scratch = None
    else:
        scratch = torch.empty(B, Cr, dtype=torch.float32, device=s.device)
    check(L().tf_se_excite_bwd_f32(ptr(_c(dgate)), ptr(_c(s)), ptr(_c(g1)), wptr(w1), wptr(w2), B, C, Cr, wptr(dw1), ptr(db1), wptr(dw2), ptr(db2),
                                   ptr(ds), ptr(scratch), int(zeroed), stream_of(s)), "tf_se_excite_bwd_f32")
    return ds


# ------------------------------------------------------------------------------------------ resampling
def pool_tokens_fwd(x, oh, ow, pos, tok, tok_off, bvec=None):
    B, H, W, C = x.shape
    check(L().tf_pool_tokens_fwd_f32(ptr(_c(x)), B, H, W, C, oh, ow, ptr(pos), ptr(bvec), ptr(tok), tok.shape[1], tok_off, stream_of(x)),
          "tf_pool_tokens_fwd_f32")
    return tok


def pool_tokens_bwd(dtok, shape, oh, ow, tok_off, add=None):
    """dx = add + pool^T(dtok)."""
    B, H, W, C = shape
    out = torch.empty(shape, dtype=torch.float32, device=dtok.device)
    check(L().tf_pool_tokens_bwd_f32(ptr(_c(dtok)), B, H, W, C, oh, ow, dtok.shape[1], tok_off, ptr(out), ptr(add), stream_of(dtok)),
          "tf_pool_tokens_bwd_f32")
    return out


class BilinearDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int) for n in ("B", "C", "Hi", "Wi", "Ho", "Wo")] + \
               [(n, ctypes.c_int64) for n in ("sb_i", "sc_i", "sh_i", "sw_i", "sb_o", "sc_o", "sh_o", "sw_o")] + [("align_corners", ctypes.c_int)]


def _bl_desc(B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, align, in_strides=None):
    si = in_strides or ((Hi * Wi * C, 1, Wi * C, C) if in_nhwc else (C * Hi * Wi, Hi * Wi, Wi, 1))
    so = (Ho * Wo * C, 1, Wo * C, C) if out_nhwc else (C * Ho * Wo, Ho * Wo, Wo, 1)
    return BilinearDesc(B, C, Hi, Wi, Ho, Wo, *si, *so, int(align))


def bilinear_fwd(x, B, C, Hi, Wi, Ho, Wo, in_nhwc=True, out_nhwc=True, align_corners=False, add=None, out=None, in_strides=None):
    """Up-sample x to (Ho, Wo).  x is NHWC / NCHW, or any layout described by ``in_strides`` =
    element strides (batch, channel, row, col) relative to x.data_ptr().  Returns NHWC or NCHW."""
    d = _bl_desc(B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, align_corners, in_strides)
    if out is None:
        out = torch.empty((B, Ho, Wo, C) if out_nhwc else (B, C, Ho, Wo), dtype=torch.float32, device=x.device)
    check(L().tf_bilinear_fwd_f32(byref(d), ptr(x), ptr(out), ptr(add), stream_of(x)), "tf_bilinear_fwd_f32")
    return out


def bilinear_bwd(dy, B, C, Hi, Wi, Ho, Wo, in_nhwc=True, out_nhwc=True, align_corners=False, out=None, accumulate=False, in_strides=None):
    d = _bl_desc(B, C, Hi, Wi, Ho, Wo, in_nhwc, out_nhwc, align_corners, in_strides)
    if out is None:
        out = torch.empty((B, Hi, Wi, C) if in_nhwc else (B, C, Hi, Wi), dtype=torch.float32, device=dy.device)
    check(L().tf_bilinear_bwd_f32(byref(d), ptr(_c(dy)), ptr(out), int(accumulate), stream_of(dy)), "tf_bilinear_bwd_f32")
    return out


def gather_sum_fwd(src, idx, Hs, Ws):
    """G1: out[b,i,:] = sum_k src[b, idx[b,i,k,1]*Ws + idx[b,i,k,0], :]; src (B, Hs*Ws, E), idx (B, n, K, 2) int64 (x, y)."""
    B, S, E = src.shape
    assert S == Hs * Ws and idx.dtype == torch.int64 and idx.dim() == 4 and idx.shape[0] == B and idx.shape[3] == 2
    n, K = idx.shape[1], idx.shape[2]
    out = torch.empty(B, n, E, dtype=torch.float32, device=src.device)
    check(L().tf_gather_sum_fwd_f32(ptr(_c(src)), ptr(_c(idx)), B, Hs, Ws, E, n, K, ptr(out), stream_of(src)), "tf_gather_sum_fwd_f32")
    return out


def gather_sum_bwd(dout, idx, Hs, Ws, out=None, accumulate=False):
    """Transposed gather of gather_sum_fwd (fixed summation order): dsrc (B, Hs*Ws, E)."""
    B, n, E = dout.shape
    K = idx.shape[2]
    if out is None:
        assert not accumulate
        out = torch.empty(B, Hs * Ws, E, dtype=torch.float32, device=dout.device)
    check(L().tf_gather_sum_bwd_f32(ptr(_c(dout)), ptr(_c(idx)), B, Hs, Ws, E, n, K, ptr(_c(out)), int(accumulate), stream_of(dout)), "tf_gather_sum_bwd_f32")
    return out


# ------------------------------------------------------------------------------------------ losses
def ce_fwd(logits, target, class_w=None):
    """logits (..., C) NHWC, target (...) int64 -> (loss 0-dim, dlogits unscaled, inv_wsum 1-elem)."""
    C = logits.shape[-1]
    rows = logits.numel() // C
    dl = torch.empty_like(logits)
    out = torch.empty(2, dtype=torch.float32, device=logits.device)
    check(L().tf_ce_fwd_f32(ptr(_c(logits)), ptr(_c(target)), ptr(class_w), ctypes.c_int64(rows), C, ptr(dl), ptr(out), ptr(out[1:]),
                            ptr(workspace(logits.device)), stream_of(logits)), "tf_ce_fwd_f32")
    return out[0], dl, out[1:]


def l1_fwd(pred, target, use_sigmoid=False):
    n = pred.numel()
    dp = torch.empty_like(pred)
    out = torch.empty(1, dtype=torch.float32, device=pred.device)
    check(L().tf_l1_fwd_f32(ptr(_c(pred)), ptr(_c(target)), ctypes.c_int64(n), int(use_sigmoid), ptr(dp), ptr(out), ptr(workspace(pred.device)),
                            stream_of(pred)), "tf_l1_fwd_f32")
    return out[0], dp


def scale_dev_(x, a=None, b=None, mult=1.0):
    check(L().tf_scale_dev_f32(ptr(x), ctypes.c_int64(x.numel()), ptr(a), ptr(b), ctypes.c_float(mult), stream_of(x)), "tf_scale_dev_f32")
    return x


def centernet_targets(label, fh, fw, ratio_w, ratio_h, nbins):
    B, nbox, _ = label.shape
    tgtf = torch.empty(B, fh, fw, 8, dtype=torch.float32, device=label.device)
    tgti = torch.empty(B, fh, fw, 2, dtype=torch.int32, device=label.device)
    cnt = torch.empty(B, dtype=torch.int32, device=label.device)
    check(L().tf_centernet_targets_f32(ptr(_c(label)), B, nbox, fh, fw, ctypes.c_float(ratio_w), ctypes.c_float(ratio_h), nbins, ptr(tgtf), ptr(tgti),
                                       ptr(cnt), stream_of(label)), "tf_centernet_targets_f32")
    return tgtf, tgti, cnt


def centernet_loss_fwd(pred, tgtf, tgti, cnt, nbins):
    B, fh, fw, _ = pred.shape
    losses = torch.empty(7, dtype=torch.float32, device=pred.device)
    check(L().tf_centernet_loss_fwd_f32(ptr(_c(pred)), ptr(tgtf), ptr(tgti), ptr(cnt), B, fh, fw, nbins, ptr(losses), ptr(workspace(pred.device)),
                                        stream_of(pred)), "tf_centernet_loss_fwd_f32")
    return losses


def centernet_loss_bwd(pred, tgtf, tgti, cnt, gup, nbins):
    B, fh, fw, _ = pred.shape
    dpred = torch.empty_like(pred)
    check(L().tf_centernet_loss_bwd_f32(ptr(_c(pred)), ptr(tgtf), ptr(tgti), ptr(cnt), ptr(gup), B, fh, fw, nbins, ptr(dpred), stream_of(pred)),
          "tf_centernet_loss_bwd_f32")
    return dpred


# ------------------------------------------------------------------------------------------ misc
def centernet_decode(pred, nbins, k=100, kernel=3, ratio=4.0):
    """decode_heatmap (model.py:436-497): pred (B, fh, fw, 9+bins) -> (B, k, 8) boxes [x, y, w, h, yaw, vel, brake, score]."""
    B, fh, fw, P = pred.shape
    assert P == 9 + nbins
    out = torch.empty(B, k, 8, dtype=torch.float32, device=pred.device)
    check(L().tf_centernet_decode_f32(ptr(_c(pred)), B, fh, fw, nbins, k, kernel, ctypes.c_float(ratio), ptr(out), stream_of(pred)), "tf_centernet_decode_f32")
    return out


def relu_mask(dy, y, out=None):
    if out is None:
        out = torch.empty_like(dy)
    check(L().tf_relu_mask_f32(ptr(_c(dy)), ptr(_c(y)), ptr(out), ctypes.c_int64(dy.numel()), stream_of(dy)), "tf_relu_mask_f32")
    return out


def sigmoid(x):
    y = torch.empty_like(x)
    check(L().tf_sigmoid_f32(ptr(_c(x)), ptr(y), ctypes.c_int64(x.numel()), stream_of(x)), "tf_sigmoid_f32")
    return y


def axpby(a, b=None, alpha=1.0, beta=1.0, out=None):
    if out is None:
        out = torch.empty_like(a)
    check(L().tf_axpby_f32(ptr(_c(a)), ptr(b), ptr(out), ctypes.c_float(alpha), ctypes.c_float(beta), ctypes.c_int64(a.numel()), stream_of(a)),
          "tf_axpby_f32")
    return out


def weighted_sum(terms, weights, out=None):
    """sum_i weights[i] * terms[i] over <= 16 device scalars (separate allocations), added in index order: one launch."""
    n = len(terms)
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=terms[0].device)
    arr = (c_p * n)(*[ptr(t) for t in terms])
    w = (ctypes.c_float * n)(*[float(v) for v in weights])
    check(L().tf_weighted_sum_f32(arr, w, n, ptr(out), stream_of(terms[0])), "tf_weighted_sum_f32")
    return out


def weighted_sum_bwd(dtotal, weights, device):
    """(n,) tensor of weights[i] * dtotal (dtotal None = 1): the gradients of weighted_sum's terms."""
    n = len(weights)
    out = torch.empty(n, dtype=torch.float32, device=device)
    w = (ctypes.c_float * n)(*[float(v) for v in weights])
    check(L().tf_weighted_sum_bwd_f32(ptr(dtotal), w, n, ptr(out), stream_of(out)), "tf_weighted_sum_bwd_f32")
    return out


def dropout(x, seed, site, p, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(L().tf_dropout_f32(ptr(_c(x)), ptr(out), ctypes.c_int64(x.numel()), ptr(seed), ctypes.c_uint32(site), ctypes.c_float(p), stream_of(x)),
          "tf_dropout_f32")
    return out


def dropout_add(x, res, seed, site, p, out=None):
    """res + dropout(x) in one launch (same mask as dropout(x, seed, site, p))."""
    if out is None:
        out = torch.empty_like(x)
    check(L().tf_dropout_add_f32(ptr(_c(x)), ptr(_c(res)), ptr(out), ctypes.c_int64(x.numel()), ptr(seed), ctypes.c_uint32(site), ctypes.c_float(p),
                                 stream_of(x)), "tf_dropout_add_f32")
    return out


def gru_waypoints_fwd(z0, target_point, gru, outl, pred_len, shift_x):
    B, H = z0.shape
    nin = gru.weight_ih.shape[1]
    L().tf_gru_waypoints_cache_floats.restype = ctypes.c_long
    cache = torch.empty(L().tf_gru_waypoints_cache_floats(B, pred_len), dtype=torch.float32, device=z0.device)
    wp = torch.empty(B, pred_len, 2, dtype=torch.float32, device=z0.device)
    check(L().tf_gru_waypoints_fwd_f32(ptr(_c(z0)), ptr(target_point), ptr(gru.weight_ih), ptr(gru.weight_hh), ptr(gru.bias_ih), ptr(gru.bias_hh),
                                       ptr(outl.weight), ptr(outl.bias), B, H, pred_len, nin, ctypes.c_float(shift_x), ptr(wp), ptr(cache),
                                       stream_of(z0)), "tf_gru_waypoints_fwd_f32")
    return wp, cache


def gru_waypoints_bwd(dwp, cache, gru, outl, grads, B, H, pred_len):
    """grads = (dw_ih, dw_hh, db_ih, db_hh, dw_out, db_out) accumulation buffers; returns dz0."""
    dz0 = torch.empty(B, H, dtype=torch.float32, device=dwp.device)
    check(L().tf_gru_waypoints_bwd_f32(ptr(_c(dwp)), ptr(cache), ptr(gru.weight_ih), ptr(gru.weight_hh), ptr(outl.weight), B, H, pred_len,
                                       gru.weight_ih.shape[1], ptr(dz0), *[ptr(g) for g in grads], stream_of(dwp)), "tf_gru_waypoints_bwd_f32")
    return dz0


def adamw_(p, g, m, v, state, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, grad_scale=1.0):
    """grad_scale: the gradients are multiplied by it on the way in (1 / loss scale of the fp16 mode; 1 / world of a summed all-reduce)."""
    _h = _hbm_begin()
    if grad_scale != 1.0:
        check(L().tf_adamw_scaled_f32(ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_int64(p.numel()), ptr(state), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                      ctypes.c_float(eps), ctypes.c_float(weight_decay), ctypes.c_float(grad_scale), stream_of(p)), "tf_adamw_scaled_f32")
    else:
        check(L().tf_adamw_f32(ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_int64(p.numel()), ptr(state), ctypes.c_float(beta1), ctypes.c_float(beta2),
                               ctypes.c_float(eps), ctypes.c_float(weight_decay), stream_of(p)), "tf_adamw_f32")
    _hbm_end(_h, "adamw (28 B / parameter: p, g, m, v read; p, m, v written)", 28 * p.numel())


def adamw_dynscale_(p, g, m, v, state, ls_state, g_check, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01):
    """AdamW under the device-side dynamic loss scale ``ls_state`` = [scale, clean steps, found_inf, growth interval] (tf_adamw_dynscale_f32):
    non-finite check over ``g_check`` (the whole gradient arena), skip-or-update with gradients / scale, scale halved / doubled."""
    _h = _hbm_begin()
    check(L().tf_adamw_dynscale_f32(ptr(p), ptr(g), ptr(m), ptr(v), ctypes.c_int64(p.numel()), ptr(state), ctypes.c_float(beta1), ctypes.c_float(beta2),
                                    ctypes.c_float(eps), ctypes.c_float(weight_decay), ptr(g_check), ctypes.c_int64(g_check.numel()), ptr(ls_state), stream_of(p)),
          "tf_adamw_dynscale_f32")
    _hbm_end(_h, "adamw (28 B / parameter: p, g, m, v read; p, m, v written)", 28 * p.numel() + 4 * g_check.numel())


def cast_bf16(x, out=None):
    """fp32 -> bf16 (round to nearest even)."""
    if out is None:
        out = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    check(L().tf_cast_f32_bf16(ptr(_c(x)), ptr(_c(out)), ctypes.c_int64(x.numel()), stream_of(x)), "tf_cast_f32_bf16")
    return out


def cast_f32(x, out, scale=1.0):
    """bf16 -> fp32 with a scale (out = x * scale)."""
    check(L().tf_cast_bf16_f32(ptr(_c(x)), ptr(_c(out)), ctypes.c_int64(x.numel()), ctypes.c_float(scale), stream_of(x)), "tf_cast_bf16_f32")
    return out


def lidar_hist(points, num_points=None):
    """points (B, N, >=3) float32 -> (B, 2, 256, 256) float32 BEV histogram (data.py:446-470)."""
    B, N, S = points.shape
    out = torch.empty(B, 2, 256, 256, dtype=torch.float32, device=points.device)
    check(L().tf_lidar_hist_ws_f32(ptr(_c(points)), ptr(num_points), B, N, S, ctypes.c_void_p(_hist_ws(B, points.device).data_ptr()), ptr(out),
                                   stream_of(points)), "tf_lidar_hist_ws_f32")
    return out


_HIST_WS = {}


def _hist_ws(B, device):
    """The histogram kernels' int32 cell counters: zeroed once here, left zeroed by every call (two launches instead of three).  One buffer per
    (device, stream): calls on different streams must not share counters."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _HIST_WS.get(key)
    if ws is None or ws.numel() < B * 2 * 256 * 256:
        ws = _HIST_WS[key] = torch.zeros(B * 2 * 256 * 256, dtype=torch.int32, device=device)
    return ws


# ------------------------------------------------------------------------------------------ H2 PointPillars front-end
def _scan(flags):
    n = flags.numel()
    out = torch.empty_like(flags)
    total = torch.empty(1, dtype=torch.int32, device=flags.device)
    ws = torch.empty(n // 1024 + 2, dtype=torch.int32, device=flags.device)
    check(L().tf_exclusive_scan_i32(ptr(flags), ctypes.c_int64(n), ptr(out), ptr(total), ptr(ws), stream_of(flags)), "tf_exclusive_scan_i32")
    return out, total


_PILLAR_V2 = os.environ.get("TF_PILLAR_V2", "1") != "0"      # 0: the seven-launch form of rounds 3-5 (A/B, tools/hbm_bench.py)


def pillar_index(points, num_points, min_x, max_x, min_y, max_y, ppm, static=False):
    """point_pillar.py:98-117 without the sort: returns a dict with the compacted points (N,4), 9 decorated features (N,9),
    inverse indices (N) int32, cellkey (P) int32 [= ((b*GX + x_idx)*GY + y_idx), sorted like torch.unique] and the grid dims.
    One host read (N, P) - the reference's unique() synchronises at the same place.
    static=True (hipGraph capture): NO host read - every tensor has its capacity (N -> B * Nmax rows, P -> min(B * Nmax, cells)), ``totals`` =
    (kept points, pillars) stays on the device, rows / slots beyond the counts are zero (points, inv, features) or -1 (cell keys);
    the consumers run over the capacity and the point net's BatchNorm reads its row count from ``totals`` (bn_rows_dev_*).
    Three launches (slab accumulation in LDS, scan, gather + decorate; csrc/pillars.cpp "Round 6"), no fills and no global atomics in either mode."""
    if not _PILLAR_V2:
        return _pillar_index_v1(points, num_points, min_x, max_x, min_y, max_y, ppm, static)
    B, Nmax, Fp = points.shape
    nx, ny = int((max_x - min_x) * ppm), int((max_y - min_y) * ppm)
    GX, GY = nx + 1, ny + 1
    dev = points.device
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
    f, i64 = ctypes.c_float, ctypes.c_int64
    L().tf_pillar_padded_cells.restype = ctypes.c_long
    CP = L().tf_pillar_padded_cells(GX, GY)
    n_all, ncells, nwords, nblk = B * Nmax, B * GX * GY, B * CP // 32, B * ((Nmax + 1023) // 1024)
    points = _c(points)
    keys, small = i32(n_all), i32(2 * nwords + 2 * nblk + 4)          # bitmap | wordprefix (16-byte aligned: nwords % 4 == 0) | blockcnt | blockoff | totals
    bitmap, wordprefix, blockcnt, blockoff, totals = small[:nwords], small[nwords:2 * nwords], small[2 * nwords:2 * nwords + nblk], \
        small[2 * nwords + nblk:2 * nwords + 2 * nblk], small[2 * nwords + 2 * nblk:2 * nwords + 2 * nblk + 2]
    cellsums = torch.empty(B * CP, 4, dtype=torch.int64, device=dev)     # written and read for the occupied cells only
    st = stream_of(points)
    check(L().tf_pillar_mark_f32(ptr(points), ptr(num_points), B, Nmax, Fp, f(min_x), f(max_x), f(min_y), f(max_y), f(ppm), GX, GY, ptr(keys), ptr(bitmap),
                                 ptr(cellsums), ptr(blockcnt), st), "tf_pillar_mark_f32")
    check(L().tf_pillar_rank_scan_i32(ptr(bitmap), i64(B * CP), ptr(blockcnt), nblk, ptr(wordprefix), ptr(blockoff), ptr(totals), st), "tf_pillar_rank_scan_i32")
    if static:
        N, P = n_all, min(n_all, ncells)
    else:
        N, P = (int(v) for v in totals.tolist())
    pts4 = torch.empty(N, 4, dtype=torch.float32, device=dev)
    inv, cellkey = i32(N), i32(P)
    feat = torch.empty(N, 9, dtype=torch.float32, device=dev)
    if N:
        check(L().tf_pillar_gather_decorate_f32(ptr(points), Fp, ptr(keys), B, Nmax, ptr(blockoff), ptr(wordprefix), ptr(bitmap), ptr(cellsums), ptr(totals), GX, GY,
                                                f(ppm), f(min_x), f(min_y), ptr(pts4), ptr(inv), ptr(feat), ptr(cellkey), i64(P), int(bool(static)), st),
              "tf_pillar_gather_decorate_f32")
    return dict(points=pts4, feat=feat, inv=inv, cellkey=cellkey, N=N, P=P, GX=GX, GY=GY, nx=nx, ny=ny, totals=totals, static=bool(static))


def _pillar_index_v1(points, num_points, min_x, max_x, min_y, max_y, ppm, static=False):
    """point_pillar.py:98-117 without the sort: returns a dict with the compacted points (N,4), 9 decorated features (N,9),
    inverse indices (N) int32, cellkey (P) int32 [= ((b*GX + x_idx)*GY + y_idx), sorted like torch.unique] and the grid dims.
    One host read (N, P) - the reference's unique() synchronises at the same place.
    static=True (hipGraph capture): NO host read - every tensor has its capacity (N -> B * Nmax rows, P -> min(B * Nmax, cells)), ``totals`` =
    (kept points, pillars) stays on the device, rows / slots beyond the counts are zero (points, inv, features of the zero point) or -1 (cell keys);
    the consumers run over the capacity and the point net's BatchNorm reads its row count from ``totals`` (bn_rows_dev_*)."""
    B, Nmax, Fp = points.shape
    nx, ny = int((max_x - min_x) * ppm), int((max_y - min_y) * ppm)
    GX, GY = nx + 1, ny + 1
    dev = points.device
    i32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=dev)
    keys, keep, occ = i32(B * Nmax), i32(B * Nmax), i32(B * GX * GY)
    f = ctypes.c_float
    check(L().tf_pillar_keys_f32(ptr(_c(points)), ptr(num_points), B, Nmax, Fp, f(min_x), f(max_x), f(min_y), f(max_y), f(ppm), GX, GY,
                                 ptr(keys), ptr(keep), ptr(occ), stream_of(points)), "tf_pillar_keys_f32")
    # both scans + the cell keys in two launches, ONE host read of (N, P)
    ncells = B * GX * GY
    pos, rank, totals = i32(B * Nmax), i32(ncells), i32(2)
    cellkey_full = torch.full((min(B * Nmax, ncells),), -1, dtype=torch.int32, device=dev) if static else i32(min(B * Nmax, ncells))
    ws = i32((B * Nmax + ncells) // 1024 + 4)
    check(L().tf_pillar_index_scan_i32(ptr(keys), ctypes.c_int64(B * Nmax), ptr(occ), ctypes.c_int64(ncells), ptr(pos), ptr(rank), ptr(cellkey_full), ptr(totals),
                                       ptr(ws), stream_of(points)), "tf_pillar_index_scan_i32")
    if static:
        N, P = B * Nmax, cellkey_full.numel()
        pts4 = torch.zeros(N, 4, dtype=torch.float32, device=dev)
        inv = torch.zeros(N, dtype=torch.int32, device=dev)
    else:
        N, P = (int(v) for v in totals.tolist())
        pts4 = torch.empty(N, 4, dtype=torch.float32, device=dev)
        inv = i32(N)
    feat = torch.empty(N, 9, dtype=torch.float32, device=dev)
    cellkey = cellkey_full[:P]
    sums = torch.empty(P, 4, dtype=torch.int64, device=dev)      # fixed-point (2^-24 m) xyz sums + count: order-independent integer atomics
    if N:
        check(L().tf_pillar_gather_f32(ptr(points), Fp, ptr(keys), ptr(pos), c_p(0), ptr(rank), ctypes.c_int64(B * Nmax), ctypes.c_int64(ncells), P,
                                       ptr(pts4), ptr(inv), ptr(sums), ptr(cellkey), stream_of(points)), "tf_pillar_gather_f32")
        check(L().tf_pillar_decorate_f32(ptr(pts4), ptr(inv), ptr(sums), ptr(cellkey), ctypes.c_int64(N), GX, GY, f(ppm), f(min_x), f(min_y), ptr(feat),
                                         stream_of(points)), "tf_pillar_decorate_f32")
    return dict(points=pts4, feat=feat, inv=inv, cellkey=cellkey, N=N, P=P, GX=GX, GY=GY, nx=nx, ny=ny, totals=totals, static=bool(static))


_bnr_ws = {}


def _bn_rows_ws(C, device):
    key = (str(device), C, torch.cuda.current_stream(device).cuda_stream if getattr(device, "type", str(device)[:4]) == "cuda" else 0)
    ws = _bnr_ws.get(key)
    if ws is None:
        L().tf_bn_rows_dev_ws_floats.restype = ctypes.c_long
        ws = _bnr_ws[key] = torch.empty(L().tf_bn_rows_dev_ws_floats(C), dtype=torch.float32, device=device)
    return ws


def bn_rows_dev_fwd(x, nrows_dev, gamma, beta, running_mean, running_var, momentum, eps, relu=True):
    """BatchNorm1d (+ ReLU) over the first ``nrows_dev[0]`` rows of x (rows_cap, C), the count read on the device; rows beyond are written as zero."""
    rows, C = x.shape
    y = torch.empty_like(x)
    sm, si = torch.empty(C, dtype=torch.float32, device=x.device), torch.empty(C, dtype=torch.float32, device=x.device)
    check(L().tf_bn_rows_dev_fwd_f32(ptr(_c(x)), ptr(nrows_dev), rows, C, ptr(gamma), ptr(beta), ptr(running_mean), ptr(running_var), ctypes.c_float(momentum),
                                     ctypes.c_float(eps), int(relu), ptr(y), ptr(sm), ptr(si), ptr(_bn_rows_ws(C, x.device)), stream_of(x)), "tf_bn_rows_dev_fwd_f32")
    return y, sm, si


def bn_rows_dev_bwd(dz, z, x, nrows_dev, gamma, sm, si, dgamma, dbeta):
    rows, C = x.shape
    dx = torch.empty_like(x)
    check(L().tf_bn_rows_dev_bwd_f32(ptr(_c(dz)), ptr(z), ptr(_c(x)), ptr(nrows_dev), rows, C, ptr(gamma), ptr(sm), ptr(si), ptr(dx), ptr(dgamma), ptr(dbeta),
                                     ptr(_bn_rows_ws(C, x.device)), stream_of(x)), "tf_bn_rows_dev_bwd_f32")
    return dx


def pillar_scatter_max(z, inv, P):
    N, C = z.shape
    pf = torch.empty(P, C, dtype=torch.float32, device=z.device)
    arg = torch.empty(P, C, dtype=torch.int32, device=z.device)
    check(L().tf_pillar_scatter_max_f32(ptr(_c(z)), ptr(inv), ctypes.c_int64(N), C, P, ptr(pf), ptr(arg), stream_of(z)), "tf_pillar_scatter_max_f32")
    return pf, arg


def pillar_canvas(pf, cellkey, B, H, W, GX, GY, extra=None):
    """(B, H, W, C + Ce) NHWC = rot90(scatter_points(...), -1) with the extra NCHW channels appended (model.py:738-742)."""
    P, C = pf.shape
    Ce = 0 if extra is None else extra.shape[1]
    out = torch.empty(B, H, W, C + Ce, dtype=torch.float32, device=pf.device)
    owner = torch.empty(B * H * W, dtype=torch.int32, device=pf.device)
    check(L().tf_pillar_canvas_f32(ptr(pf), ptr(cellkey), P, C, B, H, W, GX, GY, ptr(_c(extra) if extra is not None else None), Ce, ptr(owner), ptr(out),
                                   stream_of(out)), "tf_pillar_canvas_f32")
    return out, owner


def pillar_canvas_bwd(dout, owner, cellkey, inv, arg, C, GX, GY):
    B, H, W, Cs = dout.shape
    N = inv.numel()
    dz = torch.empty(N, C, dtype=torch.float32, device=dout.device)
    check(L().tf_pillar_canvas_bwd_f32(ptr(_c(dout)), ptr(owner), ptr(cellkey), ptr(inv), ptr(arg), ctypes.c_int64(N), C, Cs, GX, GY, H, W, ptr(dz),
                                       stream_of(dout)), "tf_pillar_canvas_bwd_f32")
    return dz


# ------------------------------------------------------------------------------------------ GPU-side batch preparation (csrc/dataprep.cpp)
def lidar_align_hist(points, transforms, num_points=None, return_aligned=False):
    """align (data.py:411-444) + lidar_to_histogram_features (:446-470) for a batch: points (B, N, >=4) float32 as loaded (y negated),
    transforms (B, 4, 4) float64 -> (B, 2, 256, 256) float32 [, aligned cloud (B, N, 4) float32]."""
    B, N, S = points.shape
    out = torch.empty(B, 2, 256, 256, dtype=torch.float32, device=points.device)
    al = torch.empty(B, N, 4, dtype=torch.float32, device=points.device) if return_aligned else None
    t = transforms.to(device=points.device, dtype=torch.float64).contiguous()
    check(L().tf_lidar_align_hist_ws_f64(ptr(_c(points)), ptr(num_points), B, N, S, ctypes.c_void_p(t.data_ptr()),
                                         ctypes.c_void_p(_hist_ws(B, points.device).data_ptr()), ptr(out), ptr(al), stream_of(points)), "tf_lidar_align_hist_ws_f64")
    return (out, al) if return_aligned else out


_corr_cam = None


def corr_camera_constants():
    """focal_x, focal_y, cos / sin of the -60 / +60 degree camera yaws, evaluated with NumPy exactly like data.py:688-712,741-747."""
    global _corr_cam
    if _corr_cam is None:
        import numpy as np
        img_width, img_height, fov_width = 352, 160, 60
        fov_height = np.rad2deg(2.0 * np.arctan((img_height / img_width) * np.tan(0.5 * np.radians(fov_width))))
        focal_x = img_width / (2.0 * np.tan(np.deg2rad(fov_width) / 2.0))
        focal_y = img_height / (2.0 * np.tan(np.deg2rad(fov_height) / 2.0))
        tl, tr = np.radians(-60.0), np.radians(60.0)
        _corr_cam = (ctypes.c_double * 6)(float(focal_x), float(focal_y), float(np.cos(tl)), float(np.sin(tl)), float(np.cos(tr)), float(np.sin(tr)))
    return _corr_cam


def lidar_cam_correspondences(points, num_points=None, seed=0, y_negated=False):
    """lidar_bev_cam_correspondences (data.py:675-842) for a batch of raw clouds (B, N, >= 3) float32 -> (bev_points (B, 8, 8, 5, 2),
    cam_points (B, 22, 5, 5, 2)) int32: the C4 inputs of the geometric-fusion backbone (train.py:280-288)."""
    B, N, S = points.shape
    dev = points.device
    L().tf_lidar_cam_correspondences_ws_bytes.restype = ctypes.c_long
    ws = torch.empty(L().tf_lidar_cam_correspondences_ws_bytes(B, N), dtype=torch.uint8, device=dev)
    bev = torch.empty(B, 8, 8, 5, 2, dtype=torch.int32, device=dev)
    cam = torch.empty(B, 22, 5, 5, 2, dtype=torch.int32, device=dev)
    check(L().tf_lidar_cam_correspondences_f32(ptr(_c(points)), ptr(num_points), B, N, S, int(bool(y_negated)), corr_camera_constants(), ctypes.c_uint32(seed & 0xffffffff),
                                               ptr(ws), ptr(bev), ptr(cam), stream_of(points)), "tf_lidar_cam_correspondences_f32")
    return bev, cam


def image_prep(src, crop_hw, start_y, start_x, mode, lut=None):
    """src (B, Hs, Ws, C) uint8 HWC; mode "rgb" -> (B, C, h, w) float32; "depth" -> (B, h, w) float32 (get_depth); "seg" -> (B, h, w) int64 (LUT)."""
    B, Hs, Ws, C = src.shape
    h, w = crop_hw
    m = {"rgb": 0, "depth": 1, "seg": 2}[mode]
    out = torch.empty((B, C, h, w) if m == 0 else (B, h, w), dtype=torch.int64 if m == 2 else torch.float32, device=src.device)
    sx = start_x.to(device=src.device, dtype=torch.int32).contiguous()
    check(L().tf_image_prep_u8(ctypes.c_void_p(src.contiguous().data_ptr()), B, Hs, Ws, C, h, w, int(start_y), ptr(sx), m,
                               ctypes.c_void_p(lut.data_ptr()) if lut is not None else ctypes.c_void_p(0), ctypes.c_void_p(out.data_ptr()), stream_of(src)),
          "tf_image_prep_u8")
    return out


def bev_prep(encoded, degrees=None):
    """encoded (B, S, S, 3) uint8 RGB top-down image -> (B, 160, 160) int64 labels (decode_pil_to_npy + load_crop_bev_npy, data.py:844-856,586-612)."""
    B, S = encoded.shape[0], encoded.shape[1]
    out = torch.empty(B, 160, 160, dtype=torch.int64, device=encoded.device)
    d = degrees.to(device=encoded.device, dtype=torch.float32).contiguous() if degrees is not None else None
    check(L().tf_bev_prep_u8(ctypes.c_void_p(encoded.contiguous().data_ptr()), B, S, ptr(d), ctypes.c_void_p(out.data_ptr()), stream_of(encoded)), "tf_bev_prep_u8")
    return out
