"""ctypes binding of libtransfuser_hip.so (the C ABI declared in include/transfuser_hip.h).

The product path has NO fallback: if the HIP library is missing, or a tensor is not on a GPU,
the call raises.  ``_install_test_backend`` exists only for tests/emu (the same kernel sources
compiled for the host against a fiber emulator) and is never called from the package.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# TF_HIP_LIB: A/B runs of two builds of the SAME library inside one GPU lease (tools/gpu_round4.sh); never a different backend
LIB_PATH = os.environ.get("TF_HIP_LIB") or os.path.join(_HERE, "libtransfuser_hip.so")

_lib = None
_test_backend = False

c_f = ctypes.c_float
c_i = ctypes.c_int
c_l = ctypes.c_int64
c_p = ctypes.c_void_p


class GemmDesc(ctypes.Structure):
    _fields_ = [("a", c_p), ("b", c_p), ("c", c_p), ("bias", c_p), ("res", c_p),
                ("m", c_i), ("n", c_i), ("k", c_i), ("a_trans", c_i), ("b_trans", c_i),
                ("lda", c_l), ("ldb", c_l), ("ldc", c_l), ("ldres", c_l),
                ("batch", c_i), ("inner", c_i),
                ("sa_outer", c_l), ("sa_inner", c_l), ("sb_outer", c_l), ("sb_inner", c_l), ("sc_outer", c_l), ("sc_inner", c_l),
                ("alpha", c_f), ("relu", c_i), ("accumulate", c_i), ("mask", c_p), ("ldmask", c_l),
                ("splitk_ws", c_p), ("splitk_ws_floats", c_l), ("sk_flags", c_p), ("colstat", c_p), ("colstat_nparts", ctypes.POINTER(ctypes.c_int)),
                ("drop_seed", c_p), ("drop_site", ctypes.c_uint32), ("drop_p", c_f)]


class ConvGeom(ctypes.Structure):
    _fields_ = [(n, c_i) for n in ("B", "Hi", "Wi", "Cin", "Ho", "Wo", "Cout", "ksize", "stride", "pad", "groups")]


def _declare(lib):
    lib.tf_last_error.restype = ctypes.c_char_p
    lib.tf_version.restype = c_i
    if hasattr(lib, "tf_build_id"):           # absent only from a pre-round-4 build selected through TF_HIP_LIB (A/B runs)
        lib.tf_build_id.restype = ctypes.c_char_p
    return lib


def build_id_check(lib):
    """Does the loaded library come from the sources beside it?  (The .so is built in the authoring container and shipped; a GPU box never
    compiles.)  Returns (library id, source id or None when the sources are absent); warns on a mismatch - a stale library is still the
    product path, but every number it produces belongs to other sources."""
    if not hasattr(lib, "tf_build_id"):
        return None, None
    got = lib.tf_build_id().decode()
    try:
        from . import build as _b
        want = _b.source_hash()
    except Exception:
        return got, None
    if got != want and not os.environ.get("TF_HIP_LIB"):
        import warnings
        warnings.warn("transfuser_amd: %s was built from other sources (tf_build_id %s, sources %s): rebuild with `python -m transfuser_amd.build`"
                      % (LIB_PATH, got, want))
    return got, want


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("transfuser_amd: %s not found - build it with `python -m transfuser_amd.build` "
                               "(hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
        _lib = _declare(ctypes.CDLL(LIB_PATH))
        build_id_check(_lib)
    return _lib


def _install_test_backend(cdll):
    """tests/emu only: route the C ABI to the host-emulated build of the same kernels."""
    global _lib, _test_backend
    _lib = _declare(cdll)
    _test_backend = True


def is_test_backend():
    return _test_backend


_TRACE = os.environ.get("TF_TRACE_CALLS", "0") == "1"


def check(rc, what=""):
    if _TRACE:      # debugging aid: wait for every C-ABI call and name it once it has COMPLETED (a GPU fault then belongs to the next call)
        torch.cuda.synchronize()
        import sys
        print("[done] %s" % what, file=sys.stderr, flush=True)
    if rc != 0:
        raise RuntimeError("transfuser_hip %s failed (%d): %s" % (what, rc, load().tf_last_error().decode()))


def stream_of(t):
    if t.is_cuda:
        return c_p(torch.cuda.current_stream(t.device).cuda_stream)
    if not _test_backend:
        raise RuntimeError("transfuser_amd ops need GPU tensors (got device %s); there is no CPU path" % t.device)
    return c_p(0)


def ptr(t):
    if t is None:
        return c_p(0)
    assert t.dtype in (torch.float32, torch.int32, torch.int64, torch.uint8, torch.bfloat16), t.dtype
    return c_p(t.data_ptr())
