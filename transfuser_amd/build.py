"""In-tree build of libtransfuser_hip.so (hipcc, gfx950).  `python -m transfuser_amd.build`.

hipcc cross-compiles without a GPU, so this also runs in the authoring container; the built
.so travels to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libtransfuser_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-x", "hip", "-Wno-unused-value",
             "-ffp-contract=off"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".cpp"))


def source_hash():
    """sha256[:16] over csrc/*.cpp, csrc/*.h and include/transfuser_hip.h (sorted): what tf_build_id() of a library built from them returns."""
    import hashlib
    h = hashlib.sha256()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cpp", ".h")))
    files.append(os.path.join(HERE, "..", "include", "transfuser_hip.h"))
    for f in files:
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def compile_objects(cc, flags, objdir, verbose=True):
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdrs.append(os.path.join(HERE, "..", "include", "transfuser_hip.h"))
    jobs = []
    # api.cpp carries the hash of ALL sources (tf_build_id): it is recompiled whenever the hash changes
    bid = source_hash()
    idfile = os.path.join(objdir, "build_id.txt")
    if not os.path.exists(idfile) or open(idfile).read().strip() != bid:
        with open(idfile, "w") as f:
            f.write(bid + "\n")
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        if _newer(obj, [src] + hdrs + ([idfile] if os.path.basename(src) == "api.cpp" else [])):
            jobs.append((src, obj))

    def run(job):
        src, obj = job
        extra = ['-DTF_BUILD_ID="%s"' % bid] if os.path.basename(src) == "api.cpp" else []
        cmd = [cc] + flags + extra + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose:
            print("  cc", os.path.basename(src), flush=True)

    # the engine instantiations (conv_*, gemm_plain_*, gemm_dma_*) take minutes each: start them first so they are not the tail of the build
    heavy = ("conv_fwd", "conv_dgrad", "conv_wgrad", "conv_stem", "gemm_plain_", "gemm_dma_", "gemm.", "conv_direct", "conv_grouped", "attention")
    jobs.sort(key=lambda j: next((i for i, h in enumerate(heavy) if os.path.basename(j[0]).startswith(h)), len(heavy)))
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(run, jobs))
    return [os.path.join(objdir, os.path.basename(s)[:-4] + ".o") for s in sources()]


def build(verbose=True):
    objs = compile_objects(HIPCC, HIP_FLAGS, os.path.join(HERE, "build", "hip"), verbose)
    if _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: %s\n%s" % (r.stdout, r.stderr))
        if verbose:
            print("  ld", LIB, flush=True)
    return LIB


if __name__ == "__main__":
    build()
    print(LIB)
