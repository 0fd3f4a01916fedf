"""Builds tests/emu/libtransfuser_emu.so: the HIP kernel sources compiled for the HOST with
-DTF_EMU against the fiber emulator.  TEST INFRASTRUCTURE ONLY (never used by the product path)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
from transfuser_amd import build as _b  # noqa: E402

CXX = os.environ.get("EMU_CXX", "/opt/rocm/lib/llvm/bin/clang++")
LIB = os.path.join(HERE, "libtransfuser_emu.so")


def build(asan=False, verbose=False):
    flags = ["-O2", "-g", "-std=c++17", "-fPIC", "-DTF_EMU", "-I", HERE, "-I", _b.CSRC, "-Wno-unused-value",
             "-ffp-contract=off", "-Wno-unknown-pragmas", "-Wno-pass-failed"]
    lib = LIB
    objdir = os.path.join(HERE, "build")
    if asan:
        flags += ["-fsanitize=address", "-DEMU_UCONTEXT", "-fno-omit-frame-pointer"]
        lib = LIB.replace(".so", "_asan.so")
        objdir += "_asan"
    objs = _b.compile_objects(CXX, flags, objdir, verbose)
    emu_obj = os.path.join(objdir, "hip_emu.o")
    emu_src = os.path.join(HERE, "hip_emu.cpp")
    if _b._newer(emu_obj, [emu_src, os.path.join(HERE, "hip_emu.h")]):
        subprocess.run([CXX] + flags + ["-c", emu_src, "-o", emu_obj], check=True)
    if _b._newer(lib, objs + [emu_obj]):
        subprocess.run([CXX, "-shared", "-fPIC", "-o", lib] + (["-fsanitize=address"] if asan else []) + objs + [emu_obj], check=True)
    return lib


if __name__ == "__main__":
    print(build(asan="--asan" in sys.argv, verbose=True))
