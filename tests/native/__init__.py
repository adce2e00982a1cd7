"""Test-only native helpers (tests/native/testkit.hip -> libsf_testkit.so); built by __graft_entry__.build() and on demand."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "testkit.hip")
LIB = os.path.join(HERE, "libsf_testkit.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-std=c++17", "-shared", "-fPIC", SRC, "-o", LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {SRC}:\n{r.stdout}{r.stderr}")
    return LIB


HOOKS_LIB = os.path.join(HERE, "libsatflow_hip_hooks.so")


def build_hooks(force: bool = False) -> str:
    """The product library with convgru_seq.hip compiled a SECOND time with -DSF_TEST_HOOKS (exports sf_convgru_seq_debug: a shorter
    hand-off spin and a half that never sends).  Test infrastructure: selected per process through SATFLOW_HIP_LIB, never shipped."""
    from satflow_amd.build import CSRC, FLAGS, LIBDIR, build_library

    build_library(force=False, verbose=False)
    src = os.path.join(CSRC, "convgru_seq.hip")
    objdir = os.path.join(LIBDIR, "obj")
    deps = [src] + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(LIBDIR, "libsatflow_hip.so")]
    if force or not os.path.exists(HOOKS_LIB) or any(os.path.getmtime(d) > os.path.getmtime(HOOKS_LIB) for d in deps):
        obj = os.path.join(HERE, "convgru_seq_hooks.o")
        r = subprocess.run([HIPCC, *FLAGS, "-DSF_TEST_HOOKS", "-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src} (-DSF_TEST_HOOKS):\n{r.stdout}{r.stderr}")
        objs = [os.path.join(objdir, f) for f in sorted(os.listdir(objdir)) if f.endswith(".o") and f != "convgru_seq.o"] + [obj]
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", HOOKS_LIB], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed for {HOOKS_LIB}:\n{r.stdout}{r.stderr}")
    return HOOKS_LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.sftest_occupy_cus.restype = C.c_int
        _lib.sftest_occupy_cus.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
    return _lib


def occupy_cus(workgroups: int, lds_bytes: int, microseconds: int, stream, census=None) -> None:
    """Park ``workgroups`` one-wave workgroups holding ``lds_bytes`` of LDS each for ``microseconds`` on ``stream`` (a torch stream)."""
    rc = lib().sftest_occupy_cus(workgroups, lds_bytes, microseconds, census.data_ptr() if census is not None else None, stream.cuda_stream)
    if rc != 0:
        raise RuntimeError(f"sftest_occupy_cus failed rc={rc}")
