"""Selects the diagnostic build of the library (csrc/diag.h: A/B knobs and ablation bits) for a measurement tool.

Call use_diag_lib() BEFORE importing kmbart: it builds km-bart_amd/lib/libkmbart_hip_diag.so when it is missing or older than
the sources and points KMB_LIB_PATH at it.  The product library ignores every diagnostic environment variable."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def use_diag_lib():
    if os.environ.get("KMB_LIB_PATH"):
        return os.environ["KMB_LIB_PATH"]
    spec = importlib.util.spec_from_file_location("kmbart_build", os.path.join(ROOT, "km-bart_amd", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = os.path.join(mod.LIBDIR, "libkmbart_hip_diag.so")
    srcs = [os.path.join(mod.CSRC, f) for f in os.listdir(mod.CSRC)]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        lib = mod.build_variant("diag", ["KMB_DIAG"], ("gemm.hip", "gemm_lean.hip", "engine.cpp", "attention.hip"))
    os.environ["KMB_LIB_PATH"] = lib
    return lib
