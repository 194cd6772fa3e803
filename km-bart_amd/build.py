"""Builds libkmbart_hip.so (gfx950) in-tree: `python km-bart_amd/build.py [--force]`.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libkmbart_hip.so")
SOURCES = ["gemm.hip", "gemm_pair.hip", "gemm_lean.hip", "attention.hip", "norm.hip", "embed.hip", "loss.hip", "optim.hip", "heads.hip", "fp32_validate.hip", "decode.hip", "engine.cpp", "capi_ops.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fno-slp-vectorize", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
# -fno-slp-vectorize: with SLP-packed fp32 math (v_pk_add_f32 / v_pk_mul_f32 with op_sel / neg modifiers on register pairs
# assembled by v_mov) hipcc 7.2 produced an ln_bwd_kernel whose dz output is wrong in a few elements per launch (lanes 48-63,
# one of the four values of a lane) whenever another kernel shares the GPU -- the source of run-to-run gradient
# differences with the weight gradients on a second stream (tools/ln_bwd_contention.py: 20 of 20 contended runs differ at
# -O2 / -O3, also with the SDWA peephole or early if-conversion off or the division replaced; 0 of 20 at -O1 and at -O3
# without the SLP vectoriser).  Explicit two-wide vector code (kmb_f32x2) is not affected.  DESIGN.md section 5.
FILE_FLAGS = {}
# the gradient exchange (kmb_allreduce_grads, csrc/engine.cpp) calls RCCL directly
RCCL_LINK = ["-L/opt/rocm/lib", "-lrccl"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "kmbart.h"))
    jobs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src + ".o")
        deps = [s] + headers + ([os.path.join(CSRC, "gemm.hip")] if src in ("gemm_pair.hip", "gemm_lean.hip") else [])   # they #include gemm.hip
        if force or _stale(o, deps):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = ["hipcc", "-x", "hip"] + FLAGS + FILE_FLAGS.get(os.path.basename(s), []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (s, r.stderr[-4000:]))
        return s

    with ThreadPoolExecutor(max_workers=4) as ex:
        for s in ex.map(cc, jobs):
            if verbose:
                print("[build] compiled", os.path.basename(s))
    objs = [os.path.join(objdir, src + ".o") for src in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + RCCL_LINK
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
        if verbose:
            print("[build] linked", LIB)
    return LIB


EXPERIMENTS = os.path.join(os.path.dirname(HERE), "tools", "experiments")   # measured-and-lost kernels: never in the product library


def build_variant(name, defines, sources=("gemm.hip",), extra_sources=()):
    """Experiment build: lib/libkmbart_hip_<name>.so with extra -D defines on the given sources (the other objects are
    the product build's) plus `extra_sources` from tools/experiments/.  Select it with KMB_LIB_PATH.  Never shipped (built
    on demand, on the box that uses it): diagnostics and A/B measurements only."""
    build(verbose=False)
    objdir = os.path.join(LIBDIR, "obj")
    objs = []
    for src in extra_sources:
        o = os.path.join(objdir, "%s.%s.o" % (src, name))
        cmd = ["hipcc", "-x", "hip"] + FLAGS + ["-I" + CSRC] + ["-D" + d for d in defines] + ["-c", os.path.join(EXPERIMENTS, src), "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        objs.append(o)
    for src in SOURCES:
        o = os.path.join(objdir, src + ".o")
        if src in sources:
            o = os.path.join(objdir, "%s.%s.o" % (src, name))
            cmd = ["hipcc", "-x", "hip"] + FLAGS + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", o]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr[-4000:]))
        objs.append(o)
    lib = os.path.join(LIBDIR, "libkmbart_hip_%s.so" % name)
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + RCCL_LINK, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python build.py --variant plain KMB_PLAIN_STORES | --variant diag KMB_DIAG
        i = sys.argv.index("--variant")
        defs = sys.argv[i + 2:]
        # KMB_DIAG (csrc/diag.h) switches the A/B environment knobs and ablation bits on in every file that has them
        srcs = ("gemm.hip", "gemm_lean.hip", "engine.cpp", "attention.hip", "optim.hip") if "KMB_DIAG" in defs else ("gemm.hip",)
        extra = ()
        if sys.argv[i + 1].startswith("rolesplit") or any(d.startswith("KMB_RS_") for d in defs):
            # the role-split GEMM (variant 10, tools/experiments/gemm_rolesplit.hip): `--variant rolesplit` (+ KMB_RS_NOEPI ...
            # for its timing-only builds); KMB_GEMM_VARIANT=10 KMB_LIB_PATH=lib/libkmbart_hip_rolesplit.so selects it
            defs = list(defs) + ["KMB_WITH_ROLESPLIT"]
            srcs, extra = ("gemm.hip",), ("gemm_rolesplit.hip",)
        if sys.argv[i + 1].startswith("resident"):
            # the resident decoder-layers kernel (tools/experiments/decode_resident.hip; measured 17 % slower than the six-launch blocks):
            # `--variant resident`; KMB_GEN_FUSED=2 KMB_LIB_PATH=lib/libkmbart_hip_resident.so selects it (tools/experiments/test_decode_resident.py)
            defs = list(defs) + ["KMB_WITH_RESIDENT_DECODE"]
            srcs, extra = ("engine.cpp",), ("decode_resident.hip",)
        if any(d.startswith("KMB_DEC_") for d in defs):   # decode-block A/B builds
            srcs = ("decode.hip",)
        if any(d.startswith("KMB_PR_") for d in defs):   # ... of the two-workgroups-per-CU kernel
            srcs = ("gemm_pair.hip",)
        print(build_variant(sys.argv[i + 1], defs, srcs, extra))
    else:
        build(force="--force" in sys.argv)
