"""Where the register spills of the GEMM kernels execute.  Compiles csrc/gemm.hip to gfx950 assembly with line tables, and for
every kernel that spills prints its static spill count, the scratch instructions per SOURCE FUNCTION (by line range) and how
many of them sit inside MFMA-dense code (>= 8 MFMAs within 40 instructions: the K loop).  No GPU needed.

    python tools/gemm_spill_map.py > profiles/rNN_gemm_spill_map.txt
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "km-bart_amd", "csrc")
SRC = os.path.join(CSRC, "gemm.hip")


def source_functions():
    """(first line, name) of every function / kernel definition in gemm.hip, by a light scan"""
    out = []
    pat = re.compile(r"^(?:__device__ __forceinline__|__global__|static|inline)?.*?\b([A-Za-z_0-9]+)\s*\((?:const KmbGemm|const bf16_t|char\*|const char\*|uint32_t|int |const float|f32x4|float\*)")
    for i, l in enumerate(open(SRC), 1):
        if l.startswith(("__device__", "__global__")) or (l.startswith("template") is False and re.match(r"^[a-z].*\)\s*\{\s*$", l)):
            m = pat.match(l)
            if m:
                out.append((i, m.group(1)))
    return out


def main():
    asm = os.path.join(tempfile.gettempdir(), "kmb_gemm_lines.s")
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-ffp-contract=off", "-gline-tables-only",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, "--cuda-device-only", "-S", SRC, "-o", asm]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    lines = open(asm).read().split("\n")
    funcs = source_functions()

    def fn_of(line):
        name = "?"
        for first, n in funcs:
            if first <= line:
                name = n
        return name

    meta = {}
    cur = None
    for l in lines:
        m = re.match(r"\s*\.name:\s+(\S+)", l)
        if m:
            cur = m.group(1)
        m = re.match(r"\s*\.vgpr_spill_count:\s+(\d+)", l)
        if m and cur:
            meta[cur] = int(m.group(1))
    starts = [(i, m.group(1)) for i, l in enumerate(lines) for m in [re.match(r"^(_ZN[A-Za-z0-9_]+):", l)] if m]
    print("kernel | static vgpr spills | scratch instructions | of them in MFMA-dense code | by source function (line of gemm.hip)")
    for idx, (a, name) in enumerate(starts):
        if meta.get(name, 0) == 0:
            continue
        body = []
        for l in lines[a:]:
            if l.startswith(".Lfunc_end"):
                break
            body.append(l)
        mf = [j for j, l in enumerate(body) if "v_mfma" in l]
        cnt = collections.Counter()
        hot = 0
        cur_line = 0
        n_sc = 0
        for j, l in enumerate(body):
            m = re.match(r"\s*\.loc\s+\d+\s+(\d+)", l)
            if m:
                cur_line = int(m.group(1))
            if "scratch_" in l:
                n_sc += 1
                cnt[fn_of(cur_line) if cur_line else "prologue / no line"] += 1
                if sum(1 for k in mf if abs(k - j) < 40) >= 8:
                    hot += 1
        short = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
        print("%s | %d | %d | %d | %s" % (short, meta[name], n_sc, hot, ", ".join("%s %d" % kv for kv in cnt.most_common())))


if __name__ == "__main__":
    sys.exit(main())
