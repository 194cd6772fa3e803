"""K-loop audit of the compiled GEMM kernels (no GPU): per kernel, the innermost loop that holds >= 32 MFMAs -- one 64-deep K step -- with its
instruction count, MFMAs, LDS-DMA pieces, plain / transposing LDS reads, `v_accvgpr` moves (accumulators shuffled between the two register
files), `s_nop`s, scratch accesses and the vector-memory waits it contains.  One counted wait per step is the design; every extra `vmcnt(0)` is a
drained prefetch (profiles/r05_gemm_transposing_reads_asm.md: hipcc puts one in front of the intrinsic transposing read).

    python tools/gemm_kloop_audit.py                      # compiles csrc/gemm.hip, gemm_lean.hip, gemm_pair.hip with the product flags
    python tools/gemm_kloop_audit.py --sig v7d            # also prints the loop's instruction signature for kernels whose name contains "v7d"
    python tools/gemm_kloop_audit.py -DKMB_TR_BUILTIN     # extra -D flags are passed to hipcc
Output: a markdown table on stdout (profiles/r06_gemm_kloop_audit.md is this tool's output)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
from build import CSRC, FLAGS  # noqa: E402


def compile_to_isa(src, defines, outdir):
    out = os.path.join(outdir, os.path.basename(src) + ".s")
    cmd = ["hipcc", "-x", "hip"] + FLAGS + defines + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit("hipcc failed on %s:\n%s" % (src, r.stderr[-3000:]))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    out = r.stdout.splitlines() if r.returncode == 0 else names
    return dict(zip(names, out))


def kernels(s):
    for m in re.finditer(r"^(_Z\S*gemm\S*): ", s, re.M):
        name = m.group(1)
        i = m.start()
        j = s.find(".Lfunc_end", i)
        yield name, s[i:j].splitlines()


def k_loop(b):
    """the smallest backward-branch loop with >= 32 MFMAs in its body"""
    labels = {}
    for n, l in enumerate(b):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = n
    best = None
    for n, l in enumerate(b):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            a = labels[m.group(1)]
            nm = sum("v_mfma" in x for x in b[a:n + 1])
            if nm >= 32 and (best is None or (n - a) < (best[1] - best[0])):
                best = (a, n)
    return best


def signature(seg):
    out = []
    for l in seg:
        t = l.strip()
        c = None
        if t.startswith("v_mfma"): c = "M"
        elif t.startswith("ds_read_b64_tr"): c = "T"
        elif t.startswith("ds_read"): c = "D"
        elif t.startswith("global_load_lds"): c = "V"
        elif t.startswith("scratch"): c = "S"
        elif t.startswith("global_") or t.startswith("buffer_"): c = "G"
        elif t.startswith("v_accvgpr"): c = "a"
        elif t.startswith("s_waitcnt") and "vmcnt" in t: c = "[W%s]" % re.search(r"vmcnt\((\d+)\)", t).group(1)
        elif t.startswith("s_waitcnt") and "lgkmcnt(0)" in t: c = "[L0]"
        elif t.startswith("s_barrier"): c = "|B|"
        if c:
            out.append(c)
    r = "".join(out)
    return re.sub(r"([A-Za-z])\1{3,}", lambda m: "%s*%d " % (m.group(1), len(m.group(0))), r)


def audit_file(path, src="gemm.hip", sig_for=()):
    """rows (kernel, loop instructions, MFMAs, LDS-DMA pieces, plain LDS reads, transposing LDS reads, v_accvgpr moves, s_nop, scratch, waits) of
    every GEMM kernel in the device assembly `path` (tests/test_cabi_cpu.py asserts on these for the product build)"""
    s = open(path).read()
    ks = list(kernels(s))
    dm = demangle([k for k, _ in ks])
    rows = []
    for name, b in ks:
        if src != "gemm.hip" and ("lean" not in name and "pair" not in name):
            continue   # gemm_lean.hip / gemm_pair.hip include gemm.hip's device code: only their own kernels
        lp = k_loop(b)
        if lp is None:
            continue
        seg = [l.strip() for l in b[lp[0]:lp[1] + 1]]
        ins = [l for l in seg if l and not l.startswith(";") and not l.startswith(".")]
        waits = " ".join("vmcnt(%s)" % re.search(r"vmcnt\((\d+)\)", l).group(1) for l in ins if l.startswith("s_waitcnt") and "vmcnt" in l)
        pretty = dm.get(name, name).replace("(anonymous namespace)::", "")
        pretty = re.sub(r"\(KmbGemm.*$", "", re.sub(r"^void ", "", pretty))
        rows.append((pretty, len(ins), sum(l.startswith("v_mfma") for l in ins), sum(l.startswith("global_load_lds") for l in ins),
                     sum(l.startswith("ds_read") and "_tr_" not in l for l in ins), sum(l.startswith("ds_read_b64_tr") for l in ins),
                     sum(l.startswith("v_accvgpr") for l in ins), sum(l.startswith("s_nop") for l in ins),
                     sum(l.startswith("scratch_") for l in ins), waits or "-"))
        if any(p in pretty for p in sig_for):
            print("<!-- %s: %s -->" % (pretty, signature(seg)))
    return rows


def main():
    defines = [a for a in sys.argv[1:] if a.startswith("-D")]
    sig_for = [sys.argv[i + 1] for i, a in enumerate(sys.argv[:-1]) if a == "--sig"]
    rows = []
    with tempfile.TemporaryDirectory() as td:
        for src in ("gemm.hip", "gemm_lean.hip", "gemm_pair.hip"):
            rows += audit_file(compile_to_isa(src, defines, td), src, sig_for)
    print("kernel | loop instructions | MFMAs | LDS-DMA pieces | plain / transposing LDS reads | v_accvgpr moves | s_nop | scratch | vector-memory waits")
    print("|---|---|---|---|---|---|---|---|---|")
    for r in rows:
        print("| `%s` | %d | %d | %d | %d / %d | %d | %d | %d | %s |" % r)


if __name__ == "__main__":
    main()
