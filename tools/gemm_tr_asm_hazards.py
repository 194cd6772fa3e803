"""Static check for the inline-asm transposing LDS reads of the GEMM kernels (csrc/gemm.hip, kmb_tr_read_asm): hipcc does not know that their
destination registers arrive later, so between such a read and the next `s_waitcnt ... lgkmcnt(0)` NO instruction may name those registers (not as a
source: stale data; not as a destination: the late LDS return would overwrite the new value).  Walks the control-flow graph of every kernel in the
device assembly (fixpoint over basic blocks, pending sets merged by union) and lists violations.

    hipcc -x hip --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -ffp-contract=off -c km-bart_amd/csrc/gemm.hip -o /tmp/g.o -save-temps=obj
    python tools/gemm_tr_asm_hazards.py /tmp/gemm-hip-amdgcn-amd-amdhsa-gfx950.s        (same for gemm_lean.hip)
Only inline-asm reads (between ;;#ASMSTART and ;;#ASMEND) are tracked: the intrinsic's reads (the 128- / 192-wide tiles, the four-stage and the register-staged kernels,
gemm_lean.hip) are covered by the compiler's own waits."""
import re
import sys

src = open(sys.argv[1]).read()
starts_at = {m.group(1): m.start() for m in re.finditer(r"^(_ZN\S*gemm\S*): ", src, re.M)}
names = list(starts_at)
SKIP = ()


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


total = 0
for name in names:
    if any(k in name for k in SKIP):
        continue
    i = starts_at[name]
    j = src.index(".Lfunc_end", i)
    lines = [l.strip() for l in src[i:j].splitlines()]
    # only the reads hipcc did not write itself: inline asm is bracketed by ";;#ASMSTART" / ";;#ASMEND" (the intrinsic's reads are covered by the compiler's own waits)
    in_app = False
    for n, l in enumerate(lines):
        if l.startswith(";;#ASMSTART"):
            in_app = True
        elif l.startswith(";;#ASMEND"):
            in_app = False
        elif in_app and l.startswith("ds_read_b64_tr_b16"):
            lines[n] = "ASM_" + l
    ins = [(n, l) for n, l in enumerate(lines) if l and not l.startswith(";") and not l.startswith(".") or re.match(r"\.LBB\d+_\d+:", l)]
    if not any(l.startswith("ASM_ds_read_b64_tr_b16") for _, l in ins):
        continue
    # basic blocks
    label_at = {}
    for k, (n, l) in enumerate(ins):
        m = re.match(r"(\.LBB\d+_\d+):", l)
        if m:
            label_at[m.group(1)] = k
    starts = {0} | set(label_at.values())
    for k, (n, l) in enumerate(ins):
        if re.match(r"s_c?branch", l) and k + 1 < len(ins):
            starts.add(k + 1)
    starts = sorted(starts)
    blk_of = {}
    blocks = []
    for b, s0 in enumerate(starts):
        e0 = starts[b + 1] if b + 1 < len(starts) else len(ins)
        blocks.append((s0, e0))
        blk_of[s0] = b
    succ = []
    for (s0, e0) in blocks:
        last = ins[e0 - 1][1]
        out = []
        m = re.match(r"s_(c?)branch\S*\s+(\.LBB\d+_\d+)", last)
        if m:
            out.append(blk_of[label_at[m.group(2)]])
            if m.group(1) == "c" and e0 < len(ins):
                out.append(blk_of[e0])
        elif e0 < len(ins):
            out.append(blk_of[e0])
        succ.append(out)
    IN = [dict() for _ in blocks]
    bad = {}
    work = [0]
    seen_state = {}
    while work:
        b = work.pop()
        pend = dict(IN[b])
        s0, e0 = blocks[b]
        for k in range(s0, e0):
            n, l = ins[k]
            if l.startswith("s_waitcnt") and "lgkmcnt(0)" in l:
                pend = {}
                continue
            toks = re.findall(r"v\[\d+:\d+\]|\bv\d+\b", l)
            if l.startswith("ASM_ds_read_b64_tr_b16"):
                for tk in toks[1:]:
                    for r in regs(tk):
                        if r in pend:
                            bad[(n, r)] = (l, pend[r])
                for r in regs(toks[0]):
                    pend[r] = n
                continue
            for tk in toks:
                for r in regs(tk):
                    if r in pend:
                        bad[(n, r)] = (l, pend[r])
        for sb in succ[b]:
            merged = dict(IN[sb])
            changed = False
            for r, o in pend.items():
                if r not in merged:
                    merged[r] = o
                    changed = True
            if changed or sb not in seen_state:
                seen_state[sb] = True
                IN[sb] = merged
                work.append(sb)
    ntr = sum(l.startswith("ASM_ds_read_b64_tr_b16") for _, l in ins)
    print(f"{name[:78]:78s} transposing reads {ntr:4d}  violations {len(bad)}")
    for (n, r), (l, o) in sorted(bad.items())[:6]:
        print(f"      line {n}: {l[:80]}   (v{r} requested at line {o})")
    total += len(bad)
print("total violations", total)
sys.exit(1 if total else 0)
