"""CPU: the C-ABI library loads without a GPU and exports every symbol include/kmbart.h declares; the
parameter census behind it matches the reference's state-dict names and sizes."""
import ctypes as C
import json
import os
import re

import pytest

from kmbart import _lib
from kmbart._lib import KmbConfig, check
from oracle import kmbart_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

VCG_BASE = dict(vocab_size=50320, d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12,
                decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072, max_position_embeddings=1024,
                extra_pos_embeddings=2, image_feature_size=2052, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                img_feat_id=50273, cls_token_id=50276, scale_embedding=0, dropout=0.1, attention_dropout=0.0,
                activation_dropout=0.0, layer_norm_eps=1e-5)


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "kmbart.h")).read()
    declared = set(re.findall(r"\b(kmb_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 40
    for name in sorted(declared):
        assert hasattr(lib, name), "libkmbart_hip.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    assert lib.kmb_version() >= 1


def test_parameter_census_matches_reference_names(lib):
    h = C.c_void_p()
    cfg = KmbConfig(**VCG_BASE)
    check(lib.kmb_create(C.byref(cfg), C.byref(h)))
    try:
        assert lib.kmb_arena_elems(h) == 141039360  # SURVEY.md section 2.4 census
        ocfg = O.OracleConfig.from_dict({k: v for k, v in VCG_BASE.items()})
        want = {n: O.param_shape(ocfg, n) for n in O.param_names(ocfg)}
        name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
        got, spans = {}, []
        for i in range(lib.kmb_param_count(h)):
            check(lib.kmb_param_info(h, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols)))
            got[name.value.decode()] = (rows.value, cols.value)
            spans.append((off.value, rows.value * cols.value))
        assert set(got) == set(want)
        for n, shp in want.items():
            r, c = got[n]
            assert (r * c) == int(__import__("numpy").prod(shp)), n
            assert (c,) == tuple(shp) if len(shp) == 1 else (r, c) == tuple(shp), n
        spans.sort()
        for (o1, n1), (o2, _) in zip(spans, spans[1:]):
            assert o1 + n1 <= o2 and o1 % 64 == 0
        # q|k|v are adjacent so that one GEMM serves the fused projection
        idx = {n: i for i, n in enumerate(sorted(got, key=lambda n: dict(zip(got, range(len(got))))[n]))}
        assert idx  # names iterate in arena order
        total = 0
        o, n = C.c_int64(), C.c_int64()
        for i in range(lib.kmb_bucket_count(h)):
            check(lib.kmb_bucket_range(h, i, C.byref(o), C.byref(n)))
            total += n.value
        assert total == 141039360 and lib.kmb_bucket_count(h) == 6 + 6 + 3
        assert lib.kmb_logits_ld(h) == 50432
        ws = lib.kmb_workspace_bytes(h, 2, 64, 32, 72)
        assert 10 << 20 < ws < 200 << 20
    finally:
        lib.kmb_destroy(h)


def test_unsupported_configs_fail_loudly(lib):
    h = C.c_void_p()
    bad = dict(VCG_BASE, d_model=1024)  # head_dim != 64
    assert lib.kmb_create(C.byref(KmbConfig(**bad)), C.byref(h)) != 0
    assert b"head_dim" in lib.kmb_last_error()
    bad = dict(VCG_BASE, attention_dropout=0.1)
    assert lib.kmb_create(C.byref(KmbConfig(**bad)), C.byref(h)) != 0


def test_unbound_handle_refuses_to_run(lib):
    h = C.c_void_p()
    check(lib.kmb_create(C.byref(KmbConfig(**VCG_BASE)), C.byref(h)))
    try:
        b = _lib.KmbBatch(B=1, S=8, T=4)
        assert lib.kmb_forward(h, C.byref(b), 0, 0, None, None, None, None) != 0
        assert b"not bound" in lib.kmb_last_error()
        assert lib.kmb_backward(h, 1.0, None) != 0
    finally:
        lib.kmb_destroy(h)


def test_exchange_without_a_communicator_fails_loudly(lib):
    """kmb_allreduce_grads / kmb_comm_wait / kmb_comm_broadcast_params before kmb_comm_init: an error, not a silent no-op
    (a data-parallel job whose gradients are not exchanged would train N different models)."""
    h = C.c_void_p()
    check(lib.kmb_create(C.byref(KmbConfig(**VCG_BASE)), C.byref(h)))
    try:
        rank, world = C.c_int32(-1), C.c_int32(-1)
        check(lib.kmb_comm_info(h, C.byref(rank), C.byref(world)))
        assert world.value == 0
        assert lib.kmb_allreduce_grads(h, None, None) != 0 and b"communicator" in lib.kmb_last_error()
        assert lib.kmb_comm_wait(h, None) != 0
        assert lib.kmb_comm_broadcast_params(h, 0, None) != 0
        assert lib.kmb_comm_gather_moments(h, None) != 0
        assert lib.kmb_comm_pieces(h, 0) >= lib.kmb_bucket_count(h)       # every bucket is at least one piece
        assert lib.kmb_comm_pieces(h, 1 << 20) > lib.kmb_comm_pieces(h, 0)  # 4 MB pieces: more of them
        assert lib.kmb_comm_destroy(h) == 0                                # idempotent
        assert lib.kmb_gen_encoder_states(h, None, None) != 0 and lib.kmb_hidden_state(h, 0, 0, None, None) != 0
    finally:
        lib.kmb_destroy(h)


def test_no_cpu_fallback():
    import torch
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = MultiModalBartConfig.from_dict(dict(vocab_size=512, d_model=128, encoder_layers=1, decoder_layers=1,
                                              encoder_attention_heads=2, decoder_attention_heads=2, encoder_ffn_dim=128,
                                              decoder_ffn_dim=128, max_position_embeddings=32, img_feat_id=488,
                                              cls_token_id=491))
    model = MultiModalBartForConditionalGeneration(cfg)
    with pytest.raises(RuntimeError, match="MI355X"):
        model(input_ids=torch.zeros((1, 4), dtype=torch.long), image_features=[torch.empty(0)])
    with pytest.raises(RuntimeError):
        model.to("cuda:0")
    src_files = []
    for base, _, files in os.walk(os.path.join(ROOT, "km-bart_amd")):
        src_files += [os.path.join(base, f) for f in files if f.endswith(".py")]
    for f in src_files + [os.path.join(ROOT, "bench.py")]:
        if not os.path.exists(f):
            continue
        text = open(f).read()
        if f.endswith("bench.py"):
            continue  # bench.py may use the oracle for its cpu_baseline leg only
        assert "oracle" not in text.replace("# oracle", ""), "product file %s mentions the oracle" % f


def _compile_to_isa(src_name, out, defines=()):
    """hipcc -S with the PRODUCT flags (imported from km-bart_amd/build.py, not copied) -- no GPU needed"""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
    from build import CSRC, FLAGS
    cmd = ["hipcc", "-x", "hip"] + FLAGS + list(defines) + ["-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                                                             os.path.join(CSRC, src_name)]
    subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
    return out


def test_l2_touches_have_no_register_destination(tmp_path):
    """The persistent GEMMs' L2 touch is a load whose result nobody reads and which stays in flight across a K step's counted wait.  Rounds 2-5 gave
    it a register (a fresh one, one register web, v255) and each form broke once the allocator reused that register; since round 6 EVERY touch -- also
    csrc/gemm_lean.hip's, whose cross-tile touches used to be in flight during the epilogue -- is a 4-byte LDS-DMA into the issuing wave's staging
    image (KMB_L2_TOUCH).  A property of the compiled code: no inline-asm load with a register destination in any GEMM kernel, touches present in the
    lean kernels, and nothing spilled to scratch in them with the intrinsic reads."""
    for src, min_kernels in (("gemm_lean.hip", 12), ("gemm_pair.hip", 6)):
        text = open(_compile_to_isa(src, str(tmp_path / (src + ".s")))).read()
        kernels = 0
        for m in re.finditer(r"^(_Z\w*gemm_kernel_(?:lean|pair)\w*):", text, re.M):
            body = text[m.start():text.index(".Lfunc_end", m.start())]
            kernels += 1
            in_asm = False
            for line in body.splitlines():
                t = line.strip()
                if t.startswith(";;#ASMSTART"):
                    in_asm = True
                elif t.startswith(";;#ASMEND"):
                    in_asm = False
                elif in_asm:
                    # (an LDS-DMA's first operand is its address offset, not a destination)
                    assert "_lds_" in t or not re.match(r"(global|buffer|flat)_load_\w+\s+v", t), \
                        "%s: inline-asm load into a register: %s" % (m.group(1), t)
            assert "global_load_lds_dword " in body, "%s has no L2 touch" % m.group(1)
        assert kernels >= min_kernels, (src, kernels)


def test_gemm_asm_transposing_reads_are_waited_for(tmp_path):
    """csrc/gemm.hip and csrc/gemm_lean.hip read their token-major operands with `ds_read_b64_tr_b16` as INLINE ASM in every kernel since round 6
    (kmb_tr_read_asm / kmb_tr_read_asm_off: the intrinsic makes hipcc wait with vmcnt(0) behind every LDS-DMA issue, i.e. for the stage requested a moment ago).
    The compiler then does not know that those registers arrive later: between such a read and the next `s_waitcnt lgkmcnt(0)` no instruction may
    name them (KMB_TR_SYNC at the top of every sub-phase).  A property of the COMPILED code: compile to ISA (no GPU) and walk every kernel's
    control-flow graph (tools/gemm_tr_asm_hazards.py)."""
    import subprocess
    import sys
    for src, least in (("gemm.hip", 17), ("gemm_lean.hip", 6)):   # round 6: every kernel with a token-major operand (KMB_TR_ALL), gemm_lean.hip's six data-gradient kernels
        out = _compile_to_isa(src, str(tmp_path / (src + ".s")))
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_tr_asm_hazards.py"), out], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout[-3000:]
        checked = [l for l in r.stdout.splitlines() if "transposing reads" in l]
        assert len(checked) >= least, "%s: expected the kernels with inline-asm reads, found %d:\n%s" % (src, len(checked), r.stdout[-2000:])
        assert "total violations 0" in r.stdout
        # the same compiled code, K loop by K loop (tools/gemm_kloop_audit.py; round 6 removed all of these, profiles/r06_gemm_kloop_audit.md):
        # no accumulator shuffled between the register files, nothing spilled to scratch inside a K step
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from gemm_kloop_audit import audit_file
        rows = audit_file(out, src)
        assert len(rows) >= (29 if src == "gemm.hip" else 12), (src, len(rows))
        for row in rows:
            name, accvgpr, scratch = row[0], row[6], row[8]
            assert accvgpr == 0, "%s: %d v_accvgpr moves inside its K step" % (name, accvgpr)
            allowed = 1 if name.startswith("gemm_kernel_v11<true, false, 256, 8>") else 0   # (one reload at the TILE switch, in the loop's tail block)
            assert scratch <= allowed, "%s: %d scratch accesses inside its K step" % (name, scratch)


def test_tr_asm_hazard_checker_sees_a_violation(tmp_path):
    """The checker itself: a synthetic kernel in which an inline-asm transposing read's register is used before the next full LDS wait (directly,
    and along a branch that skips the wait) must be flagged; the same kernel with the wait in front of the use must pass."""
    import subprocess
    import sys
    tool = os.path.join(ROOT, "tools", "gemm_tr_asm_hazards.py")
    head = "_ZN4test11gemm_kernelEv: ; @x\n"
    read = "\t;;#ASMSTART\n\tds_read_b64_tr_b16 v[10:11], v3\n\t;;#ASMEND\n"
    tail = "\ts_endpgm\n.Lfunc_end0:\n"
    cases = {
        "use_before_wait": (head + read + "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[20:23], a[0:3]\n\ts_waitcnt lgkmcnt(0)\n" + tail, 1),
        "overwritten_before_wait": (head + read + "\tv_mov_b32_e32 v11, 0\n\ts_waitcnt lgkmcnt(0)\n" + tail, 1),
        "branch_around_the_wait": (head + read + "\ts_cbranch_scc1 .LBB0_2\n\ts_waitcnt lgkmcnt(0)\n.LBB0_2:\n\tv_add_u32_e32 v1, v10, v2\n" + tail, 1),
        "partial_wait_is_not_enough": (head + read + "\ts_waitcnt lgkmcnt(1)\n\tv_add_u32_e32 v1, v10, v2\n" + tail, 1),
        "waited": (head + read + "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[4:7], v[20:23], a[0:3]\n\ts_waitcnt vmcnt(0) lgkmcnt(0)\n"
                   "\tv_mfma_f32_16x16x32_bf16 a[0:3], v[10:13], v[20:23], a[0:3]\n" + tail, 0),
        "intrinsic_reads_are_the_compilers": (head + "\tds_read_b64_tr_b16 v[10:11], v3\n\tv_add_u32_e32 v1, v10, v2\n" + tail, 0),
    }
    for name, (text, want) in cases.items():
        p = tmp_path / (name + ".s")
        p.write_text(text)
        r = subprocess.run([sys.executable, tool, str(p)], capture_output=True, text=True)
        assert r.returncode == want, "%s: rc %d, expected %d\n%s" % (name, r.returncode, want, r.stdout + r.stderr)
