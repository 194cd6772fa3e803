"""KM-BART vcg_base training throughput on MI355X (BASELINE.json metric: training tokens/sec, whole node).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B_PER_GPU]

A step = forward (dropout 0.1 as config/vcg_base.json) + backward + gradient all-reduce (N > 1) + fused
AdamW on one synthetic VCG batch per GPU (36 regions x 2052-d, 64 encoder tokens, 32 decoder tokens;
SURVEY.md section 8d), inputs resident in HBM.  For N > 1 the driver launches one rank per GPU with
torch.distributed.run; RCCL ("nccl") carries the all-reduce.  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "km-bart_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

VCG_BASE = dict(  # reference config/vcg_base.json
    activation_dropout=0.0, activation_function="gelu", attention_dropout=0.0, d_model=768,
    decoder_attention_heads=12, decoder_ffn_dim=3072, decoder_layers=6, dropout=0.1, encoder_attention_heads=12,
    encoder_ffn_dim=3072, encoder_layers=6, init_std=0.02, max_position_embeddings=1024, vocab_size=50320,
    cls_token_id=50276, img_feat_id=50273, pad_token_id=1, bos_token_id=0, eos_token_id=2)
S_ENC, T_DEC, REGIONS = 64, 32, 36
GFLOP_PER_TOKEN = 0.3822          # 36.69 GFLOP per training sample / 96 tokens
PEAK_BF16_TFLOPS = 2500.0         # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)


def gemm_sources_sha16():
    """Fingerprint of the GEMM kernel sources in this tree (csrc/gemm*.hip + the headers they include).  A committed PMC traffic file
    (profiles/r*_pmc_traffic.json) records the fingerprint of the code it was measured on; `roofline.traffic` is only quoted from a
    file whose fingerprint is this tree's (the r05 line carried the traffic of the kernels BEFORE the round's last change)."""
    import glob
    import hashlib
    csrc = os.path.join(ROOT, "km-bart_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(csrc, "gemm*.hip"))) + [os.path.join(csrc, n) for n in ("common.h", "kernels.h", "diag.h")]:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def usable_cores():
    """Cores this process may run on (cgroup / affinity aware; os.cpu_count() reports the whole host)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:  # cgroup v2 CPU quota
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 64))


def cpu_baseline(n_steps=8, warmup=2, budget_s=30.0):
    """The oracle (CPU restatement of the reference path, --cpu semantics of vcg_train.py:62-64) timed on
    this box's host cores: b=2, fp32, dropout 0.1, HF-AdamW.  Bounded sample of the same workload."""
    from oracle import kmbart_oracle as O
    from src.data.synthetic import make_batch
    cores = usable_cores()
    torch.set_num_threads(cores)
    cfg = O.OracleConfig.from_dict(VCG_BASE)
    model = O.OracleModel(cfg, seed=0).train()
    opt = O.HFAdamW(model.parameters(), lr=1e-5)
    b = make_batch(2, seed=1234)
    times = []
    t_begin = time.perf_counter()
    for i in range(warmup + n_steps):
        if times and time.perf_counter() - t_begin > budget_s:
            break  # bounded sample: keep the default bench run within minutes on any host
        t0 = time.perf_counter()
        loss = model(b["input_ids"], b["image_features"], b["attention_mask"],
                     decoder_input_ids=b["decoder_input_ids"], decoder_attention_mask=b["decoder_attention_mask"],
                     labels=b["labels"])[0]
        loss.item()
        opt.zero_grad()
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        if i >= warmup:
            times.append(dt)
        elif dt > budget_s / 3:  # pathologically slow host: one untimed + one timed step is all we spend
            warmup, n_steps = i + 1, 1
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(2 * (S_ENC + T_DEC) / med, 2), "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": "oracle (pure PyTorch fp32) b=2, S=64, T=32, 36 regions, %d timed steps, median %.3f s/step"
                      % (len(times), med)}


def loss_parity(model, dev):
    """CE-loss delta vs the fp32 oracle on BASELINE config 1 (b=2, dropout off, identical weights)."""
    from oracle import kmbart_oracle as O
    from src.data.synthetic import make_batch
    cfg = O.OracleConfig.from_dict(dict(VCG_BASE, dropout=0.0))
    sd = model.state_dict()
    b = make_batch(2, seed=1234)
    with torch.no_grad():
        ref, _, _ = O.forward(sd, cfg, b["input_ids"], b["image_features"], b["attention_mask"],
                              b["decoder_input_ids"], b["decoder_attention_mask"], b["labels"])
        was = model.training
        model.eval()
        got = model(input_ids=b["input_ids"].to(dev), image_features=[f.to(dev) for f in b["image_features"]],
                    attention_mask=b["attention_mask"].to(dev), decoder_input_ids=b["decoder_input_ids"].to(dev),
                    decoder_attention_mask=b["decoder_attention_mask"].to(dev), labels=b["labels"].to(dev))[0]
        model.train(was)
    return abs(float(got) - float(ref)) / abs(float(ref)), float(got), float(ref)


def gemm_roofline(model, step, per_gpu_batch, n_prof=3):
    """The dominant kernel = the bf16 MFMA GEMM: every launch of `n_prof` steps timed with HIP events on its own stream
    (weight gradients moved to the caller's stream for this leg, so launches do not overlap each other)."""
    import glob
    import tempfile
    from kmbart import _lib
    lib = _lib.load()
    agg = {}
    lib.kmb_set_side_stream(model._engine.h, 0)  # kernels timed one at a time, not overlapped with each other
    step()
    alg_bytes, alg_n = 0.0, 0
    ce_us, ce_fl, last_us, last_fl = 0.0, 0.0, 0.0, 0.0   # the launch whose epilogue carries the cross-entropy (act 5), last profiled step
    for i in range(n_prof):
        lib.kmb_profile_gemm(1)
        step()
        torch.cuda.synchronize()
        for variant, name in ((3, "fwd"), (2, "dgrad"), (0, "wgrad")):
            n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
            _lib.check(lib.kmb_profile_read(variant, C.byref(n), C.byref(ms), C.byref(fl)))
            a = agg.setdefault(name, [0, 0.0, 0.0])
            a[0] += n.value
            a[1] += ms.value
            a[2] += fl.value
        if i == n_prof - 1:   # minimal bytes of the same launches: A and B (and a fused side operand) read once, C written once
            dump = os.path.join(tempfile.gettempdir(), "kmb_gemm_launches_%d.txt" % os.getpid())
            _lib.check(lib.kmb_profile_dump(dump.encode()))
            for line in open(dump):
                v, M, N, K, sp, act, us, *rest = line.split()
                v, M, N, K, act = int(v), int(M), int(N), int(K), int(act)
                out_b = 4 if v == 0 else 2                               # weight gradients are fp32, everything else bf16
                extra = M * N * 2 if act in (1, 2) else 0                # GeLU' written / read beside the output
                if rest and int(rest[0]):
                    extra += M * N * 2                                   # the residual / other gradient term the epilogue adds
                alg_bytes += 2.0 * (M * K + N * K) + out_b * M * N + extra
                alg_n += 1.0 / int(rest[1]) if len(rest) > 1 else 1   # a problem of a grouped weight-gradient launch: 1 / n of a launch
                last_us += float(us)
                last_fl += 2.0 * M * N * K
                if act == 5:
                    ce_us += float(us)
                    ce_fl += 2.0 * M * N * K
            os.remove(dump)
        lib.kmb_profile_gemm(0)
    lib.kmb_set_side_stream(model._engine.h, 1)
    traffic, traffic_src, traffic_sha = None, None, None
    # HBM bytes are NOT measured in this run (PMC counters need rocprofv3 around the process): they come from the newest committed
    # rocprofv3 --pmc passes of this workload (tools/r6_profile.sh) -- and only if that file was measured on THIS tree's GEMM sources
    # (gemm_sources_sha16); otherwise traffic is null and traffic_source says which file was refused.
    here = gemm_sources_sha16()
    for pmc in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
        rec = json.load(open(pmc))
        if rec.get("per_gpu_batch") != per_gpu_batch:
            continue
        traffic_sha = rec.get("gemm_sources_sha16")
        if traffic_sha == here:
            traffic = rec["gemm_hbm_bytes_per_launch"]
            traffic_src = "profiles/%s: %s" % (os.path.basename(pmc), rec["method"])
        else:
            traffic_src = ("refused: profiles/%s was measured on GEMM sources %s, this tree's are %s (re-run tools/r6_profile.sh)"
                           % (os.path.basename(pmc), traffic_sha or "unrecorded (before round 6)", here))
        break
    tot_ms = sum(a[1] for a in agg.values())
    tot_fl = sum(a[2] for a in agg.values())
    launches = sum(a[0] for a in agg.values())
    ach = tot_fl / (tot_ms * 1e-3) / 1e12
    return {
        "bound": "mfma", "achieved": round(ach, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
        "frac": round(ach / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_unit": "HBM bytes per GEMM launch",
        "traffic_source": traffic_src, "traffic_measured_in_run": False, "traffic_gemm_sources_sha16": traffic_sha,
        "gemm_sources_sha16": here,
        "algorithmic_bytes_per_launch": round(alg_bytes / max(alg_n, 1)),
        "kernel": "gemm_kernel_v7 / v8 / v11 / v14 (all GEMM launches of a step, timed serially on their stream)",
        "per_gpu_batch": per_gpu_batch,
        "launches_per_step": launches // n_prof, "avg_launch_us": round(tot_ms / launches * 1e3, 2),
        "gemm_ms_per_step": round(tot_ms / n_prof, 3),
        # round 3: the tied head's forward GEMM also computes the cross-entropy's exponentials and row sums in its epilogue (the
        # 1.76 ms softmax kernel of round 2 is gone); its time counts as GEMM time here.  For comparisons across rounds:
        "ce_head_launch": ({"us": round(ce_us, 1), "tflops": round(ce_fl / ce_us * 1e-6, 1),
                            "frac_of_other_launches": round((last_fl - ce_fl) / (last_us - ce_us) * 1e-6 / PEAK_BF16_TFLOPS, 4)}
                           if ce_us > 0 and last_us > ce_us else None),
        "by_variant": {k: {"launches_per_step": a[0] // n_prof, "avg_us": round(a[1] / max(a[0], 1) * 1e3, 2),
                           "tflops": round(a[2] / (a[1] * 1e-3) / 1e12, 1) if a[1] > 0 else None}
                       for k, a in agg.items()},
    }


def bench_batch_check(model, dev, bsz, chunk=64):
    """Correctness gate at the measured size (the GEMM variants, split-K rules and tile orders a batch of `bsz` switches
    on are not the ones a small batch runs): loss and EVERY gradient of one ragged batch of `bsz` samples (eval mode: no
    dropout) against the token-weighted sum over its 64-sample chunks -- mean-token CE, reference src/model/model.py:400-402:
    whole-batch gradient = sum_c (n_c / n) x chunk gradient -- each chunk going through the small-batch kernels.  The
    same comparison with an oracle check of the chunks is tests/test_bench_batch_gpu.py; here it guards the numbers of
    THIS run (a broken kernel is often a fast kernel)."""
    from src.data.synthetic import make_batch
    g = torch.Generator().manual_seed(bsz)
    lab = torch.randint(1, T_DEC + 1, (bsz,), generator=g).tolist()
    b = make_batch(bsz, enc_len=S_ENC, dec_len=T_DEC, num_regions=REGIONS, seed=555 + bsz, label_lens=lab)
    eng = model._engine

    def run(lo, hi, scale):
        feats = [f.to(dev) for f in b["image_features"][lo:hi]]
        loss, _, _ = eng.forward(b["input_ids"][lo:hi].to(dev), feats, b["attention_mask"][lo:hi].to(dev),
                                 b["decoder_input_ids"][lo:hi].to(dev), b["decoder_attention_mask"][lo:hi].to(dev),
                                 b["labels"][lo:hi].to(dev), train=False, need_grad=True, want_encoder=False)
        eng.backward(scale)
        torch.cuda.synchronize()
        return float(loss)

    loss_big = run(0, bsz, 1.0)
    g_big = eng.grads.clone()
    ntot = int((b["labels"] != -100).sum())
    acc = torch.zeros_like(g_big)
    loss_acc = 0.0
    for c in range(0, bsz, chunk):
        w = int((b["labels"][c: c + chunk] != -100).sum()) / ntot
        loss_acc += w * run(c, min(c + chunk, bsz), w)
        acc += eng.grads
    worst, worst_name = 0.0, ""
    for name, (off, rows, cols) in eng.index.items():
        if "k_proj.bias" in name:   # zero true gradient (softmax is shift-invariant): rounding noise only
            continue
        a, r = g_big[off: off + rows * cols], acc[off: off + rows * cols]
        e = float((a - r).norm() / (r.norm() + 1e-30))
        if e > worst or e != e:
            worst, worst_name = e, name
    loss_rel = abs(loss_big - loss_acc) / abs(loss_acc)
    # measured: loss 2e-6, worst gradient 7.5e-3 (a decoder q/k projection: bf16 rounding of per-row quantities scaled by
    # different non-power-of-two weights in the batch and in its chunks); a broken kernel is off by O(1) or NaN
    ok = bool(loss_rel < 1e-3 and worst < 3e-2)
    out = {"per_gpu_batch": bsz, "chunks": (bsz + chunk - 1) // chunk, "loss_rel_err": float("%.3e" % loss_rel),
           "worst_gradient_rel_err": float("%.3e" % worst), "worst_gradient": worst_name, "ok": ok,
           "bounds": {"loss": 1e-3, "gradient_norm_wise": 3e-2}}
    if not ok:   # reported, not raised: the JSON line must still come out, with the failure in it (`self_checks_ok`)
        print("[bench] BENCH-BATCH CHECK FAILED: %s" % json.dumps(out), file=sys.stderr, flush=True)
    return out


def timed_steps(step, n_warm, n_steps):
    for _ in range(n_warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_steps):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n_steps


def fine_tune_leg(model, opt, dev, batch_size, n_steps=24):
    """The drop-in API path (reference src/training.py:96-171 as vcg_train.py drives it): `src.training.fine_tune`
    over pinned HOST batches behind kmbart.data.DevicePrefetcher -- model.forward -> loss.backward() through autograd
    -> optimizer.step(), per-step loss read (deferred by one step), nothing skipped.  Returns tokens/s."""
    import types
    from kmbart.data import DevicePrefetcher, PackedFeatures
    from src.data.synthetic import make_batch
    from src.training import fine_tune
    host = []
    for i in range(3):
        hb = make_batch(batch_size, enc_len=S_ENC, dec_len=T_DEC, num_regions=REGIONS, seed=199 + i)
        hb["image_features"] = PackedFeatures.from_list(hb["image_features"], 2052, pin=True)
        host.append({k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in hb.items()})

    class Loader:
        def __init__(self, n):
            self.n = n

        def __iter__(self):
            return (host[i % len(host)] for i in range(self.n))

        def __len__(self):
            return self.n

    args = types.SimpleNamespace(amp=False, epochs=1)
    fine_tune(0, model, DevicePrefetcher(Loader(4), dev), opt, dev, args)      # warm-up epoch
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fine_tune(0, model, DevicePrefetcher(Loader(n_steps), dev), opt, dev, args)
    torch.cuda.synchronize()
    return batch_size * (S_ENC + T_DEC) * n_steps / (time.perf_counter() - t0)


def generation_leg(dev, batch=64, beams=5, max_length=20, reps=5):
    """BASELINE config 5: beam-5 KV-cached generation (reference vcg_generate.py -> src/generation.py:22-32) on a
    FRESH random-init vcg_base (the training legs teach the benchmark model to emit </s> at once -- it is the most
    frequent target of the synthetic batches -- which would end every search after one step).  A decode step is
    HBM-bound: the decoder's bf16 weights + the tied head are read once per step whatever the batch, plus the logits
    and the K/V caches; `hbm_frac` = those bytes / measured step time / 6.3 TB/s achievable."""
    from src.data.synthetic import make_batch
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration
    torch.manual_seed(0)
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(VCG_BASE)).to(dev).eval()
    # N(0, 0.02) weights give logits of standard deviation ~0.5: every beam decision is a near-tie and bf16 rounding
    # picks the winner.  The tied matrix is scaled x8 (logit std ~4, still random-init, same work per step) so that the
    # search is decided by the model and the fused-vs-unfused id comparison below means something.
    with torch.no_grad():
        model._engine.view(model._engine.params, "model.shared.weight").mul_(8.0)
    model._engine.sync_params()
    b = make_batch(batch, seed=4321)
    kw = dict(input_ids=b["input_ids"].to(dev), image_features=[f.to(dev) for f in b["image_features"]],
              attention_mask=b["attention_mask"].to(dev), num_beams=beams, num_return_sequences=1,
              max_length=max_length, early_stopping=True)
    def timed(n, **over):
        k2 = dict(kw, **over)
        o = model.generate(**k2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            o = model.generate(**k2)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n, o

    dt, out = timed(reps)
    # correctness gate on the measured path: the fused decode blocks (csrc/decode.hip, what `value` times) must return
    # the token ids of the launch-per-operation path (KMB_GEN_FUSED=0: GEMM / attention / LayerNorm kernels that
    # tests/test_model_gpu.py pins on the oracle) on the same inputs
    prev = os.environ.get("KMB_GEN_FUSED")
    out, sc = model.generate(return_scores=True, **kw)
    os.environ["KMB_GEN_FUSED"] = "0"
    try:
        ref_ids, ref_sc = model.generate(return_scores=True, **kw)
    finally:
        if prev is None:
            del os.environ["KMB_GEN_FUSED"]
        else:
            os.environ["KMB_GEN_FUSED"] = prev
    # The two paths sum the same bf16 products in different orders (logits agree to ~1e-2 relative, as each agrees with
    # the oracle, tests/test_decode_fused_gpu.py), and this benchmark model is RANDOM-INIT: its beam hypotheses are
    # near-ties.  So: rows are compared id for id, and a row that differs must be a tie -- the length-normalised scores
    # of the two paths' winners within 2e-2 of each other -- otherwise the leg fails.
    n_cmp = min(out.shape[1], ref_ids.shape[1])
    same_row = (out[:, :n_cmp] == ref_ids[:, :n_cmp]).all(dim=1) if out.shape[0] == ref_ids.shape[0] else torch.zeros(0, dtype=torch.bool)
    rows_equal = int(same_row.sum())
    ids_match = bool(out.shape == ref_ids.shape and rows_equal == out.shape[0])
    gap = float((sc.float() - ref_sc.float()).abs().max())
    diff_gap = float((sc.float() - ref_sc.float())[~same_row.cpu()].abs().max()) if rows_equal < out.shape[0] else 0.0
    same_gap = float((sc.float() - ref_sc.float())[same_row.cpu()].abs().max()) if rows_equal else 0.0
    # every differing row must be a near-tie (its two winners' scores within 2e-2), and at least 90 % of the rows identical
    gen_ok = bool(out.shape[0] == ref_ids.shape[0] and rows_equal >= 0.9 * out.shape[0] and
                  (rows_equal == out.shape[0] or diff_gap <= 2e-2))
    if not gen_ok:
        print("[bench] GENERATION CHECK FAILED: fused decode blocks disagree with the launch-per-operation path (%d/%d rows "
              "identical, score gap of differing rows %.3e)" % (rows_equal, out.shape[0], diff_gap), file=sys.stderr, flush=True)
    dt32, out32 = timed(3, max_length=32)       # the "32-token-out" variant
    dt1, out1 = timed(3, num_beams=1)           # greedy: the reference CLI's default (vcg_generate.py:99)
    steps = out.shape[1] - 1
    d, L, F, V = VCG_BASE["d_model"], VCG_BASE["decoder_layers"], VCG_BASE["decoder_ffn_dim"], VCG_BASE["vocab_size"]
    R = batch * beams
    weights = 2 * (L * (6 * d * d + 2 * d * F) + V * d)                    # bf16: self q/k/v/o, cross q/o, FFN, tied head
    vpad = (V + 127) // 128 * 128
    logits = R * vpad * model._engine.gen_logits_bytes                           # written by the head GEMM ...
    if 256 < R <= 320 and os.environ.get("KMB_GEN_HEAD_STATS", "1") != "0":
        # ... and read back only where a row's best tokens can be (round 6): per row the (max, sum-exp) pairs of the 256-column blocks
        # (written by the GEMM's epilogue, read by the beam step) and 2 * beams + 1 blocks of 1 KB
        nblk = (V + 255) // 256
        logits += R * (2 * nblk * 8 + (2 * beams + 1) * 1024)
    else:
        logits *= 2                                                              # ... the top-k kernel streams them again
    del model
    self_kv = 2 * L * R * (steps / 2.0) * d * 2                             # average cache length
    cross_kv = 2 * L * batch * S_ENC * d * 2                                # per batch item, shared by its beams
    per_step = weights + logits + self_kv + cross_kv
    return {"metric": "generate_sequences_per_sec", "value": round(batch / dt, 1), "unit": "sequences/s",
            "ms_per_generate": round(dt * 1e3, 2), "decoder_steps": int(steps),
            "us_per_decoder_step": round(dt / steps * 1e6, 1),
            "hbm_bytes_per_step": int(per_step), "hbm_frac": round(per_step / (dt / steps) / 6.3e12, 4),
            # SURVEY.md section 8d's algorithmic bytes (weights + self-KV + cross-KV, NO logits round trip) over the 8 TB/s peak
            "hbm_bytes_per_step_survey": int(weights + self_kv + cross_kv),
            "hbm_frac_survey": round((weights + self_kv + cross_kv) / (dt / steps) / 8.0e12, 4),
            "gen_ids_match": ids_match, "gen_check_ok": gen_ok, "gen_rows_identical": "%d/%d" % (rows_equal, int(out.shape[0])),
            "gen_score_gap_max": float("%.3e" % gap), "gen_score_gap_of_differing_rows": float("%.3e" % diff_gap),
            "gen_score_gap_of_identical_rows": float("%.3e" % same_gap),
            "gen_ids_note": "fused decode blocks vs KMB_GEN_FUSED=0 (launch-per-operation path) on the same inputs; "
                            "length-normalised beam scores of the two paths differ by their bf16 rounding even on identical ids",
            "max_length_32": {"value": round(batch / dt32, 1), "ms_per_generate": round(dt32 * 1e3, 2),
                              "decoder_steps": int(out32.shape[1] - 1)},
            "greedy": {"value": round(batch / dt1, 1), "ms_per_generate": round(dt1 * 1e3, 2),
                       "decoder_steps": int(out1.shape[1] - 1)},
            "config": {"workload": "vcg_base generate, beam search, KV cache", "batch": batch, "num_beams": beams,
                       "max_length": max_length}}


def spawn_ranks(n):
    """One child `python -m torch.distributed.run --nnodes=1 --nproc-per-node n ... bench.py <same arguments>`; its stdout /
    stderr are inherited, so rank 0's JSON line comes out of this process's stdout unchanged.  Returns the exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:   # a free rendezvous port on the loopback interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] --gpus %d without WORLD_SIZE: launching %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env)


class PowerProbe:
    """Board power (hwmon PPT, W) and the hwmon shader clock of THIS process's card, sampled every 10 ms by a daemon thread while the timed
    windows run (best effort: silently absent where /sys is not readable).  Round 6 found that the b = 1024 step holds the MI355X at its
    1400 W cap and that the clock -- hence `value` and `roofline.frac` -- follows the power a kernel draws (profiles/r06_power_and_clock.txt)."""

    def __init__(self, dev):
        import glob
        import threading
        self.samples, self._stop, self.cap, self._thread = [], False, None, None
        try:
            p = torch.cuda.get_device_properties(dev)
            bus = "%04x:%02x:%02x" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
            card = [c for c in glob.glob("/sys/class/drm/card*/device") if bus in os.path.realpath(c)][0]
            self._pw = glob.glob(card + "/hwmon/hwmon*/power1_input")[0]
            self._fq = glob.glob(card + "/hwmon/hwmon*/freq1_input")[0]
            self.cap = int(open(glob.glob(card + "/hwmon/hwmon*/power1_cap")[0]).read()) / 1e6
            self._thread = threading.Thread(target=self._run, daemon=True)
        except Exception:
            self._thread = None

    def _run(self):
        while not self._stop:
            try:
                self.samples.append((int(open(self._pw).read()) / 1e6, int(open(self._fq).read()) / 1e6))
            except Exception:
                pass
            time.sleep(0.01)

    def start(self):
        if self._thread is not None:
            self._thread.start()

    def report(self):
        self._stop = True
        if self._thread is None or len(self.samples) < 8:
            return None
        w = sorted(x[0] for x in self.samples)
        f = sorted(x[1] for x in self.samples)
        return {"cap_w": self.cap, "median_w": round(w[len(w) // 2]), "p90_w": round(w[int(0.9 * len(w))]),
                "hwmon_clock_mhz_median": round(f[len(f) // 2]), "samples": len(w),
                "source": "hwmon power1_input / freq1_input of this card, 10 ms period, over the timed windows"}


class ClockProbe:
    """The shader clock the chip held between two points of the current stream (kmb_clock_stamp: s_memtime over
    s_memrealtime per XCD).  A box that runs the GEMMs 3 % slower shows it here, not as a regression of the code."""

    def __init__(self, lib, dev):
        self.lib, self.buf = lib, torch.zeros((2, 16), dtype=torch.int64, device=dev)

    def stamp(self, i):
        from kmbart import _lib
        _lib.check(self.lib.kmb_clock_stamp(C.c_void_p(self.buf[i].data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream)))

    def mhz(self):
        b = self.buf.cpu().view(2, 8, 2)
        out = []
        for x in range(8):
            dc, dr = int(b[1, x, 0] - b[0, x, 0]), int(b[1, x, 1] - b[0, x, 1])
            if b[0, x, 1] and b[1, x, 1] and dr > 0:
                out.append(dc / dr * 100.0)
        out.sort()
        return round(out[len(out) // 2], 1) if out else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024,
                    help="per-GPU batch (weak scaling); 1024 x 96 tokens uses ~60 of the 288 GB; the GEMM grids fill better "
                         "with every doubling (`batch_sweep` reports 64 .. 2048: 0.79 / 1.53 / 1.72 / 1.80 / 1.89 M "
                         "tokens/s); rounds 1-2 quoted 512; the reference default is 64")
    ap.add_argument("--windows", type=int, default=3,
                    help="timed windows of --steps steps each; `value` / `ms_per_step` are the median window's")
    ap.add_argument("--lr", type=float, default=1e-5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-pcie", action="store_true", help="skip the host-batch (PCIe-inclusive) side measurement")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the fine_tune (API path), batch-sweep and generation legs")
    ap.add_argument("--serial", action="store_true",
                    help="profiling aid: weight-gradient GEMMs on the main stream (no overlap), so a rocprofv3 "
                         "kernel trace shows every kernel's stand-alone duration; not the product configuration")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as a plain command (reference: vcg_train.py:350-355 mp.spawn's its ranks from one
        # command): start N FRESH rank processes through torch.distributed.run -- before this process has made any GPU
        # call, and as children, never as a re-exec -- relay their output (rank 0 prints the JSON line) and their exit code
        sys.exit(spawn_ranks(args.gpus))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.set_num_threads(usable_cores())
    # test hooks for a one-GPU box: KMB_BENCH_ONE_DEVICE=1 puts every rank on cuda:0 and KMB_BENCH_BACKEND=gloo carries
    # the collectives (RCCL refuses two ranks on one device), so the N > 1 control flow can be run end to end
    if os.environ.get("KMB_BENCH_ONE_DEVICE", "0") == "1":
        local = 0
    backend = os.environ.get("KMB_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    # KMB_BENCH_FORCE_DIST=1: take the multi-GPU code path (process group, bucketed RCCL all-reduce on the side stream,
    # barrier + MAX over ranks) with ONE rank -- the only way to run that path on a one-GPU box
    force_dist = os.environ.get("KMB_BENCH_FORCE_DIST", "0") == "1"
    use_dist = world > 1 or force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from kmbart import _lib
    from kmbart.optim import AdamW
    from kmbart.parallel import DistributedDataParallel
    from src.data.synthetic import make_batch
    from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration

    torch.manual_seed(0)  # identical random-init weights on every rank (no checkpoints offline)
    model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(VCG_BASE))
    model.to(dev)
    model._engine.set_seed(1234 + rank)
    if args.serial:
        _lib.load().kmb_set_side_stream(model._engine.h, 0)
    ddp = DistributedDataParallel(model, device_ids=[local], reduce_single_rank=force_dist) if use_dist else model
    ddp.train()
    opt = AdamW(model.parameters(), lr=args.lr)
    opt.allow_overlap(os.environ.get("KMB_BENCH_NO_OPT_OVERLAP", "0") != "1")   # train_step_fwd_bwd + step run back to back: nothing touches the gradients in between (the env hook: A/B of the per-bucket overlap)
    if use_dist:
        ddp.attach_optimizer(opt)   # each piece's AdamW right behind its all-reduce on the communication stream

    parity = None
    if rank == 0 and args.gpus == 1:
        parity = loss_parity(model, dev)

    b = make_batch(args.batch, enc_len=S_ENC, dec_len=T_DEC, num_regions=REGIONS, seed=1234 + rank)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["image_features"] = [f.to(dev) for f in b["image_features"]]
    # packed once: the timed region starts with inputs resident in HBM as ONE [Ntot, 2052] buffer + CSR offsets, the form
    # the collator / DevicePrefetcher deliver (the list-of-tensors H2D + cat of the reference is row (f), not the step)
    from kmbart.data import PackedFeatures
    batch["image_features"] = PackedFeatures.from_list(b["image_features"], 2052).to(dev)

    def step():
        loss = ddp.train_step_fwd_bwd(batch)
        opt.step()
        return loss

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    # THREE timed windows of exactly --steps steps each, every one bracketed by barrier + synchronize and reduced with
    # MAX over ranks; `value` is the MEDIAN window, `value_min` / `value_max` and `clock_mhz` (shader clock held inside each
    # window) go beside it: a round whose gains are <= 3 % cannot be read off one window on one box
    probe = ClockProbe(_lib.load(), dev)
    power = PowerProbe(dev) if rank == 0 else None
    if power is not None:
        power.start()
    windows = []
    for _w in range(args.windows):
        fence()
        probe.stamp(0)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        probe.stamp(1)
        fence()
        wdt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([wdt], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wdt = float(t)
        windows.append((wdt, probe.mhz()))
    board_power = power.report() if power is not None else None
    order = sorted(range(len(windows)), key=lambda i: windows[i][0])
    dt, clock = windows[order[len(order) // 2]]
    last_loss = float(loss)

    tokens_per_step = world * args.batch * (S_ENC + T_DEC)
    value = tokens_per_step * args.steps / dt
    out = {
        "metric": "train_tokens_per_sec", "value": round(value, 1), "unit": "tokens/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "vcg_base train step (fwd+bwd+allreduce+AdamW), dropout 0.1, %d regions x 2052-d, "
                               "%d enc tokens, %d dec tokens, random-init weights" % (REGIONS, S_ENC, T_DEC),
                   "global_batch": world * args.batch, "per_gpu_batch": args.batch, "parallelism": "dp%d" % world,
                   "target_tokens_per_sec": round(world * args.batch * T_DEC * args.steps / dt, 1)},
        "windows": len(windows), "value_min": round(tokens_per_step * args.steps / windows[order[-1]][0], 1),
        "value_max": round(tokens_per_step * args.steps / windows[order[0]][0], 1),
        "ms_per_step_windows": [round(w[0] / args.steps * 1e3, 3) for w in windows],
        "clock_mhz": clock, "clock_mhz_windows": [w[1] for w in windows], "board_power": board_power,
        "final_loss": round(last_loss, 4),
        "model_tflops_per_gpu": round(value / world * GFLOP_PER_TOKEN / 1e3, 1),
        "mfma_frac_whole_step": round(value / world * GFLOP_PER_TOKEN / 1e3 / PEAK_BF16_TFLOPS, 4),
    }
    if parity is not None:
        out["ce_loss_rel_delta_vs_oracle_b2"] = float("%.3e" % parity[0])

    leg_steps = {}
    if use_dist and hasattr(ddp, "comm_report"):
        out["comm"] = ddp.comm_report()   # every rank measures (matched collectives); rank 0 prints

    if rank == 0 and args.gpus == 1 and not args.no_pcie:
        # side measurement, never `value`: batches start in pinned HOST memory (fp32 region features, 295 KB/sample)
        # and reach the GPU through the packed-feature prefetcher (copy of batch i+1 overlaps step i)
        from kmbart.data import DevicePrefetcher, PackedFeatures
        host = []
        for i in range(3):
            hb = make_batch(args.batch, enc_len=S_ENC, dec_len=T_DEC, num_regions=REGIONS, seed=99 + i)
            hb["image_features"] = PackedFeatures.from_list(hb["image_features"], 2052, pin=True)
            host.append(hb)
        n_host = 12

        def host_loader():
            for i in range(n_host):
                yield host[i % len(host)]

        class _L:
            def __iter__(self):
                return host_loader()

            def __len__(self):
                return n_host

        torch.cuda.synchronize()
        th = None
        for i, db in enumerate(DevicePrefetcher(_L(), dev)):
            if i == 2:
                torch.cuda.synchronize()
                th = time.perf_counter()
            ddp.train_step_fwd_bwd(db)
            opt.step()
        torch.cuda.synchronize()
        out["pcie_inclusive_tokens_per_sec"] = round(args.batch * (S_ENC + T_DEC) * (n_host - 2) / (time.perf_counter() - th), 1)

    if rank == 0 and args.gpus == 1 and not args.no_extras:
        # (1) the drop-in API path: src.training.fine_tune over host batches (autograd loss.backward, optimizer.step)
        out["fine_tune_tokens_per_sec"] = round(fine_tune_leg(model, opt, dev, args.batch), 1)
        out["fine_tune_over_value"] = round(out["fine_tune_tokens_per_sec"] / value, 4)
        # (2) other per-GPU batch sizes (the reference's default is 64, vcg_train.py:330); same step as `value`
        sweep = {}
        for bsz in (64, 256, 512, 1024, 2048):   # 2048 x 96 tokens: ~120 of the 288 GB
            if bsz == args.batch:
                continue
            sb = make_batch(bsz, enc_len=S_ENC, dec_len=T_DEC, num_regions=REGIONS, seed=77)
            sbatch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in sb.items()}
            sbatch["image_features"] = PackedFeatures.from_list(sb["image_features"], 2052).to(dev)

            def sstep(sbatch=sbatch):
                model.train_step_fwd_bwd(sbatch)
                opt.step()
            sdt = timed_steps(sstep, 4, 10 if bsz <= 512 else 6)
            sweep[str(bsz)] = {"tokens_per_sec": round(bsz * (S_ENC + T_DEC) / sdt, 1), "ms_per_step": round(sdt * 1e3, 3)}
            if bsz in (64, 256, 512, 1024):
                leg_steps[bsz] = (sstep, sweep[str(bsz)])
        sweep[str(args.batch)] = {"tokens_per_sec": round(value, 1), "ms_per_step": round(dt / args.steps * 1e3, 3)}
        out["batch_sweep"] = sweep
        out["legs"] = sweep   # the same objects: b = 64 / 256 / 512 / 1024 get their GEMM roofline below
        # (3) the measured batch computes what its 64-sample chunks compute (reported in `bench_batch_check.ok`, folded into `self_checks_ok`)
        out["bench_batch_check"] = bench_batch_check(model, dev, args.batch)
        model.train()
        # (4) BASELINE config 5: generation
        out["generation"] = generation_leg(dev)
        out["self_checks_ok"] = bool(out["bench_batch_check"]["ok"] and out["generation"]["gen_check_ok"] and
                                     (parity is None or parity[0] < 1e-3))

    if rank == 0 and not args.no_roofline:
        model._post_backward = None   # rank 0 only from here on: no collectives (the other ranks are at the final barrier)
        out["roofline"] = gemm_roofline(model, step, args.batch)
        if args.gpus == 1 and not args.no_extras:
            # every named batch carries its own GEMM roofline (SURVEY section 8d names b = 64 and 256; rounds 1-2 quoted
            # 512, round 2's default is 1024): round-to-round comparisons do not depend on the default
            for bsz, (sstep, leg) in sorted(leg_steps.items()):
                r = gemm_roofline(model, sstep, bsz, n_prof=2)
                leg["roofline"] = {k: r[k] for k in ("achieved", "frac", "launches_per_step", "gemm_ms_per_step", "by_variant",
                                                     "traffic", "algorithmic_bytes_per_launch")}
            out["legs"][str(args.batch)]["roofline"] = {k: out["roofline"][k] for k in (
                "achieved", "frac", "launches_per_step", "gemm_ms_per_step", "by_variant", "traffic",
                "algorithmic_bytes_per_launch")}
    if rank == 0 and args.gpus == 1 and not args.no_cpu_baseline:
        print("[bench] GPU leg done: %.1f tokens/s, %.3f ms/step; timing the CPU baseline..." %
              (value, dt / args.steps * 1e3), file=sys.stderr, flush=True)
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
