"""When do the per-bucket AdamW launches of the overlapped optimizer step actually execute?  Timing events on the optimizer
stream behind every piece, against the start of the step and the end of backward on the caller's stream (no profiler: its
host overhead moves the answer).  python tools/adamw_overlap_events.py [batch ...]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "km-bart_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from kmbart._lib import KmbAdamW, check  # noqa: E402
from kmbart.data import PackedFeatures  # noqa: E402
from kmbart.optim import AdamW  # noqa: E402
from src.data.synthetic import make_batch  # noqa: E402
from src.model import MultiModalBartConfig, MultiModalBartForConditionalGeneration  # noqa: E402

dev = torch.device("cuda", 0)
model = MultiModalBartForConditionalGeneration(MultiModalBartConfig.from_dict(bench.VCG_BASE)).to(dev).train()
opt = AdamW(model.parameters(), lr=1e-5)
opt.allow_overlap(True)
eng = model._engine
for B in (int(a) for a in (sys.argv[1:] or ["64", "256", "1024"])):
    b = make_batch(B, seed=1)
    batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in b.items()}
    batch["image_features"] = PackedFeatures.from_list(b["image_features"], 2052).to(dev)
    for _ in range(6):
        model.train_step_fwd_bwd(batch)
        opt.step()
    side = eng._opt_stream
    main = torch.cuda.current_stream()
    buckets = eng._bucket_list
    out = []
    for rep in range(3):
        ev0 = torch.cuda.Event(enable_timing=True)
        evf = torch.cuda.Event(enable_timing=True)
        evb = torch.cuda.Event(enable_timing=True)
        ev0.record(main)
        loss, _, _ = eng.forward(batch["input_ids"], batch["image_features"], batch.get("attention_mask"),
                                 batch.get("decoder_input_ids"), batch.get("decoder_attention_mask"), batch["labels"],
                                 train=True, need_grad=True, want_logits=False, want_encoder=False)
        evf.record(main)
        model._backward(1.0)
        evb.record(main)
        eng.step_count += 1
        hp = KmbAdamW(lr=1e-5, beta1=0.9, beta2=0.999, eps=1e-6, weight_decay=0.0, step=eng.step_count, correct_bias=1, grad_scale=1.0)
        evs = []
        for i, (off, cnt) in enumerate(buckets):
            eng.stream_wait_bucket(i, side)
            check(eng.lib.kmb_adamw_step(eng.h, C.byref(hp), off, cnt, C.c_void_p(side.cuda_stream)))
            e = torch.cuda.Event(enable_timing=True)
            e.record(side)
            evs.append(e)
        main.wait_stream(side)
        eve = torch.cuda.Event(enable_timing=True)
        eve.record(main)
        torch.cuda.synchronize()
        out.append((ev0.elapsed_time(evf), ev0.elapsed_time(evb), [round(ev0.elapsed_time(e), 2) for e in evs], ev0.elapsed_time(eve)))
    for f, bw, pieces, end in out:
        print("batch %4d: forward ends %.2f ms, backward (caller's stream) ends %.2f, step ends %.2f; AdamW pieces end at %s" % (B, f, bw, end, pieces), flush=True)
