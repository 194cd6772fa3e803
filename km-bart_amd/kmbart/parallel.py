"""Single-node data parallelism: one process per GPU, gradient all-reduce over RCCL/xGMI overlapped
with backward.  Replaces torch DistributedDataParallel as used by the reference
(vcg_train.py:98: DDP(model, device_ids=[rank], find_unused_parameters=True); src/utils.py:13).

Semantics kept: parameters are broadcast from rank 0 at wrap time; after `loss.backward()` every
rank holds the arithmetic MEAN over ranks of the per-rank gradients (each rank's loss is the mean over
its own non-ignored tokens); `.module` reaches the wrapped model.

Mechanics: the engine's gradients are one flat fp32 arena cut into buckets in backward completion
order (decoder layers last->first, decoder embedding, encoder layers, encoder embedding + image
projection, tied matrix).  kmb_backward records an event per bucket on the compute stream; the
reducer makes the communication stream wait on that event and issues the bucket's all-reduce, so
the collectives run while the rest of backward is still computing.  `find_unused_parameters` is
not needed: every gradient is written every step (no per-step bitmap collective).

The optimizer tail.  With `attach_optimizer(opt)` (training loops that run `loss.backward()` and `opt.step()` back to
back: src.training.fine_tune / pretrain, bench.py) each piece's fused AdamW is enqueued on the COMMUNICATION stream
right behind that piece's all-reduce, so the HBM-bound update of a bucket runs while later buckets are still being
reduced and while backward is still computing; `opt.step()` then only makes the compute stream wait for the
communication stream.  The tied matrix (154 MB, complete last) travels in <= 64 MB pieces, each followed by its own
update, so the un-overlapped tail is one piece, not 564 MB of all-reduce plus a 1.1 ms optimizer launch.
`grad_dtype="bf16"` halves the bytes on the wire (282 MB): a piece is cast into a bf16 staging buffer, reduced, and
cast back (fp32 accumulation inside the optimizer is unchanged); off by default -- xGMI has the bandwidth at the
benchmark batch, SURVEY.md section 5 prices when it does not.

Native exchange (nccl backend; the default for a one-rank group, OPT-IN with KMB_DP_NATIVE=1 / native=True for more than one
rank until a multi-GPU run of tests/dp_rccl_worker.py has passed on hardware: none of its world > 1 collectives has ever
executed; their offsets are pinned on the CPU by tests/test_comm_plan_cpu.py).  The library owns an RCCL communicator and a communication stream
(include/kmbart.h "data parallelism: native RCCL"): `kmb_allreduce_grads` enqueues EVERY bucket's ncclAllReduce(avg)
behind its completion event -- and, with an optimizer attached, each piece's fused AdamW behind its collective -- from
C++ in one call; torch.distributed only bootstraps the communicator's id and serves bench.py's barrier.  `KMB_DP_ALGO=rsag`
(or algo="rsag") selects ncclReduceScatter -> AdamW on this rank's shard -> ncclAllGather of the updated parameters
(optimizer HBM traffic / world; exp_avg / exp_avg_sq are then valid on the owning rank only:
`gather_optimizer_state()` before saving a checkpoint).  `native=False`, the gloo backend and `grad_dtype="bf16"` take the
torch.distributed path below (BucketedAllReducer), which the CPU and two-ranks-on-one-GPU tests exercise.
"""
import os

import torch
import torch.distributed as dist


class BucketedAllReducer:
    """All-reduce (mean) of slices of one flat tensor, in a given bucket order.

    wait_ready(i, stream): optional hook that makes `stream` wait until bucket i is complete
    (HIP event on the GPU path; None on the CPU/gloo test path where backward has already finished).
    """

    def __init__(self, flat, buckets, process_group=None, wait_ready=None, comm_stream=None, max_bucket_elems=None,
                 always=False, grad_dtype=None, after_piece=None):
        self.flat = flat
        self.grad_dtype = grad_dtype     # None: reduce the fp32 slices in place; torch.bfloat16: staged bf16 pieces
        self.after_piece = after_piece   # callable(offset, count) run on the communication stream behind each piece
        self.always = always   # issue the collectives even in a one-rank group (single-GPU test of the RCCL path)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.wait_ready = wait_ready
        self.comm_stream = comm_stream
        self.backend = dist.get_backend(process_group) if dist.is_initialized() else None
        # split very large buckets (the tied matrix is 154 MB) so that the tail latency is one piece, not the whole
        self.pieces = []
        for i, (off, cnt) in enumerate(buckets):
            if max_bucket_elems and cnt > max_bucket_elems:
                n = (cnt + max_bucket_elems - 1) // max_bucket_elems
                step = ((cnt + n - 1) // n + 63) // 64 * 64   # piece boundaries stay 64-element aligned (fused AdamW)
                for s in range(0, cnt, step):
                    self.pieces.append((i, off + s, min(step, cnt - s)))
            else:
                self.pieces.append((i, off, cnt))
        self._works = []

    def launch(self):
        if self.world == 1 and not self.always:
            return
        use_avg = self.backend == "nccl"
        op = dist.ReduceOp.AVG if use_avg else dist.ReduceOp.SUM
        on_gpu = self.flat.is_cuda
        ctx = torch.cuda.stream(self.comm_stream) if (on_gpu and self.comm_stream is not None) else _Null()
        waited = set()
        with ctx:
            for i, off, cnt in self.pieces:
                if self.wait_ready is not None and i not in waited:
                    stream = self.comm_stream
                    if stream is None and on_gpu:
                        stream = torch.cuda.current_stream()
                    self.wait_ready(i, stream)
                    waited.add(i)
                piece = self.flat[off: off + cnt]
                if self.grad_dtype is not None and self.grad_dtype != piece.dtype:
                    staged = piece.to(self.grad_dtype)
                    w = dist.all_reduce(staged, op=op, group=self.group, async_op=True)
                    self._works.append((w, piece, use_avg, staged, off, cnt))
                else:
                    w = dist.all_reduce(piece, op=op, group=self.group, async_op=True)
                    self._works.append((w, piece, use_avg, None, off, cnt))
                if use_avg and self.after_piece is not None:
                    # RCCL work is stream-ordered: wait() only orders this stream behind the collective
                    self._complete(self._works.pop())

    def _complete(self, item):
        w, piece, use_avg, staged, off, cnt = item
        w.wait()
        if staged is not None:
            piece.copy_(staged)
        if not use_avg:
            piece.div_(self.world)
        if self.after_piece is not None:
            self.after_piece(off, cnt)

    def finish(self):
        """Makes the current stream (GPU) or the host (CPU) wait for every outstanding all-reduce (and for the
        per-piece work chained behind them)."""
        on_gpu = self.flat.is_cuda and self.comm_stream is not None
        ctx = torch.cuda.stream(self.comm_stream) if on_gpu else _Null()
        with ctx:
            for item in self._works:
                self._complete(item)
        self._works = []
        if on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)


class _Null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


class DistributedDataParallel(torch.nn.Module):
    def __init__(self, module, device_ids=None, find_unused_parameters=False, process_group=None,
                 max_bucket_mb=64, reduce_single_rank=False, grad_dtype=None, native=None, algo=None):
        super().__init__()
        self.__dict__["module"] = module  # not a registered child: parameters() must not be re-wrapped
        eng = module._need_engine()
        self.engine = eng
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.reducer = None
        self.native = False
        active = self.world > 1 or (reduce_single_rank and dist.is_initialized())
        if native is None:
            # world > 1 has never run on hardware through the library's own collectives: torch.distributed's reducer (same
            # RCCL, same buckets, same fused optimizer pieces) stays the default there until it has; KMB_DP_NATIVE=1 opts in
            native = os.environ.get("KMB_DP_NATIVE", "1" if self.world == 1 else "0") != "0"
        if algo is None:
            algo = os.environ.get("KMB_DP_ALGO", "allreduce")
        if algo not in ("allreduce", "rsag"):
            raise ValueError("algo must be 'allreduce' or 'rsag'")
        self.native_error = None
        use_native = bool(active and native and grad_dtype is None and dist.get_backend(process_group) == "nccl")
        if use_native:
            # The library's communicator is created here, by a collective call (ncclCommInitRank).  If that -- or the
            # parameter broadcast over it -- raises on any rank, EVERY rank takes the torch.distributed path below instead
            # (same RCCL, torch's communicator): the ranks agree on the outcome through the bootstrap group, the reason is
            # printed and kept for comm_report().  A failure is never silent and never a CPU path.
            ok = 1
            try:
                eng.comm_init(process_group)
                eng.comm_broadcast_params(0)      # C2: parameters and the logits-bias buffer start identical on every rank
            except Exception as e:                # noqa: BLE001 -- any failure of the native bootstrap takes the fallback
                ok, self.native_error = 0, "%s: %s" % (type(e).__name__, e)
            flag = torch.tensor([ok], dtype=torch.int32, device=eng.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
            if int(flag.item()) != 1:
                if ok:
                    self.native_error = "the native RCCL bootstrap failed on another rank"
                eng.comm_destroy()   # idempotent: also drops a communicator whose parameter broadcast raised on this rank
                import sys
                print("[kmbart] native RCCL exchange unavailable (%s): using the torch.distributed path" % self.native_error,
                      file=sys.stderr, flush=True)
                use_native = False
        if use_native:
            # native exchange: the library's own communicator and communication stream (module docstring)
            self.native = True
            self.algo = 1 if (algo == "rsag" and 8 % self.world == 0) else 0
            if algo == "rsag" and self.algo == 0:
                import warnings
                warnings.warn("algo='rsag' needs a world size that divides 8 (got %d): using all-reduce" % self.world)
            self.max_piece_elems = max_bucket_mb * (1 << 20) // 4
            self._opt = None
            module._post_backward = self._reduce_native
            from . import _lib
            _lib.load().kmb_gemm_shared_device(1)
            self._first_reduce = True
            self._fusing = False
            self._tail_events = []
            eng.adamw_overlap_ok = False
        elif active:
            # C2: parameters and the logits-bias buffer start identical on every rank
            dist.broadcast(eng.params, src=0, group=process_group)
            dist.broadcast(eng.final_logits_bias, src=0, group=process_group)
            eng.sync_params()
            comm = torch.cuda.Stream(device=eng.device)
            if grad_dtype in ("bf16", "bfloat16"):
                grad_dtype = torch.bfloat16
            self.reducer = BucketedAllReducer(
                eng.grads, eng.buckets(), process_group,
                wait_ready=lambda i, stream: eng.stream_wait_bucket(i, stream), comm_stream=comm,
                max_bucket_elems=max_bucket_mb * (1 << 20) // 4, always=reduce_single_rank, grad_dtype=grad_dtype,
                after_piece=self._step_piece)
            self._opt = None          # attach_optimizer(): fused AdamW behind each piece's all-reduce
            module._post_backward = self._reduce
            # RCCL's kernel holds CUs while backward runs: the persistent GEMMs hand out every tile dynamically
            from . import _lib
            _lib.load().kmb_gemm_shared_device(1)
            self._first_reduce = True
            self._fusing = False
            self._tail_events = []
            eng.adamw_overlap_ok = False   # gradients are final only after the all-reduce, not at the bucket events

    def attach_optimizer(self, optimizer):
        """Chains `optimizer`'s fused AdamW behind every gradient piece's all-reduce on the communication stream (module
        docstring).  Only for loops that call `optimizer.step()` right after `loss.backward()` with nothing touching the
        gradients in between: the parameters of a bucket are already updated when backward returns, `step()` becomes
        the point where the compute stream waits for the communication stream."""
        if (self.reducer is None and not self.native) or not hasattr(optimizer, "fused_piece_step"):
            return False
        self._opt = optimizer
        optimizer._ddp_fused = self
        return True

    def comm_report(self):
        """What the data-parallel exchange of a step looked like, for bench.py's JSON line: ranks, collective backend,
        gradient buckets / pieces, bytes all-reduced per step and the exposed tail -- the time the compute stream spent
        waiting for the communication stream after backward's last kernel (median over the last steps; syncs)."""
        if self.native:
            tails = []
            if self._tail_events:
                torch.cuda.synchronize(self.engine.device)
                tails = sorted(a.elapsed_time(b) for a, b in self._tail_events)
            from . import _lib
            return {"rccl_ranks": self.world, "backend": "rccl-native", "algo": "rsag" if self.algo == 1 else "allreduce",
                    "buckets": len(self.engine.buckets()),
                    "pieces": int(_lib.load().kmb_comm_pieces(self.engine.h, self.max_piece_elems)),
                    "bytes_reduced_per_step": int(sum(c for _, c in self.engine.buckets()) * 4), "wire_dtype": "fp32",
                    "fused_optimizer": self._opt is not None,
                    "exposed_tail_ms": round(tails[len(tails) // 2], 3) if tails else None,
                    "exposed_tail_ms_max": round(tails[-1], 3) if tails else None, "steps_measured": len(tails)}
        if self.reducer is None:
            return {"rccl_ranks": self.world, "reduced": False}
        r = self.reducer
        esz = 2 if r.grad_dtype == torch.bfloat16 else self.engine.grads.element_size()
        tails = []
        if self._tail_events:
            torch.cuda.synchronize(self.engine.device)
            tails = sorted(a.elapsed_time(b) for a, b in self._tail_events)
        return {"rccl_ranks": self.world, "backend": r.backend, "native_fallback": self.native_error,
                "buckets": len(self.engine.buckets()),
                "pieces": len(r.pieces), "bytes_reduced_per_step": int(sum(c for _, _, c in r.pieces) * esz),
                "wire_dtype": "bf16" if esz == 2 else "fp32", "fused_optimizer": self._opt is not None,
                "exposed_tail_ms": round(tails[len(tails) // 2], 3) if tails else None,
                "exposed_tail_ms_max": round(tails[-1], 3) if tails else None, "steps_measured": len(tails)}

    def detach_optimizer(self):
        """Ends the fusion (the training loops call this when they return): `loss.backward()` is a plain backward +
        all-reduce again and `optimizer.step()` a plain step."""
        if getattr(self, "_opt", None) is not None:
            self._opt._ddp_fused = None
            self._opt._fused_pending = False
        self._opt = None

    def _step_piece(self, off, cnt):
        if self._opt is not None and self._fusing:
            self._opt.fused_piece_step(self.engine, off, cnt)

    def gather_optimizer_state(self):
        """rsag: exp_avg / exp_avg_sq are updated on the owning rank only; this all-gathers them (before a checkpoint).
        A COLLECTIVE: every rank must call it (calling it on rank 0 only hangs).  `AdamW.state_dict()` refuses to dump
        sharded moments and a plain `AdamW.step()` gathers them first (every rank runs `step()`)."""
        if self.native and self.algo == 1:
            self.engine.comm_gather_moments()

    def _reduce_native(self):
        """loss.backward()'s tail on the native path: ONE library call enqueues the step's whole exchange (and the fused
        optimizer pieces) on the communication stream, a second makes the compute stream wait for it."""
        eng = self.engine
        first = self._first_reduce     # the first backward times the GEMM variants of every shape: keep RCCL out of it
        self._first_reduce = False
        if self._opt is not None and getattr(self._opt, "_fused_pending", False):
            raise RuntimeError("loss.backward() ran twice without optimizer.step() while an optimizer is attached to the "
                               "data-parallel wrapper (attach_optimizer fuses the AdamW step into backward); call "
                               "detach_optimizer() for gradient accumulation or custom loops")
        self._fusing = self._opt is not None and self._opt.begin_fused_step(eng)
        hp = self._opt.fused_hyperparams(eng) if self._fusing else None
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        pair[0].record()
        eng.allreduce_grads(algo=self.algo, adamw=hp, max_piece_elems=self.max_piece_elems, after_compute=first)
        eng.comm_wait()
        pair[1].record()
        self._tail_events.append(pair)
        del self._tail_events[:-32]
        if self._fusing:
            self._opt.end_fused_step(eng)
        self._fusing = False

    def _reduce(self):
        if self._first_reduce:
            # the first backward times the GEMM variants of every shape (autotune): keep the all-reduces out of it
            self._first_reduce = False
            if self.reducer.comm_stream is not None:
                self.reducer.comm_stream.wait_stream(torch.cuda.current_stream())
        if self._opt is not None and getattr(self._opt, "_fused_pending", False):
            raise RuntimeError("loss.backward() ran twice without optimizer.step() while an optimizer is attached to the "
                               "data-parallel wrapper (attach_optimizer fuses the AdamW step into backward); call "
                               "detach_optimizer() for gradient accumulation or custom loops")
        self._fusing = self._opt is not None and self._opt.begin_fused_step(self.engine)
        # exposed tail: the compute stream has nothing left to run between these two events except waiting for the
        # communication stream (the all-reduces that did not fit under backward + the optimizer pieces chained behind them)
        pair = None
        if self.engine.grads.is_cuda:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        self.reducer.launch()
        self.reducer.finish()
        if pair is not None:
            pair[1].record()
            self._tail_events.append(pair)
            del self._tail_events[:-32]
        if self._fusing:
            self._opt.end_fused_step(self.engine)
        self._fusing = False

    def forward(self, *args, **kwargs):
        return self.module.forward(*args, **kwargs)

    def generate(self, *args, **kwargs):
        return self.module.generate(*args, **kwargs)

    def parameters(self, recurse=True):
        return self.module.parameters()

    def named_parameters(self, prefix="", recurse=True, remove_duplicate=True):
        return self.module.named_parameters(prefix="module")

    def train(self, mode=True):
        self.module.train(mode)
        return super().train(mode)

    def eval(self):
        return self.train(False)

    def state_dict(self, *a, **k):
        return {"module." + n: v for n, v in self.module.state_dict().items()}

    def train_step_fwd_bwd(self, batch, loss_scale=1.0):
        return self.module.train_step_fwd_bwd(batch, loss_scale)
