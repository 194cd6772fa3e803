"""Host->device batch pipeline (the step right upstream of the hot path).

The reference moves every field with a blocking `.to(device)` inside the step and hands the model a Python list of
per-sample feature tensors (src/training.py:120-129; 295 KB per sample of fp32 region features).  Here the ragged
list is packed ONCE on the host into a [Ntot, 2052] pinned buffer with CSR offsets, copied on a side stream, and
the next batch's copy overlaps the current step."""
import torch


class PackedFeatures:
    """Region features of a batch as one [Ntot, F] tensor + int32 offsets [B+1] (what kmb_batch carries)."""

    def __init__(self, packed, offsets, n_total):
        self.packed, self.offsets, self.n_total = packed, offsets, int(n_total)

    @classmethod
    def from_list(cls, image_features, feat_dim, pin=False):
        lens = [int(x.shape[0]) if x.dim() == 2 else 0 for x in image_features]
        offs = [0]
        for n in lens:
            offs.append(offs[-1] + n)
        non_empty = [x for x, n in zip(image_features, lens) if n > 0]
        packed = torch.cat(non_empty, 0).float().contiguous() if non_empty else torch.zeros((1, feat_dim))
        offsets = torch.tensor(offs, dtype=torch.int32)
        if pin and torch.cuda.is_available():
            packed, offsets = packed.pin_memory(), offsets.pin_memory()
        return cls(packed, offsets, offs[-1])

    def to(self, device, non_blocking=False):
        return PackedFeatures(self.packed.to(device, non_blocking=non_blocking),
                              self.offsets.to(device, non_blocking=non_blocking), self.n_total)

    def __len__(self):
        return self.offsets.numel() - 1

    # list-of-tensors view (what the reference's collator hands out, collation.py:73-76): views, no copies
    def __getitem__(self, i):
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        a, b = int(self.offsets[i]), int(self.offsets[i + 1])
        return self.packed[a:b] if b > a else torch.empty(0)

    def __iter__(self):
        for i in range(len(self)):
            yield self[i]

    def as_list(self):
        return list(self)


class DevicePrefetcher:
    """Iterates a loader of collated batches (dicts), returning batches whose tensors already live on `device`;
    the copy of batch i+1 is issued on a side stream while batch i is being consumed."""

    def __init__(self, loader, device, feat_dim=2052):
        self.loader, self.device, self.feat_dim = loader, torch.device(device), feat_dim
        self.stream = torch.cuda.Stream(device=self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        out = {}
        with torch.cuda.stream(self.stream):
            for k, v in batch.items():
                if k == "image_features" and not isinstance(v, PackedFeatures):
                    v = PackedFeatures.from_list(v, self.feat_dim, pin=True)
                if isinstance(v, PackedFeatures):
                    out[k] = v.to(self.device, non_blocking=True)
                elif torch.is_tensor(v):
                    out[k] = (v.pin_memory() if not v.is_pinned() else v).to(self.device, non_blocking=True)
                else:
                    out[k] = v
        return out

    def __iter__(self):
        it = iter(self.loader)
        try:
            nxt = self._stage(next(it))
        except StopIteration:
            return
        while nxt is not None:
            main = torch.cuda.current_stream(self.device)
            main.wait_stream(self.stream)
            cur = nxt
            # the tensors were allocated on the copy stream: tell the caching allocator that the compute stream uses
            # them too, or their memory is handed to the NEXT staging copy while kernels of this step still read it
            # (seen as HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION at batch 512: offsets overwritten by features)
            for v in cur.values():
                if isinstance(v, PackedFeatures):
                    v.packed.record_stream(main)
                    v.offsets.record_stream(main)
                elif torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(main)
            try:
                nxt = self._stage(next(it))
            except StopIteration:
                nxt = None
            yield cur
