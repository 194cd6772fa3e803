"""CPU: the partition of the native RCCL gradient exchange for world sizes that have never run on hardware.

`kmb_allreduce_grads` (reduce-scatter -> sharded AdamW -> all-gather) and `kmb_comm_gather_moments` take every offset from
`kmb_comm_plan` (csrc/engine.cpp::comm_plan_piece), a pure host function of the arena layout.  With one rank every collective
is the identity and `mine == offset`, so the one-GPU tests (tests/test_dp_rccl_gpu.py) say nothing about the world > 1
arithmetic; this test pins it without a GPU: for W in {2, 4, 8} and both piece caps the shards of all ranks tile [0, arena)
exactly once, every shard is 8-element aligned (the fused optimizer's vector width), pieces stay inside their bucket, and the
image-projection weight's re-pad is requested by exactly the pieces / shards that overlap it.
Reference: DDP's bucket assignment, vcg_train.py:98; one process per GPU, src/utils.py:9-17.
"""
import ctypes as C

import pytest

from kmbart import _lib
from kmbart._lib import KmbCommPiece, KmbConfig, check

VCG_BASE = dict(vocab_size=50320, d_model=768, encoder_layers=6, decoder_layers=6, encoder_attention_heads=12,
                decoder_attention_heads=12, encoder_ffn_dim=3072, decoder_ffn_dim=3072, max_position_embeddings=1024,
                extra_pos_embeddings=2, image_feature_size=2052, pad_token_id=1, bos_token_id=0, eos_token_id=2,
                img_feat_id=50273, cls_token_id=50276, scale_embedding=0, dropout=0.1, attention_dropout=0.0,
                activation_dropout=0.0, layer_norm_eps=1e-5)
PRETRAIN = dict(VCG_BASE, num_labels=1601, num_attributes=129, num_relations=11)   # config/pretrain_base.json heads
TINY = dict(VCG_BASE, vocab_size=512, d_model=64, encoder_layers=2, decoder_layers=2, encoder_attention_heads=1,
            decoder_attention_heads=1, encoder_ffn_dim=128, decoder_ffn_dim=128, max_position_embeddings=64,
            img_feat_id=505, cls_token_id=508)
CAPS = (0, 16 << 20, 4 << 20, 1 << 18)   # 0 = the library default (16 Mi elements = the wrapper's 64 MB); smaller caps cut more


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__
    __graft_entry__.build()
    return _lib.load()


def _handle(lib, cfg):
    h = C.c_void_p()
    c = KmbConfig(**cfg)
    check(lib.kmb_create(C.byref(c), C.byref(h)))
    return h


def _plan(lib, h, world, rank, cap):
    out = []
    for i in range(lib.kmb_comm_pieces(h, cap)):
        pc = KmbCommPiece()
        check(lib.kmb_comm_plan(h, world, rank, cap, i, C.byref(pc)))
        out.append(pc)
    return out


def _image_weight_range(lib, h):
    name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int32(), C.c_int32()
    for i in range(lib.kmb_param_count(h)):
        check(lib.kmb_param_info(h, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols)))
        if name.value.decode() == "model.encoder.embed_images.linear.weight":
            return off.value, off.value + rows.value * cols.value
    raise AssertionError("no image projection weight")


@pytest.mark.parametrize("cfg_name", ["vcg_base", "pretrain_base", "tiny"])
@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_shards_tile_the_arena_exactly_once(lib, cfg_name, world):
    h = _handle(lib, {"vcg_base": VCG_BASE, "pretrain_base": PRETRAIN, "tiny": TINY}[cfg_name])
    try:
        arena = lib.kmb_arena_elems(h)
        buckets = []
        off, cnt = C.c_int64(), C.c_int64()
        for i in range(lib.kmb_bucket_count(h)):
            check(lib.kmb_bucket_range(h, i, C.byref(off), C.byref(cnt)))
            buckets.append((off.value, cnt.value))
        assert sorted(buckets)[0][0] == 0 and sum(c for _, c in buckets) == arena   # the buckets themselves tile the arena
        iw0, iw1 = _image_weight_range(lib, h)
        for cap in CAPS:
            plans = [_plan(lib, h, world, r, cap) for r in range(world)]
            n = len(plans[0])
            assert n == lib.kmb_comm_pieces(h, cap) and all(len(p) == n for p in plans)
            limit = cap if cap > 0 else 16 << 20
            pieces, shards = [], []
            last_bucket = -1
            for i in range(n):
                p0 = plans[0][i]
                # every rank sees the same piece (the collectives must match across ranks)
                for r in range(world):
                    pr = plans[r][i]
                    assert (pr.bucket, pr.offset, pr.count, pr.shard, pr.repad_piece) == \
                           (p0.bucket, p0.offset, p0.count, p0.shard, p0.repad_piece)
                    assert pr.mine == p0.offset + r * p0.shard
                    shards.append((pr.mine, pr.shard))
                    overlaps = pr.mine < iw1 and pr.mine + pr.shard > iw0
                    assert pr.repad_shard == int(overlaps), (cap, i, r)
                assert p0.bucket >= last_bucket, "pieces follow the buckets' completion order"
                last_bucket = p0.bucket
                boff, bcnt = buckets[p0.bucket]
                assert boff <= p0.offset and p0.offset + p0.count <= boff + bcnt
                assert p0.count > 0 and p0.count <= max(limit, 64) + 63
                assert p0.offset % 64 == 0 and p0.count % 64 == 0      # fused AdamW / 16-byte vector alignment
                assert p0.shard * world == p0.count and p0.shard % 8 == 0 and p0.shard > 0
                assert p0.repad_piece == int(p0.offset < iw1 and p0.offset + p0.count > iw0)
                pieces.append((p0.offset, p0.count))
            # pieces tile [0, arena) exactly once, and so do the shards of all ranks
            for spans in (pieces, shards):
                spans = sorted(spans)
                assert spans[0][0] == 0
                for (o1, c1), (o2, _) in zip(spans, spans[1:]):
                    assert o1 + c1 == o2, "gap or overlap at %d" % (o1 + c1)
                assert spans[-1][0] + spans[-1][1] == arena
            # the image weight's padded bf16 copy is rebuilt by at least one piece, and only by overlapping ones
            assert sum(p.repad_piece for p in plans[0]) >= 1
    finally:
        lib.kmb_destroy(h)


def test_plan_rejects_bad_arguments(lib):
    h = _handle(lib, TINY)
    try:
        pc = KmbCommPiece()
        assert lib.kmb_comm_plan(h, 0, 0, 0, 0, C.byref(pc)) != 0
        assert lib.kmb_comm_plan(h, 2, 2, 0, 0, C.byref(pc)) != 0
        assert lib.kmb_comm_plan(h, 2, 1, 0, lib.kmb_comm_pieces(h, 0), C.byref(pc)) != 0
        assert b"out of range" in lib.kmb_last_error()
        # a world size that does not divide the piece into 8-aligned shards is reported as shard 0 (algo 1 refuses it)
        check(lib.kmb_comm_plan(h, 3, 1, 0, 0, C.byref(pc)))
        assert pc.shard == 0 or (pc.shard * 3 == pc.count and pc.shard % 8 == 0)
        assert lib.kmb_comm_moments_sharded(h) == 0
    finally:
        lib.kmb_destroy(h)
