"""ctypes binding of libkmbart_hip.so (include/kmbart.h).  Fails loudly when the library is missing:
there is no CPU fallback for the hot path."""
import ctypes as C
import os

import torch  # noqa: F401  -- must come first: torch bundles its own HIP runtime (libamdhip64.so.7); loading ours
#                              before it would map a second runtime from /opt/rocm and break stream sharing

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("KMB_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libkmbart_hip.so")   # KMB_LIB_PATH: a diagnostic build of the same library

c_p = C.c_void_p
i32, i64, u32, f32, f64 = C.c_int32, C.c_int64, C.c_uint32, C.c_float, C.c_double


class KmbConfig(C.Structure):
    _fields_ = [(n, i32) for n in (
        "vocab_size", "d_model", "encoder_layers", "decoder_layers", "encoder_attention_heads",
        "decoder_attention_heads", "encoder_ffn_dim", "decoder_ffn_dim", "max_position_embeddings",
        "extra_pos_embeddings", "image_feature_size", "pad_token_id", "bos_token_id", "eos_token_id",
        "img_feat_id", "cls_token_id", "scale_embedding")] + [
        ("dropout", f32), ("attention_dropout", f32), ("activation_dropout", f32), ("layer_norm_eps", f32),
        ("num_labels", i32), ("num_attributes", i32), ("num_relations", i32)]


class KmbBatch(C.Structure):
    _fields_ = [("B", i32), ("S", i32), ("T", i32),
                ("input_ids", c_p), ("attention_mask", c_p), ("image_features", c_p), ("feat_offsets", c_p),
                ("n_features", i32), ("decoder_input_ids", c_p), ("decoder_attention_mask", c_p), ("labels", c_p)]


class KmbPretrain(C.Structure):
    _fields_ = [("n_mrm", i32), ("mrm_rows", c_p), ("mrm_targets", c_p),
                ("n_attr", i32), ("attr_rows", c_p), ("attr_labels", c_p),
                ("n_rel", i32), ("rel_obj_rows", c_p), ("rel_subj_rows", c_p), ("rel_labels", c_p),
                ("lm_factor", f32), ("mrm_factor", f32), ("attr_factor", f32), ("rel_factor", f32),
                ("losses_out", c_p)]


class KmbForwardOpts(C.Structure):
    _fields_ = [("encoder_states", c_p), ("decoder_states_out", c_p), ("skip_head", i32)]


class KmbGemm(C.Structure):
    _fields_ = [("A", c_p), ("B", c_p), ("lda", i32), ("ldb", i32), ("a_kc", i32), ("b_kc", i32),
                ("M", i32), ("N", i32), ("K", i32), ("bias", c_p), ("col_scale", f32), ("col_scale_n", i32),
                ("act", i32), ("preact", c_p), ("ld_preact", i32), ("aux", c_p), ("ld_aux", i32),
                ("drop_thr16", u32), ("drop_seed", u32), ("drop_scale", f32), ("residual", c_p), ("ld_res", i32),
                ("out_bf16", c_p), ("ld_out_bf16", i32), ("out_f32", c_p), ("ld_out_f32", i32), ("beta", f32),
                ("split_k", i32), ("slab", c_p), ("colsum", c_p), ("tile_order", i32),
                ("row_shift", c_p), ("row_sums", c_p), ("row_sums_ld", i32), ("pick_col", c_p), ("pick_out", c_p)]


class KmbAttn(C.Structure):
    _fields_ = [("Q", c_p), ("K", c_p), ("V", c_p), ("ldq", i32), ("ldk", i32), ("ldv", i32),
                ("B", i32), ("H", i32), ("Tq", i32), ("Tk", i32), ("key_mask", c_p), ("causal", i32),
                ("O", c_p), ("ldo", i32), ("lse", c_p), ("dO", c_p), ("lddo", i32),
                ("dQ", c_p), ("dK", c_p), ("dV", c_p), ("lddq", i32), ("lddk", i32), ("lddv", i32),
                ("dq_scale", f32), ("dq_colsum", c_p), ("dk_colsum", c_p), ("dv_colsum", c_p), ("ld_colsum", i32)]


class KmbAttnDecode(C.Structure):
    _fields_ = [("Q", c_p), ("ldq", i32), ("Kc", c_p), ("Vc", c_p), ("Tmax", i32), ("ldc", i32), ("kv_row", c_p),
                ("key_mask", c_p), ("mask_ld", i32), ("mask_row", c_p), ("R", i32), ("H", i32), ("Tk", i32),
                ("O", c_p), ("ldo", i32), ("new_k", c_p), ("new_v", c_p), ("ld_new", i32), ("Kw", c_p), ("Vw", c_p), ("hist", c_p)]


class KmbDecodeBlock(C.Structure):
    _fields_ = [("kind", i32), ("in_", c_p), ("ld_in", i32), ("gamma", c_p), ("beta", c_p), ("eps", f32), ("ln_out", c_p),
                ("W", c_p), ("bias", c_p), ("R", i32), ("K", i32), ("N", i32), ("act", i32), ("residual", c_p),
                ("ld_res", i32), ("out", c_p), ("ld_out", i32), ("H", i32), ("q_scale", f32), ("Kc", c_p), ("Vc", c_p),
                ("Tmax", i32), ("ldc", i32), ("Tk", i32), ("kv_row", c_p), ("key_mask", c_p), ("mask_ld", i32), ("hist", c_p), ("kv_group", i32)]


class KmbDrop(C.Structure):
    _fields_ = [("thr16", u32), ("seed", u32), ("scale", f32)]


class KmbAdamW(C.Structure):
    _fields_ = [("lr", f64), ("beta1", f64), ("beta2", f64), ("eps", f64), ("weight_decay", f64),
                ("step", i32), ("correct_bias", i32), ("grad_scale", f32)]


class KmbAllreduceOpts(C.Structure):
    _fields_ = [("algo", i32), ("after_compute", i32), ("max_piece_elems", i64), ("adamw", C.POINTER(KmbAdamW))]


class KmbCommPiece(C.Structure):
    _fields_ = [("bucket", i32), ("repad_piece", i32), ("repad_shard", i32), ("reserved", i32),
                ("offset", i64), ("count", i64), ("shard", i64), ("mine", i64)]


# name -> (restype, argtypes); must list every function include/kmbart.h declares
PROTOTYPES = {
    "kmb_last_error": (C.c_char_p, []),
    "kmb_version": (C.c_int, []),
    "kmb_create": (C.c_int, [C.POINTER(KmbConfig), C.POINTER(c_p)]),
    "kmb_destroy": (None, [c_p]),
    "kmb_param_count": (C.c_int, [c_p]),
    "kmb_param_info": (C.c_int, [c_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i64), C.POINTER(i32), C.POINTER(i32)]),
    "kmb_arena_elems": (i64, [c_p]),
    "kmb_bf16_arena_elems": (i64, [c_p]),
    "kmb_bind_arenas": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "kmb_workspace_bytes": (i64, [c_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kmb_bind_workspace": (C.c_int, [c_p, c_p, i64]),
    "kmb_sync_params": (C.c_int, [c_p, c_p]),
    "kmb_set_seed": (C.c_int, [c_p, C.c_uint64]),
    "kmb_bucket_count": (C.c_int, [c_p]),
    "kmb_bucket_range": (C.c_int, [c_p, C.c_int, C.POINTER(i64), C.POINTER(i64)]),
    "kmb_stream_wait_bucket": (C.c_int, [c_p, C.c_int, c_p]),
    "kmb_logits_ld": (C.c_int, [c_p]),
    "kmb_forward": (C.c_int, [c_p, C.POINTER(KmbBatch), C.c_int, C.c_int, c_p, c_p, c_p, c_p]),
    "kmb_forward_ex": (C.c_int, [c_p, C.POINTER(KmbBatch), C.POINTER(KmbForwardOpts), C.c_int, C.c_int, c_p, c_p, c_p, c_p]),
    "kmb_last_logits": (C.c_int, [c_p, c_p, c_p]),
    "kmb_hidden_state": (C.c_int, [c_p, C.c_int, C.c_int, c_p, c_p]),
    "kmb_attention_probs": (C.c_int, [c_p, C.c_int, C.c_int, c_p, c_p]),
    "kmb_set_precision": (C.c_int, [c_p, C.c_int]),
    "kmb_act_bytes": (C.c_int, [c_p]),
    "kmb_reserve_head_rows": (C.c_int, [c_p, C.c_int]),
    "kmb_forward_pretrain": (C.c_int, [c_p, C.POINTER(KmbBatch), C.POINTER(KmbPretrain), C.c_int, C.c_int, c_p, c_p, c_p]),
    "kmb_forward_pretrain_ex": (C.c_int, [c_p, C.POINTER(KmbBatch), C.POINTER(KmbPretrain), C.POINTER(KmbForwardOpts), C.c_int, C.c_int,
                                          c_p, c_p, c_p]),
    "kmb_backward": (C.c_int, [c_p, f32, c_p]),
    "kmb_backward_dev": (C.c_int, [c_p, c_p, c_p]),
    "kmb_adamw_step": (C.c_int, [c_p, C.POINTER(KmbAdamW), i64, i64, c_p]),
    "kmb_read_status": (C.c_int, [c_p, C.POINTER(i32), c_p]),
    "kmb_read_status_async": (C.c_int, [c_p, c_p, c_p]),
    "kmb_gen_begin": (C.c_int, [c_p, C.POINTER(KmbBatch), C.c_int, C.c_int, c_p]),
    "kmb_gen_encoder_states": (C.c_int, [c_p, c_p, c_p]),
    "kmb_gen_step": (C.c_int, [c_p, c_p, C.c_int, c_p, c_p]),
    "kmb_gen_reorder": (C.c_int, [c_p, c_p, C.c_int, c_p]),
    "kmb_gen_last_hidden": (C.c_int, [c_p, c_p, c_p]),
    "kmb_beam_merge": (C.c_int, [c_p, c_p, C.c_int, C.c_int, C.c_int, C.c_int, c_p, c_p]),
    "kmb_beam_merge_select": (C.c_int, [c_p, c_p, C.c_int, C.c_int, C.c_int, C.c_int, c_p, C.c_int, c_p, c_p, c_p, c_p]),
    "kmb_beam_step": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, c_p, c_p, c_p,
                                c_p, C.c_int64, c_p]),
    "kmb_gen_beam_step": (C.c_int, [c_p, c_p, C.c_int, C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, c_p, c_p, c_p,
                                    c_p, C.c_int64, C.c_int, c_p]),
    "kmb_beam_step_stats": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, c_p, c_p,
                                      c_p, c_p, C.c_int, c_p]),
    "kmb_logsoftmax_topk": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, c_p, c_p]),
    "kmb_logsoftmax_topk_scratch": (C.c_int64, [C.c_int]),
    "kmb_logsoftmax_topk_ws": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, C.c_int, C.c_int, c_p, c_p, c_p,
                                         C.c_int64, c_p]),
    "kmb_gen_workspace_bytes": (i64, [c_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int]),
    "kmb_gemm_shared_device": (C.c_int, [C.c_int]),
    "kmb_comm_unique_id": (C.c_int, [c_p]),
    "kmb_comm_init": (C.c_int, [c_p, C.c_int, C.c_int, c_p]),
    "kmb_comm_destroy": (C.c_int, [c_p]),
    "kmb_encoder_states_grad": (C.c_int, [c_p, c_p, c_p]),
    "kmb_comm_info": (C.c_int, [c_p, C.POINTER(i32), C.POINTER(i32)]),
    "kmb_comm_broadcast_params": (C.c_int, [c_p, C.c_int, c_p]),
    "kmb_allreduce_grads": (C.c_int, [c_p, C.POINTER(KmbAllreduceOpts), c_p]),
    "kmb_comm_wait": (C.c_int, [c_p, c_p]),
    "kmb_comm_gather_moments": (C.c_int, [c_p, c_p]),
    "kmb_comm_pieces": (i64, [c_p, i64]),
    "kmb_comm_plan": (C.c_int, [c_p, C.c_int, C.c_int, i64, i64, C.POINTER(KmbCommPiece)]),
    "kmb_comm_moments_sharded": (C.c_int, [c_p]),
    "kmb_debug_trace": (C.c_int, [C.c_int]),
    "kmb_debug_trace_dump": (C.c_int, [C.c_char_p]),
    "kmb_set_side_stream": (C.c_int, [c_p, C.c_int]),
    "kmb_profile_gemm": (C.c_int, [C.c_int]),
    "kmb_profile_read": (C.c_int, [C.c_int, C.POINTER(i64), C.POINTER(f64), C.POINTER(f64)]),
    "kmb_profile_dump": (C.c_int, [C.c_char_p]),
    "kmb_clock_stamp": (C.c_int, [c_p, c_p]),
    "kmb_pack_features": (C.c_int, [c_p, c_p, C.c_int32, C.c_int32, c_p, c_p]),
    "kmb_op_gemm": (C.c_int, [C.POINTER(KmbGemm), c_p]),
    "kmb_op_gemm_allrows": (C.c_int, [C.POINTER(KmbGemm), c_p]),
    "kmb_op_gemm_allrows_stats_floats": (C.c_int64, [C.c_int]),
    "kmb_op_gemm_allrows_stats": (C.c_int, [C.POINTER(KmbGemm), c_p, c_p]),
    "kmb_op_gemm_group": (C.c_int, [C.POINTER(KmbGemm), C.c_int32, c_p]),
    "kmb_op_attn_fwd": (C.c_int, [C.POINTER(KmbAttn), c_p]),
    "kmb_op_attn_bwd": (C.c_int, [C.POINTER(KmbAttn), c_p]),
    "kmb_op_attn_decode": (C.c_int, [C.POINTER(KmbAttnDecode), c_p]),
    "kmb_op_decode_block": (C.c_int, [C.POINTER(KmbDecodeBlock), c_p]),
    "kmb_op_decode_pack": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, c_p, c_p]),
    "kmb_op_ln_fwd": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, C.c_int, C.c_int, f32, c_p]),
    "kmb_op_ln_bwd_scratch": (i64, [C.c_int, C.c_int]),
    "kmb_op_ln_bwd": (C.c_int, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, C.POINTER(KmbDrop), C.POINTER(KmbDrop), c_p, c_p,
                                c_p, C.c_int, C.c_int, c_p]),
    "kmb_op_colsum_scratch": (i64, [C.c_int, C.c_int]),
    "kmb_op_colsum": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, c_p, c_p, c_p]),
    "kmb_op_img_rowmap": (C.c_int, [c_p, c_p, C.c_int, C.c_int, i64, i64, c_p, c_p, c_p]),
    "kmb_op_cast_pad": (C.c_int, [c_p, C.c_int, C.c_int, c_p, C.c_int, c_p]),
    "kmb_op_embed_ln_fwd": (C.c_int, [c_p, c_p, c_p, c_p, c_p, C.c_int, C.c_int, f32, c_p, c_p, c_p, c_p, c_p, c_p,
                                      C.c_int, C.c_int, f32, C.POINTER(KmbDrop), c_p]),
    "kmb_op_embed_bwd": (C.c_int, [c_p, c_p, c_p, f32, c_p, c_p, i64, C.c_int, C.c_int, c_p]),
    "kmb_op_pos_bwd": (C.c_int, [c_p, C.c_int, C.c_int, C.c_int, c_p, C.c_int, C.c_int, c_p]),
    "kmb_op_ce": (C.c_int, [c_p, C.c_int, C.c_int, c_p, C.c_int, f32, c_p, c_p, c_p, c_p, c_p]),
    "kmb_op_adamw": (C.c_int, [c_p, c_p, c_p, c_p, c_p, i64, C.POINTER(KmbAdamW), c_p]),
    "kmb_op_cast_bf16": (C.c_int, [c_p, c_p, i64, c_p]),
    "kmb_op_dropout_mask": (C.c_int, [u32, f32, C.c_int, C.c_int, c_p, c_p]),
}

_lib = None


class KmbError(RuntimeError):
    pass


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KmbError(
            "libkmbart_hip.so is missing (%s). Build it with `python km-bart_amd/build.py`; "
            "the KM-BART hot path has no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc != 0:
        raise KmbError(load().kmb_last_error().decode("utf-8", "replace"))


def ptr(t):
    """Device (or host) address of a torch tensor, or NULL."""
    return None if t is None else C.c_void_p(t.data_ptr())
