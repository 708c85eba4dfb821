"""ctypes binding of libdmt_hip.so (the C ABI declared in include/dmt_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (see DESIGN.md "Failure behaviour").
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libdmt_hip.so")

DMT_F32, DMT_BF16, DMT_FP8_E4M3 = 0, 1, 2
DMT_OPT_SGD, DMT_OPT_ADAGRAD, DMT_OPT_ADADELTA, DMT_OPT_RMSPROP, DMT_OPT_FTRL = 1, 2, 3, 4, 5
DMT_MAX_FEATURES, DMT_MAX_SEQS, DMT_MAX_TABLES = 32, 4, 32
DMT_SEQ_TARGET = 100
DMT_ERR_UNSUPPORTED = -3
DMT_ABI_VERSION = 6        # include/dmt_hip.h: the revision this binding was written against (checked by load())

c_i32, c_i64, c_f32, c_vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GatherFeature(C.Structure):
    _fields_ = [("table", c_vp), ("rows", c_i32), ("dim", c_i32), ("idx", c_vp), ("wts", c_vp), ("lens", c_vp),
                ("T", c_i32), ("pooled_off", c_i32), ("seq_id", c_i32), ("seq_off", c_i32), ("group", c_i32),
                ("inv_wsum", c_vp), ("row_stride", c_i32), ("idx_seq", c_vp)]


class GatherDesc(C.Structure):
    _fields_ = [("B", c_i32), ("n_features", c_i32), ("feat", GatherFeature * DMT_MAX_FEATURES), ("n_seq", c_i32),
                ("seq_out", c_vp * DMT_MAX_SEQS), ("seq_T", c_i32 * DMT_MAX_SEQS), ("pos", c_vp * DMT_MAX_SEQS),
                ("tar_out", c_vp), ("d_model", c_i32), ("seq_scale", c_f32), ("pooled", c_vp), ("ld_pooled", c_i64),
                ("dense", c_vp), ("n_dense", c_i32), ("out_dtype", c_i32),
                ("seq_drop_seed", C.c_uint32 * DMT_MAX_SEQS), ("seq_drop_keep", c_f32),
                ("seq_row_off", c_vp * DMT_MAX_SEQS), ("seq_row_len", c_vp * DMT_MAX_SEQS)]      # revision 5: packed rows


class EmbGradDesc(C.Structure):
    _fields_ = [("B", c_i32), ("n_features", c_i32), ("feat", GatherFeature * DMT_MAX_FEATURES),
                ("row_base", c_i32 * DMT_MAX_FEATURES), ("entry_base", c_i32 * (DMT_MAX_FEATURES + 1)),
                ("total_rows", c_i32), ("dseq", c_vp * DMT_MAX_SEQS), ("seq_T", c_i32 * DMT_MAX_SEQS), ("dtar", c_vp),
                ("dpooled", c_vp), ("ld_pooled", c_i64), ("d_model", c_i32), ("seq_scale", c_f32), ("grad_dtype", c_i32),
                ("seq_drop_seed", C.c_uint32 * DMT_MAX_SEQS), ("seq_drop_keep", c_f32),
                ("seq_row_off", c_vp * DMT_MAX_SEQS), ("seq_row_len", c_vp * DMT_MAX_SEQS)]      # revision 5: packed rows


class GemmDesc(C.Structure):
    _fields_ = [("in_dtype", c_i32), ("out_dtype", c_i32), ("M", c_i32), ("N", c_i32), ("K", c_i32),
                ("A", c_vp), ("a_rs", c_i64), ("a_cs", c_i64), ("B", c_vp), ("b_rs", c_i64), ("b_cs", c_i64),
                ("C", c_vp), ("ldc", c_i64), ("bias", c_vp), ("act_ncols", c_i32), ("gate", c_vp), ("ldg", c_i64),
                ("resid", c_vp), ("ldr", c_i64), ("a_ones_row", c_i32), ("c_last", c_vp), ("split_k", c_i32), ("accumulate", c_i32),
                ("batch", c_i32), ("a_bs", c_i64), ("b_bs", c_i64), ("c_bs", c_i64), ("bias_bs", c_i64),
                ("gate_bs", c_i64), ("resid_bs", c_i64), ("clast_bs", c_i64)]


class AttnDesc(C.Structure):
    _fields_ = [("dtype", c_i32), ("B", c_i32), ("H", c_i32), ("dh", c_i32), ("Tq", c_i32), ("Tk", c_i32),
                ("Q", c_vp), ("q_bs", c_i64), ("q_rs", c_i64), ("K", c_vp), ("k_bs", c_i64), ("k_rs", c_i64),
                ("V", c_vp), ("v_bs", c_i64), ("v_rs", c_i64), ("q_lens", c_vp), ("k_lens", c_vp),
                ("resid", c_vp), ("r_bs", c_i64), ("r_rs", c_i64), ("out", c_vp), ("o_bs", c_i64), ("o_rs", c_i64),
                ("drop_seed", C.c_uint32), ("drop_keep", c_f32), ("mma_dtype", c_i32),
                ("row_off", c_vp), ("ex_list", c_vp), ("n_list", c_i32), ("max_len", c_i32)]      # revision 5: packed rows (dmt_attn_bwd)


class AttnBwdDesc(C.Structure):
    _fields_ = [("f", AttnDesc), ("dout", c_vp), ("do_bs", c_i64), ("do_rs", c_i64), ("dQ", c_vp), ("dq_bs", c_i64),
                ("dq_rs", c_i64), ("dK", c_vp), ("dk_bs", c_i64), ("dk_rs", c_i64), ("dV", c_vp), ("dv_bs", c_i64),
                ("dv_rs", c_i64)]


class MmoeDesc(C.Structure):
    _fields_ = [("B", c_i32), ("E", c_i32), ("T", c_i32), ("u0", c_i32), ("u1", c_i32), ("u2", c_i32), ("g1", c_vp), ("ldg", c_i64),
                ("w1t", c_vp), ("w1t_expert_stride", c_i64), ("w1t_ld", c_i64), ("w2t", c_vp), ("w2t_expert_stride", c_i64), ("w2t_ld", c_i64),
                ("w1", c_vp), ("w1_expert_stride", c_i64), ("w2", c_vp), ("w2_expert_stride", c_i64),
                ("b1", c_vp), ("b1_expert_stride", c_i64), ("b2", c_vp), ("b2_expert_stride", c_i64),
                ("h1", c_vp), ("h2", c_vp), ("gates", c_vp), ("mix", c_vp), ("dmix", c_vp), ("dh1", c_vp), ("dh2", c_vp), ("dg1", c_vp), ("lddg", c_i64),
                ("ws", c_vp), ("ws_bytes", c_i64), ("gate_dx", c_i32)]


class HeadsDesc(C.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("u_in", c_i32), ("u_fc", c_i32), ("b_in", c_i32), ("b_h0", c_i32), ("b_h1", c_i32),
                ("mix", c_vp), ("zb", c_vp), ("ld_zb", c_i64), ("fc_w", c_vp * 2), ("fc_b", c_vp * 2), ("out_w", c_vp * 2), ("out_b", c_vp * 2),
                ("bias_w", c_vp * 3), ("bias_b", c_vp * 3), ("drop_seed", C.c_uint32 * 2), ("drop_keep", c_f32 * 2),
                ("logits", c_vp), ("h_fc", c_vp), ("h0", c_vp), ("h1", c_vp), ("dlogits", c_vp), ("dmix", c_vp), ("dzb", c_vp), ("ld_dzb", c_i64),
                ("dz_fc", c_vp), ("dz0", c_vp), ("dz1", c_vp), ("g_out_w", c_vp * 2), ("g_out_b", c_vp * 2), ("g_bias_w2", c_vp), ("g_bias_b2", c_vp)]


class Q1memDesc(C.Structure):
    _fields_ = [("B", c_i32), ("T", c_i32), ("H", c_i32), ("d", c_i32), ("dh", c_i32), ("mem", c_vp), ("m_bs", c_i64), ("m_rs", c_i64),
                ("k_lens", c_vp), ("qp", c_vp), ("ctx", c_vp), ("ctx_hs", c_i64), ("drop_seed", C.c_uint32), ("drop_keep", c_f32),
                ("dctx", c_vp), ("dout", c_vp), ("do_bs", c_i64), ("bv", c_vp), ("dqp", c_vp), ("dmem", c_vp), ("dm_bs", c_i64), ("dm_rs", c_i64),
                ("row_off", c_vp)]                                                                # revision 5: packed rows


class CastJob(C.Structure):
    _fields_ = [("src", c_vp), ("dst_plain", c_vp), ("dst_t", c_vp), ("ld_src", c_i64), ("ld_plain", c_i64), ("ld_t", c_i64),
                ("rows", c_i32), ("cols", c_i32), ("tile_begin", c_i32), ("tiles_x", c_i32)]


class ChainDesc(C.Structure):
    _fields_ = [("mode", c_i32), ("kin", c_i32), ("nmid", c_i32), ("nout", c_i32), ("M", c_i64), ("in_", c_vp), ("ld_in", c_i64),
                ("image", c_vp), ("bias2", c_vp), ("gamma", c_vp), ("beta", c_vp), ("eps", c_f32), ("s_out", c_vp), ("y_out", c_vp),
                ("ld_out", c_i64), ("stats", c_vp), ("mid_out", c_vp), ("ld_mid", c_i64), ("mask", c_vp)]


DMT_CHAIN_FFN_LN, DMT_CHAIN_FFN_BWD = 0, 1


class MhsaDesc(C.Structure):
    _fields_ = [("d_model", c_i32), ("num_heads", c_i32), ("B", c_i32), ("T", c_i32), ("x", c_vp), ("lens", c_vp), ("image", c_vp),
                ("bias", c_vp), ("gamma", c_vp), ("beta", c_vp), ("eps", c_f32), ("qkv", c_vp), ("s_out", c_vp), ("y_out", c_vp),
                ("stats", c_vp), ("drop_seed", C.c_uint32), ("drop_keep", c_f32),
                ("blocks", c_vp), ("n_tiles", c_i32), ("n_rows", c_i64)]                          # revision 5: packed rows


class MhsaBwdDesc(C.Structure):
    _fields_ = [("d_model", c_i32), ("num_heads", c_i32), ("B", c_i32), ("T", c_i32), ("ds", c_vp), ("qkv", c_vp), ("lens", c_vp), ("image", c_vp),
                ("dqkv", c_vp), ("dx", c_vp), ("drop_seed", C.c_uint32), ("drop_keep", c_f32), ("blocks", c_vp), ("n_tiles", c_i32), ("n_rows", c_i64)]


class WgradDesc(C.Structure):
    _fields_ = [("A", c_vp), ("ld_a", c_i64), ("a_cols", c_i32), ("B", c_vp), ("ld_b", c_i64), ("M", c_i64), ("N", c_i32), ("C", c_vp),
                ("ldc", c_i64), ("transposed", c_i32), ("bias", c_vp), ("bias_of", c_i32), ("det_ws", c_vp), ("det_ws_bytes", C.c_uint64)]


class LnFinishJob(C.Structure):      # include/dmt_hip.h: dmt_ln_finish_job
    _fields_ = [("partials", C.c_void_p), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("n_part", C.c_int32), ("d", C.c_int32)]


LN_FINISH_MAX = 16


class TableMap(C.Structure):
    _fields_ = [("n_tables", c_i32), ("row_base", c_i32 * (DMT_MAX_TABLES + 1)), ("dim", c_i32 * DMT_MAX_TABLES),
                ("elem_off", c_i64 * DMT_MAX_TABLES), ("shard_w", c_i32), ("shard_r", c_i32)]


_SIGS = {
    "dmt_gather_fwd": [C.POINTER(GatherDesc), c_vp],
    "dmt_embgrad_keys": [C.POINTER(EmbGradDesc), c_vp, c_vp, c_vp],
    "dmt_mhsa_bwd_image_bytes": [C.POINTER(c_i64)],
    "dmt_mhsa_bwd_image_build": [c_vp, c_i64, c_vp, c_vp],
    "dmt_mhsa_block_bwd": [C.POINTER(MhsaBwdDesc), c_vp],
    "dmt_sort_pairs": [c_vp, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp, C.POINTER(C.c_uint64), c_vp],
    "dmt_segment_heads": [c_vp, c_i64, C.c_uint32, c_vp, c_vp, c_vp, c_vp, C.POINTER(C.c_uint64), c_vp],
    "dmt_entry_slots": [C.POINTER(EmbGradDesc), c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp],
    "dmt_embgrad_reduce": [C.POINTER(EmbGradDesc), c_vp, c_vp, c_vp, c_i64, c_vp, c_i32, c_vp, C.c_uint64, c_vp],
    "dmt_rows_reduce": [c_vp, c_vp, c_vp, c_i64, C.c_uint32, c_vp, c_vp, c_i32, c_vp, C.c_uint64, c_vp],
    "dmt_rows_permute": [c_vp, c_vp, c_i64, c_i32, c_i32, c_vp, c_vp],
    "dmt_zero_rows": [c_vp, c_vp, c_i64, c_i64, c_i32, c_vp],
    "dmt_rows_reduce_bf16": [c_vp, c_vp, c_vp, c_i64, C.c_uint32, c_vp, c_vp, c_i32, c_vp, C.c_uint64, c_vp],
    "dmt_gemm": [C.POINTER(GemmDesc), c_vp],
    "dmt_attn_fwd": [C.POINTER(AttnDesc), c_vp],
    "dmt_attn_bwd": [C.POINTER(AttnBwdDesc), c_vp],
    "dmt_attn_long_fwd": [C.POINTER(AttnDesc), c_vp],
    "dmt_q1mem_fwd": [C.POINTER(Q1memDesc), c_vp],
    "dmt_q1mem_bwd": [C.POINTER(Q1memDesc), c_vp],
    "dmt_heads_fwd": [C.POINTER(HeadsDesc), c_vp],
    "dmt_heads_bwd": [C.POINTER(HeadsDesc), c_vp],
    "dmt_mmoe_experts_fwd": [C.POINTER(MmoeDesc), c_vp],
    "dmt_mmoe_experts_bwd": [C.POINTER(MmoeDesc), c_vp],
    "dmt_attn_long_bwd": [C.POINTER(AttnBwdDesc), c_vp],
    "dmt_ln_fwd": [c_i32, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_f32, c_vp, c_i64, c_vp, c_vp],
    "dmt_ln_bwd": [c_i32, c_i64, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp],
    "dmt_ln_bwd_finish_batched": [c_vp, c_i32, c_vp],
    "dmt_mmoe_mix_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp],
    "dmt_mmoe_mix_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp],
    "dmt_scale_add_pos": [c_i32, c_i64, c_i32, c_i32, c_vp, c_f32, c_vp, c_vp, c_vp],
    "dmt_dropout": [c_i32, c_i64, c_vp, c_vp, C.c_uint32, c_f32, c_vp],
    "dmt_relu_bwd": [c_i32, c_i64, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "dmt_loss_unbias": [c_i32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_i32, c_i32, c_f32, c_vp, c_vp, c_vp,
                        c_vp, c_vp, c_vp, c_vp],
    "dmt_adam_begin_step": [c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_vp],
    "dmt_adam_end_step": [c_vp, c_f32, c_f32, c_vp],
    "dmt_adam_dense": [c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_vp, c_f32, c_f32, c_f32, c_vp, c_vp],
    "dmt_adam_sparse_rows": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_f32, c_vp,
                             c_vp, c_f32, c_f32, c_f32, c_vp],
    "dmt_adam_sparse_rows_bf16": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_f32, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp],
    "dmt_adam_catchup_rows": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp],
    "dmt_adam_catchup_rows_to": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_f32, c_f32, c_f32, c_i32, c_vp, c_i32, c_vp],
    "dmt_rows_stamp": [C.POINTER(TableMap), c_vp, c_vp, c_i32, c_vp, c_i32, c_vp],
    "dmt_adam_flush_rows": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_vp],
    "dmt_adam_rebase": [c_vp, c_vp, c_i64, c_vp],
    "dmt_opt_dense": [c_i32, c_i64, c_vp, c_vp, c_vp, c_vp, c_f32, c_f32, c_f32, c_f32, c_f32, c_vp, c_vp],
    "dmt_opt_sparse_rows": [c_i32, C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_vp, c_i32, c_i32, c_f32, c_i32, c_f32, c_f32,
                            c_f32, c_f32, c_vp],
    "dmt_opt_flush_rows": [c_i32, C.POINTER(TableMap), c_vp, c_vp, c_vp, c_vp, c_i32, c_f32, c_f32, c_f32, c_f32, c_vp],
    "dmt_rows_gather": [C.POINTER(TableMap), c_vp, c_vp, c_i64, c_vp, c_i32, c_vp],
    "dmt_cast_bf16": [c_i64, c_vp, c_vp, c_vp],
    "dmt_cast_transpose_bf16": [c_i32, c_i32, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp],
    "dmt_cast_transpose_bf16_batched": [c_i32, c_vp, c_i32, c_vp],
    "dmt_softmax_fwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_i64, c_vp, c_vp, c_f32, C.c_uint32, c_f32, c_vp, c_i32, c_vp],
    "dmt_softmax_bwd": [c_i32, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_f32, C.c_uint32, c_f32, c_i32, c_vp],
    "dmt_colsum": [c_i32, c_i64, c_i64, c_vp, c_i64, c_f32, c_vp, c_i32, c_vp],
    "dmt_colsum_drop": [c_i32, c_i64, c_i64, c_vp, c_f32, c_vp, C.c_uint32, c_f32, c_i32, c_vp],
    "dmt_gemm_dw_batched": [C.POINTER(GemmDesc), c_i32, c_vp],
    "dmt_colsum_rows_packed": [c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_f32, c_vp, C.c_uint32, c_f32, c_i32, c_vp],
    "dmt_auc_hist": [c_i32, c_vp, c_vp, c_i32, c_vp, c_vp],
    "dmt_confusion_counts": [c_i32, c_vp, c_vp, c_f32, c_vp, c_vp],
    "dmt_l2_unique_rows": [c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp],
    "dmt_l2_unique_rows_count": [c_i32, c_i32, c_vp, c_vp, c_vp, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp],
    "dmt_l2_rows_add": [C.POINTER(TableMap), c_vp, c_vp, c_vp, c_i32, c_vp, c_vp, c_vp, c_i32, c_vp],
    "dmt_chain_image_bytes": [c_i32, c_i32, c_i32, C.POINTER(c_i64)],
    "dmt_chain_image_build": [c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp],
    "dmt_chain2": [C.POINTER(ChainDesc), c_vp],
    "dmt_proj_image_bytes": [c_i32, c_i32, C.POINTER(c_i64)],
    "dmt_proj_image_build": [c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp],
    "dmt_proj": [c_i32, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp],
    "dmt_chain_image_job": [c_i32, c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp],
    "dmt_proj_image_job": [c_i32, c_i32, c_vp, c_i64, c_i64, c_vp, c_vp, c_vp],
    "dmt_image_build_batched": [c_i32, c_vp, c_vp],
    "dmt_wgrad320": [C.POINTER(WgradDesc), c_vp],
    "dmt_mhsa_image_bytes": [C.POINTER(c_i64)],
    "dmt_mhsa_image_build": [c_vp, c_i64, c_vp, c_vp],
    "dmt_mhsa_block_fwd": [C.POINTER(MhsaDesc), c_vp],
}

EXPORTED_SYMBOLS = sorted(list(_SIGS.keys()) + ["dmt_last_error", "dmt_version", "dmt_build_arch", "dmt_ln_bwd_partials", "dmt_struct_size", "dmt_chain_supported", "dmt_proj_supported", "dmt_image_job_bytes",
                                                 "dmt_attn_long_supported", "dmt_mmoe_experts_supported", "dmt_mmoe_experts_ws_bytes", "dmt_heads_supported", "dmt_q1mem_supported", "dmt_reduce_det_ws_bytes", "dmt_wgrad320_det_ws_bytes", "dmt_route_trace", "dmt_route_count", "dmt_route_dump"])

_lib = None


def load():
    """Load libdmt_hip.so; fail loudly when it is missing (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libdmt_hip.so is missing (%s). Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C cikm2020_dmt_amd/csrc`. There is no CPU fallback." % LIB_PATH)
    # torch FIRST: its wheel carries its own libamdhip64.  Loaded after torch, this library's HIP dependency resolves to that copy (one
    # runtime in the process); loaded BEFORE torch it pulls in /opt/rocm's copy, torch then brings its own, and this library's launches
    # go to a runtime that holds none of torch's state ("no ROCm-capable device is detected" -- seen with build() and smoke() in one process)
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = c_i32
    lib.dmt_last_error.restype = C.c_char_p
    lib.dmt_last_error.argtypes = []
    lib.dmt_build_arch.restype = C.c_char_p
    lib.dmt_version.restype = c_i32
    lib.dmt_version.argtypes = []
    if lib.dmt_version() != DMT_ABI_VERSION:
        raise RuntimeError("libdmt_hip.so reports ABI revision %d, this binding was written against %d (include/dmt_hip.h: DMT_ABI_VERSION); "
                           "rebuild the library (make -C cikm2020_dmt_amd/csrc)" % (lib.dmt_version(), DMT_ABI_VERSION))
    lib.dmt_ln_bwd_partials.restype = c_i32
    lib.dmt_ln_bwd_partials.argtypes = [c_i64]
    lib.dmt_chain_supported.restype = c_i32
    lib.dmt_chain_supported.argtypes = [c_i32, c_i32, c_i32]
    lib.dmt_proj_supported.restype = c_i32
    lib.dmt_proj_supported.argtypes = [c_i32, c_i32]
    lib.dmt_image_job_bytes.restype = c_i32
    lib.dmt_image_job_bytes.argtypes = []
    lib.dmt_q1mem_supported.restype = c_i32
    lib.dmt_q1mem_supported.argtypes = [c_i32] * 4
    lib.dmt_heads_supported.restype = c_i32
    lib.dmt_heads_supported.argtypes = [c_i32] * 6
    lib.dmt_mmoe_experts_supported.restype = c_i32
    lib.dmt_mmoe_experts_supported.argtypes = [c_i32] * 5
    lib.dmt_mmoe_experts_ws_bytes.restype = c_i64
    lib.dmt_mmoe_experts_ws_bytes.argtypes = [c_i32]
    lib.dmt_reduce_det_ws_bytes.restype = C.c_uint64
    lib.dmt_reduce_det_ws_bytes.argtypes = [c_i64, c_i32]
    lib.dmt_attn_long_supported.restype = c_i32
    lib.dmt_attn_long_supported.argtypes = [c_i32, c_i32, c_i32, c_i32]
    lib.dmt_wgrad320_det_ws_bytes.restype = C.c_uint64
    lib.dmt_wgrad320_det_ws_bytes.argtypes = [c_i64, c_i32]
    lib.dmt_route_trace.restype = c_i32
    lib.dmt_route_trace.argtypes = [c_i32]
    lib.dmt_route_count.restype = c_i64
    lib.dmt_route_count.argtypes = [C.c_char_p]
    lib.dmt_route_dump.restype = c_i32
    lib.dmt_route_dump.argtypes = [C.c_char_p, c_i32]
    _lib = lib
    return lib


class route_trace:
    """with route_trace() as rt: ...; rt.counts -> {route label: launches} of the kernels launched inside (diagnostic; tests)."""

    def __enter__(self):
        load().dmt_route_trace(1)
        self.counts = {}
        return self

    def __exit__(self, *a):
        lib = load()
        lib.dmt_route_trace(0)
        buf = C.create_string_buffer(16384)
        lib.dmt_route_dump(buf, 16384)
        for line in buf.value.decode().splitlines():
            k, _, v = line.rpartition("=")
            self.counts[k] = int(v)
        return False


class DmtError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().dmt_last_error().decode("utf-8", "replace")
        raise DmtError("%s failed (%d): %s" % (what or "libdmt_hip call", rc, msg))


def call(name: str, *args):
    check(getattr(load(), name)(*args), name)
