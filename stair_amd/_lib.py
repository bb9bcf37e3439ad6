"""ctypes binding of libstair_hip.so (include/stair_hip.h).

There is NO fallback: if the shared library is missing or a symbol is absent, importing this module
raises, and so does every product entry point that depends on it.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE the library is dlopened: torch ships its own libamdhip64.so (SONAME
# libamdhip64.so.7) and our NEEDED entry then resolves to that already-loaded runtime.  Loading in the
# other order puts two HIP runtimes into one process and the second one finds no device.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('STAIR_LIB_PATH') or os.path.join(_HERE, 'lib', 'libstair_hip.so')   # override: kernel experiments

ABI_VERSION = 6
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class StairConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('hidden_size', 'video_size', 'text_size', 'answer_vocab_length',
                                         'max_video_length', 'object_types', 'have_pretrain_head')]


class GemmArgs(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64), ('a_gstride', C.c_int64), ('a_gidx', C.c_void_p),
                ('W', C.c_void_p), ('ldw', C.c_int64), ('bias', C.c_void_p),
                ('C', C.c_void_p), ('ldc', C.c_int64), ('c_gstride', C.c_int64), ('c_gidx', C.c_void_p),
                ('row_scale', C.c_void_p), ('rs_gstride', C.c_int64), ('rs_gidx', C.c_void_p),
                ('groups', C.c_int32), ('rows_per_group', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('act', C.c_int32), ('accumulate', C.c_int32),
                ('splitk_ws', C.c_void_p), ('splitk_ws_floats', C.c_int64)]


class GemmTnArgs(C.Structure):
    _fields_ = [('A', C.c_void_p), ('lda', C.c_int64),
                ('B', C.c_void_p), ('ldb', C.c_int64), ('b_gstride', C.c_int64), ('b_gidx', C.c_void_p),
                ('row_scale', C.c_void_p), ('rs_gstride', C.c_int64), ('rs_gidx', C.c_void_p),
                ('C', C.c_void_p), ('ldc', C.c_int64),
                ('M', C.c_int32), ('rows_per_group', C.c_int32), ('N', C.c_int32), ('K', C.c_int32),
                ('colsum', C.c_void_p), ('colsum2', C.c_void_p), ('b_is_bf16', C.c_int32)]


class GemmPlanesArgs(C.Structure):
    _fields_ = [('A_hi', C.c_void_p), ('A_lo', C.c_void_p), ('lda', C.c_int64),
                ('W_hi', C.c_void_p), ('W_lo', C.c_void_p), ('ldw', C.c_int64),
                ('bias', C.c_void_p), ('C', C.c_void_p), ('ldc', C.c_int64),
                ('M', C.c_int32), ('N', C.c_int32), ('K', C.c_int32), ('act', C.c_int32), ('w_tiled', C.c_int32)]


class LstmArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ldx', C.c_int64), ('rows', C.c_int32), ('n', C.c_int32),
                ('max_len', C.c_int32), ('I', C.c_int32), ('Hh', C.c_int32), ('seq_off', C.c_void_p),
                ('w_ih', C.c_void_p * 2), ('w_hh', C.c_void_p * 2), ('b_ih', C.c_void_p * 2), ('b_hh', C.c_void_p * 2),
                ('xproj_ws', C.c_void_p), ('bias_ws', C.c_void_p), ('whh_pack_ws', C.c_void_p),
                ('out', C.c_void_p), ('ldo', C.c_int64), ('h_n', C.c_void_p), ('cbuf', C.c_void_p),
                ('x_bf16', C.c_void_p), ('wih_planes_ws', C.c_void_p), ('coop_ws', C.c_void_p), ('coop_ws_bytes', C.c_int64),
                ('seq_len', C.c_void_p), ('status', C.c_void_p), ('x_planes_ws', C.c_void_p)]


class LstmBwdArgs(C.Structure):
    _fields_ = [('x', C.c_void_p), ('ldx', C.c_int64), ('rows', C.c_int32), ('n', C.c_int32),
                ('max_len', C.c_int32), ('I', C.c_int32), ('Hh', C.c_int32), ('seq_off', C.c_void_p),
                ('w_hh', C.c_void_p * 2),
                ('gates', C.c_void_p), ('cbuf', C.c_void_p), ('out', C.c_void_p), ('ldo', C.c_int64),
                ('d_out', C.c_void_p), ('ldd', C.c_int64), ('d_hn', C.c_void_p),
                ('whh_pack_ws', C.c_void_p), ('hprev_ws', C.c_void_p),
                ('dw_ih', C.c_void_p * 2), ('dw_hh', C.c_void_p * 2), ('db_ih', C.c_void_p * 2), ('db_hh', C.c_void_p * 2),
                ('x_bf16', C.c_void_p), ('seq_len', C.c_void_p), ('coop_ws', C.c_void_p), ('coop_ws_bytes', C.c_int64),
                ('status', C.c_void_p), ('tn_ws', C.c_void_p), ('tn_ws_floats', C.c_int64)]


class TileMlpArgs(C.Structure):
    _fields_ = [('X', C.c_void_p), ('x_gstride', C.c_int64), ('x_idx', C.c_void_p),
                ('row_scale', C.c_void_p), ('rs_idx', C.c_void_p),
                ('W', C.c_void_p * 3), ('bias', C.c_void_p * 3), ('act', C.c_int32 * 3), ('n_layers', C.c_int32),
                ('save', C.c_void_p * 3),
                ('mid_rowdot', C.c_int32), ('vw', C.c_void_p), ('vb', C.c_void_p), ('extra', C.c_void_p), ('rs_out', C.c_void_p),
                ('tail', C.c_int32),
                ('out', C.c_void_p), ('out_gstride', C.c_int64), ('out_idx', C.c_void_p),
                ('gamma', C.c_void_p), ('beta', C.c_void_p), ('ln_eps', C.c_float),
                ('kb', C.c_void_p), ('pair_first', C.c_void_p), ('pair_cnt', C.c_void_p), ('att_idx', C.c_void_p), ('att', C.c_void_p),
                ('len', C.c_void_p),
                ('cnt', C.c_int32), ('T', C.c_int32), ('H', C.c_int32),
                ('act_mask', C.c_void_p * 3), ('act_scale', C.c_float),
                ('in_mask', C.c_void_p), ('in_mask_gstride', C.c_int64), ('in_mask_idx', C.c_void_p), ('in_scale', C.c_float),
                ('x_broadcast', C.c_int32), ('save_in', C.c_void_p),
                ('vec_pack', C.c_int32), ('pk_a', C.c_void_p), ('pk_b', C.c_void_p), ('pk_a_idx', C.c_void_p), ('pk_b_idx', C.c_void_p),
                ('vec_cnt', C.c_int32), ('cat_save', C.c_void_p), ('out_row_idx', C.c_void_p),
                ('ln_bwd', C.c_int32), ('dgamma', C.c_void_p), ('dbeta', C.c_void_p),
                ('adj_feat', C.c_void_p), ('adj_feat_gstride', C.c_int64), ('adj_feat_idx', C.c_void_p),
                ('adj_rs', C.c_void_p), ('adj_rs_idx', C.c_void_p), ('adj_drs', C.c_void_p), ('acc_exclusive', C.c_int32),
                ('save_bits', C.c_void_p * 3), ('act_bits', C.c_void_p * 3), ('in_bits', C.c_void_p),
                ('drop_site', C.c_uint32 * 3), ('drop_p', C.c_float), ('drop_seed', C.c_uint64)]


class VecProblem(C.Structure):
    _fields_ = [('kind', C.c_int32), ('rows', C.c_int32),
                ('a', C.c_void_p), ('b', C.c_void_p), ('ia', C.c_void_p), ('ib', C.c_void_p), ('lda', C.c_int64), ('ldb', C.c_int64),
                ('pack', C.c_int32), ('in_scale', C.c_float), ('kred', C.c_int32),
                ('W', C.c_void_p), ('ldw', C.c_int64), ('bias', C.c_void_p), ('N', C.c_int32), ('act', C.c_int32), ('wplanes', C.c_void_p),
                ('emask', C.c_void_p), ('ldm', C.c_int64), ('escale', C.c_float),
                ('out', C.c_void_p), ('io', C.c_void_p), ('ldo', C.c_int64), ('accumulate', C.c_int32),
                ('in_save', C.c_void_p), ('ld_save', C.c_int64),
                ('adj', C.c_int32), ('fa', C.c_void_p), ('fb', C.c_void_p), ('fia', C.c_void_p), ('fib', C.c_void_p),
                ('ldfa', C.c_int64), ('ldfb', C.c_int64), ('ga', C.c_void_p), ('gb', C.c_void_p), ('gia', C.c_void_p), ('gib', C.c_void_p),
                ('drop_site', C.c_uint32), ('drop_p', C.c_float), ('drop_seed', C.c_uint64)]


class PlanInfo(C.Structure):
    _fields_ = [('workspace_bytes', C.c_int64), ('vec_off', C.c_int64), ('map_off', C.c_int64), ('att_off', C.c_int64),
                ('tok_off', C.c_int64), ('qfeat_off', C.c_int64), ('logits_off', C.c_int64),
                ('gvec_off', C.c_int64), ('gmap_off', C.c_int64), ('gatt_off', C.c_int64), ('status_off', C.c_int64),
                ('n_vec', C.c_int32), ('n_map', C.c_int32), ('n_att', C.c_int32), ('n_tok_rows', C.c_int32),
                ('n_nodes', C.c_int32), ('n_launches', C.c_int32), ('n_levels', C.c_int32), ('n_questions', C.c_int32),
                ('T', C.c_int32), ('n_aliased', C.c_int32), ('n_vec_stage', C.c_int32), ('n_map_stage', C.c_int32), ('n_att_stage', C.c_int32)]


# every symbol include/stair_hip.h declares: (name, restype, argtypes)
SIGNATURES = [
    ('stair_abi_version', C.c_int, []),
    ('stair_last_error', C.c_char_p, []),
    ('stair_acct_enable', None, [C.c_int32]),
    ('stair_acct_dump', C.c_int, [C.c_char_p, C.c_int32]),
    ('stair_ctx_create', C.c_int, [C.POINTER(StairConfig), C.POINTER(C.c_void_p)]),
    ('stair_ctx_destroy', None, [C.c_void_p]),
    ('stair_weight_count', C.c_int, [C.c_void_p]),
    ('stair_weight_name', C.c_char_p, [C.c_void_p, C.c_int]),
    ('stair_weight_numel', C.c_int64, [C.c_void_p, C.c_int]),
    ('stair_ctx_set_weight', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ('stair_ctx_set_grad', C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64]),
    ('stair_set_matmul_mode', C.c_int, [C.c_int32]),
    ('stair_get_matmul_mode', C.c_int, []),
    ('stair_set_split_min_rows', C.c_int, [C.c_int32]),
    ('stair_lstm_coop_limit', C.c_int, [C.c_int32]),
    ('stair_vec_group', C.c_int, [C.POINTER(VecProblem), C.c_int32, C.c_void_p]),
    ('stair_set_tile_mlp', C.c_int, [C.c_int32]),
    ('stair_ctx_set_option', C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    ('stair_ctx_get_option', C.c_int, [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]),
    ('stair_set_tile_queue', C.c_int, [C.c_int32]),
    ('stair_tile_mlp_fwd', C.c_int, [C.POINTER(TileMlpArgs), C.c_void_p]),
    ('stair_pack_wfrag', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_pack_wfrag_ld', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_tile_timing', C.c_int, [C.c_int32]),
    ('stair_tile_timing_read', C.c_int, [C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    ('stair_gemm_f32', C.c_int, [C.POINTER(GemmArgs), C.c_void_p]),
    ('stair_gemm_tn_f32', C.c_int, [C.POINTER(GemmTnArgs), C.c_void_p]),
    ('stair_gemm_tn_slabs_scratch', C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    ('stair_set_tn_slab_min_rows', C.c_int, [C.c_int32]),
    ('stair_gemm_tn_slabs', C.c_int, [C.POINTER(GemmTnArgs), C.c_void_p, C.c_int64, C.c_void_p]),
    ('stair_split_planes', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ('stair_split_planes_tiled', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_gemm_planes', C.c_int, [C.POINTER(GemmPlanesArgs), C.c_void_p]),
    ('stair_lstm_coop_ws_bytes', C.c_int64, [C.c_int32]),
    ('stair_lstm_coop_bwd_ws_bytes', C.c_int64, [C.c_int32]),
    ('stair_lstm_bidir_fwd', C.c_int, [C.POINTER(LstmArgs), C.c_void_p]),
    ('stair_lstm_bidir_bwd', C.c_int, [C.POINTER(LstmBwdArgs), C.c_void_p]),
    ('stair_cosine_attn_fwd', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_temporal_relate_fwd', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p),
                                            C.c_void_p]),
    ('stair_cosine_attn_bwd', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_temporal_relate_bwd', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                            C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_void_p]),
    ('stair_cosine_topk', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    ('stair_l2normalize_fwd', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_plan_build', C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                   C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ('stair_plan_build_shared', C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                          C.c_int32, c_int32_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ('stair_plan_build_ragged', C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p,
                                          C.c_int32, c_int32_p, c_int32_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ('stair_comm_unique_id', C.c_int, [C.c_void_p]),
    ('stair_comm_create', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    ('stair_comm_destroy', None, [C.c_void_p]),
    ('stair_comm_info', C.c_int, [C.c_void_p, c_int32_p, c_int32_p]),
    ('stair_allreduce_grads', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ('stair_mfma_probe', C.c_int, [C.c_int32, C.c_int32, C.POINTER(C.c_double), C.c_void_p]),
    ('stair_loss_groups', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32]),
    ('stair_grad_shadows_begin', C.c_int, [C.c_void_p, C.c_void_p]),
    ('stair_loss_decoder_ce', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_score_cosine_to_mean', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_loss_attention_len', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                           C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_plan_backward', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                      C.c_float, C.c_void_p, C.c_int32, C.c_void_p]),
    ('stair_loss_filterframe', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_loss_filterframe_len', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_plan_zero_grads', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('stair_loss_attention', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_loss_head', C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_loss_contrastive', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                         C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_loss_contrastive_table', C.c_int, [C.c_void_p] * 7 + [C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p]),
    ('stair_plan_regions', C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.c_int32]),
    ('stair_plan_touched', C.c_int, [C.c_void_p, C.c_void_p, c_int32_p, C.c_int32]),
    ('stair_adam_step', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int64, C.c_void_p, C.c_void_p]),
    ('stair_plan_destroy', None, [C.c_void_p]),
    ('stair_plan_get_info', C.c_int, [C.c_void_p, C.POINTER(PlanInfo)]),
    ('stair_plan_status', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    ('stair_debug_memset', C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    ('stair_debug_queue_probe', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    ('stair_plan_node', C.c_int, [C.c_void_p, C.c_int32, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p]),
    ('stair_plan_saved_offset', C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_int64)]),
    ('stair_plan_nodes', C.c_int, [C.c_void_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, c_int32_p, C.c_int32]),
    ('stair_plan_run', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    ('stair_dropout_fwd', C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint64, C.c_uint32,
                                   C.c_void_p]),
    ('stair_plan_set_dropout', C.c_int, [C.c_void_p, C.c_float, C.c_uint64]),
    ('stair_plan_set_backward_event', C.c_int, [C.c_void_p, C.c_void_p]),
    ('stair_plan_upload', C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    ('stair_projection_floats', C.c_int64, [C.c_void_p, C.c_int32, C.c_int32, C.c_int64]),
    ('stair_encoders_project', C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64,
                                          C.c_void_p]),
    ('stair_plan_set_projection', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    ('stair_plan_run_flags', C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p,
                                       C.c_void_p, C.c_int32, C.c_void_p]),
]


class StairError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'libstair_hip.so not found at %s -- build it with `python -c "import __graft_entry__ as g; g.build()"` '
            'or `make -C stair_amd/csrc`. There is no CPU fallback for the product path.' % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, restype, argtypes in SIGNATURES:
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.stair_abi_version() != ABI_VERSION:
        raise ImportError('libstair_hip.so ABI version %d, expected %d' % (lib.stair_abi_version(), ABI_VERSION))
    return lib


lib = _load()


def check(rc):
    if rc != 0:
        raise StairError(lib.stair_last_error().decode('utf-8', 'replace'))
