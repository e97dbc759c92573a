"""ctypes binding of libadn.so (the C ABI declared in include/adn.h).

The product path has NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  Build with ``make -C audio-depth-estimation_amd/csrc`` (or
``__graft_entry__.build()``); the .so is kept in-tree next to this file.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('ADN_LIB', os.path.join(_HERE, 'libadn.so'))      # ADN_LIB: A/B builds (tools/ab)

ADN_F32, ADN_BF16 = 0, 1
GEMM_S2, GEMM_T2, GEMM_S1 = 0, 1, 2
EPI_RAW, EPI_Z_STATS, EPI_ACT, EPI_BWD, EPI_FINAL, EPI_ADD = 0, 1, 2, 3, 4, 5

c_void_p, c_int32, c_int64, c_float = C.c_void_p, C.c_int32, C.c_int64, C.c_float


class AdnEpiSeg(C.Structure):
    _fields_ = [
        ('out0', c_void_p), ('out1', c_void_p), ('ref', c_void_p), ('z', c_void_p),
        ('mean', c_void_p), ('istd', c_void_p), ('scale', c_void_p), ('shift', c_void_p),
        ('bias', c_void_p), ('partials', c_void_p),
        ('channels', c_int32), ('slope', c_float), ('accumulate', c_int32), ('final_act', c_int32),
    ]


class AdnIgemmDesc(C.Structure):
    _fields_ = [
        ('dtype', c_int32), ('geom', c_int32), ('B', c_int32), ('Hs', c_int32), ('Ws', c_int32),
        ('C0', c_int32), ('C1', c_int32), ('N', c_int32),
        ('in0', c_void_p), ('in1', c_void_p), ('w', c_void_p),
        ('epi', c_int32), ('seg', AdnEpiSeg * 2),
        ('workspace', c_void_p), ('workspace_bytes', c_int64),
        ('ks', c_int32), ('reserved', c_int32),
    ]


class AdnMx8ConvDesc(C.Structure):
    _fields_ = [
        ('B', c_int32), ('H', c_int32), ('W', c_int32), ('C0', c_int32), ('C1', c_int32), ('N', c_int32),
        ('in0', c_void_p), ('sc0', c_void_p), ('in1', c_void_p), ('sc1', c_void_p), ('w', c_void_p), ('wsc', c_void_p),
        ('epi', c_int32), ('reserved', c_int32), ('seg', AdnEpiSeg * 2),
    ]


class AdnWgradDesc(C.Structure):
    _fields_ = [
        ('dtype', c_int32), ('B', c_int32), ('Hs', c_int32), ('Ws', c_int32),
        ('plain0', c_void_p), ('plain1', c_void_p), ('R0', c_int32), ('R1', c_int32),
        ('gath0', c_void_p), ('gath1', c_void_p), ('C0', c_int32), ('C1', c_int32),
        ('dw', c_void_p), ('workspace', c_void_p), ('workspace_bytes', c_int64), ('c_valid', c_int32),
        ('geom', c_int32), ('ks', c_int32), ('sq_partials', c_void_p),
    ]


class AdnAttnDesc(C.Structure):
    _fields_ = [
        ('dtype', c_int32), ('B2', c_int32), ('N', c_int32), ('dqk', c_int32), ('dv', c_int32), ('kv_shift', c_int32),
        ('q', c_void_p), ('k', c_void_p), ('v', c_void_p),
        ('ld_q', c_int32), ('ld_k', c_int32), ('ld_v', c_int32),
        ('o', c_void_p), ('ld_o', c_int32),
        ('lse', c_void_p), ('scale', c_float),
        ('dout', c_void_p), ('ld_do', c_int32),
        ('dq', c_void_p), ('dk', c_void_p), ('dvp', c_void_p),
        ('ld_dq', c_int32), ('ld_dk', c_int32), ('ld_dv', c_int32),
        ('workspace', c_void_p), ('workspace_bytes', c_int64),
    ]


class AdnDistillSmall(C.Structure):
    _fields_ = [
        ('mean_student', c_void_p), ('mean_teacher', c_void_p), ('centers_student', c_void_p), ('centers_teacher', c_void_p),
        ('feat_stats', c_void_p * 5), ('feat_channels', c_int32 * 5),
        ('pix_stats', c_void_p),
        ('B', c_int32), ('nb', c_int32), ('has_teacher', c_int32),
        ('temperature', c_float), ('lambda_task', c_float), ('lambda_response', c_float), ('lambda_feature', c_float),
        ('lambda_bin', c_float), ('lambda_sparse', c_float),
        ('terms', c_void_p), ('dmean', c_void_p), ('dcent', c_void_p),
    ]


# name -> (restype, argtypes).  Must list every symbol include/adn.h declares
# (tests/test_abi.py cross-checks this table against the header and the built library).
_PROTOS = {
    'adn_last_error': (C.c_char_p, []),
    'adn_version': (C.c_int, []),
    'adn_debug_poison_lds': (C.c_int, [c_void_p]),
    'adn_debug_stream_rmw': (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p]),
    'adn_igemm_num_partials': (c_int64, [C.POINTER(AdnIgemmDesc)]),
    'adn_igemm_workspace_bytes': (c_int64, [C.POINTER(AdnIgemmDesc)]),
    'adn_igemm': (C.c_int, [C.POINTER(AdnIgemmDesc), c_void_p]),
    'adn_wgrad_workspace_bytes': (c_int64, [C.POINTER(AdnWgradDesc)]),
    'adn_wgrad': (C.c_int, [C.POINTER(AdnWgradDesc), c_void_p]),
    'adn_wgrad_batchable': (c_int32, [C.POINTER(AdnWgradDesc)]),
    'adn_wgrad_batch_sq_count': (c_int32, [C.POINTER(AdnWgradDesc)]),
    'adn_wgrad_batch': (C.c_int, [C.POINTER(AdnWgradDesc), c_int32, c_void_p]),
    'adn_wgrad_patch_batch_workspace_bytes': (c_int64, [C.POINTER(AdnWgradDesc), c_int32]),
    'adn_wgrad_patch_batch': (C.c_int, [C.POINTER(AdnWgradDesc), c_int32, c_void_p]),
    'adn_wgrad_sq_count': (c_int32, [C.POINTER(AdnWgradDesc)]),
    'adn_pack_weights': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'adn_mx8_quantize': (C.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p]),
    'adn_mx8_pack': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'adn_bn_partials_reduce': (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_tail_stats_blocks': (c_int64, [c_int64, c_int32]),
    'adn_maxpool2_bwd_tail': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                        c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_upsample2x_bwd_tail': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                          c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_maxpool2_fwd_mx8': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'adn_upsample2x_fwd_mx8': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                         c_void_p, c_void_p]),
    'adn_bn_act_mx8': (C.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_bn_bwd_apply_mx8': (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                       c_void_p, c_void_p]),
    'adn_conv3x3_mx8_num_partials': (c_int64, [C.POINTER(AdnMx8ConvDesc)]),
    'adn_conv3x3_mx8': (C.c_int, [C.POINTER(AdnMx8ConvDesc), c_void_p]),
    'adn_pack_rows': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_pack_transpose_taps': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                          c_void_p]),
    'adn_maxpool2_fwd': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_maxpool2_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_upsample2x_fwd': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_upsample2x_bwd': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_pixel_shuffle2': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_relu_bwd_stats_num_partials': (c_int64, [c_int64, c_int32]),
    'adn_relu_bwd_stats': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_head1x1_fwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p]),
    'adn_head1x1_bwd_workspace_bytes': (c_int64, [c_int64, c_int32]),
    'adn_head1x1_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_void_p, c_void_p,
                                  c_int64, c_void_p]),
    'adn_l1tv_workspace_bytes': (c_int64, [c_int64]),
    'adn_l1tv_stats': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_l1tv_finish': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_void_p, c_int32, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'adn_nchw_to_nhwc': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32,
                                   c_void_p]),
    'adn_nchw_slice_to_nhwc': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_attn_fwd': (C.c_int, [C.POINTER(AdnAttnDesc), c_void_p]),
    'adn_attn_bwd_workspace_bytes': (c_int64, [C.POINTER(AdnAttnDesc)]),
    'adn_attn_bwd': (C.c_int, [C.POINTER(AdnAttnDesc), c_void_p]),
    'adn_channel_sum_workspace_bytes': (c_int64, [c_int64, c_int32]),
    'adn_channel_sum': (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_gate_bwd': (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_int64,
                               c_void_p]),
    'adn_pool_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'adn_pool': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_binpred_fwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_float, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                  c_void_p]),
    'adn_binpred_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_float, c_float, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p,
                                  c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_dropout_mask': (C.c_int, [c_void_p, c_int64, c_float, C.c_uint64, c_void_p, c_void_p]),
    'adn_bcast_add': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_int32, c_int32, c_void_p]),
    'adn_bins_fwd': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_bins_bwd_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32]),
    'adn_bins_bwd': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int32, c_void_p, c_int64, c_void_p]),
    'adn_distill_pix_stats': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_distill_pix_grad': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'adn_featcos_grad': (C.c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p]),
    'adn_distill_small': (C.c_int, [C.POINTER(AdnDistillSmall), c_void_p]),
    'adn_resize_nearest': (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_resize_nearest_bwd': (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_image_prepare': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_depth_prepare': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_float, c_void_p,
                                    c_void_p]),
    'adn_lowpass_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32, c_int32]),
    'adn_lowpass': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_clamp_add': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
    'adn_baseres_stats': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, c_float, c_float, c_float, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_baseres_grad': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_float, c_void_p, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'adn_nhwc_to_nchw': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p]),
    'adn_bn_fwd_finalize': (C.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_float, c_float,
                                      c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                      c_void_p]),
    'adn_bn_fwd_fused': (C.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_float, c_float, c_void_p, c_void_p,
                                   c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_void_p,
                                   c_void_p, c_void_p]),
    'adn_bn_bwd_fused': (C.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32,
                                   c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_bn_eval_affine': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int32, c_void_p, c_void_p,
                                     c_void_p]),
    'adn_bn_act': (C.c_int, [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_float, c_void_p, c_void_p,
                             c_void_p]),
    'adn_bn_bwd_finalize': (C.c_int, [c_void_p, c_int64, c_int32, c_int64, c_void_p, c_void_p, c_void_p, c_void_p]),
    'adn_bn_bwd_apply': (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_void_p]),
    'adn_loss_stats': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_int32, c_float, c_void_p, c_void_p,
                                 c_int64, c_void_p]),
    'adn_loss_workspace_bytes': (c_int64, [c_int64]),
    'adn_loss_finish': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_int32, c_float, c_void_p, c_int32,
                                  c_float, c_float, c_float, c_void_p, c_void_p, c_void_p]),
    'adn_loss_finish_dz': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_int32, c_float, c_void_p, c_int32,
                                     c_float, c_float, c_float, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64,
                                     c_void_p]),
    'adn_final_act_bwd': (C.c_int, [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_convt_n1_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32]),
    'adn_convt_n1_forward': (C.c_int, [c_int32, c_int32, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32,
                                       c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_l0_forward': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_float, c_void_p, c_void_p,
                                 c_void_p]),
    'adn_d0_dgrad_num_partials': (c_int64, [c_int32, c_int32, c_int32]),
    'adn_d0_dgrad': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, C.POINTER(AdnEpiSeg), C.POINTER(AdnEpiSeg),
                               c_void_p]),
    'adn_thin_wgrad_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32, c_int32, c_int32, c_int32]),
    'adn_thin_wgrad': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p,
                                 c_void_p, c_int64, c_void_p]),
    'adn_sum_to_scalar': (C.c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_grad_norm': (C.c_int, [c_void_p, c_int64, c_float, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_grad_norm_workspace_bytes': (c_int64, [c_int64]),
    'adn_grad_sqsum_partials': (C.c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p]),
    'adn_grad_norm_ranges': (C.c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_int32, c_float, c_void_p, c_void_p,
                                       c_int64, c_void_p]),
    'adn_optimizer_step': (C.c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int32, c_float, c_float,
                                     c_float, c_float, c_float, c_int32, c_void_p, c_void_p, c_void_p]),
    'adn_pack_t2_multi': (C.c_int, [c_void_p, c_int32, c_void_p, c_int32, c_int64, c_int32, c_void_p, c_void_p]),
    'adn_edge_loss_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32]),
    'adn_edge_loss': (C.c_int, [c_void_p, c_void_p, c_int32, c_int32, c_int32, c_float, c_float, c_float, c_void_p, c_void_p,
                                c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_compute_errors': (C.c_int, [c_void_p, c_void_p, c_int32, c_int64, c_void_p, c_void_p, c_int64, c_void_p]),
    'adn_compute_errors_workspace_bytes': (c_int64, [c_int32, c_int64]),
    'adn_frontend_workspace_bytes': (c_int64, [c_int32, c_int32, c_int32]),
    'adn_frontend': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p, c_int64,
                               c_void_p]),
    'adn_resize_bilinear': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_resize_bilinear_bwd': (C.c_int, [c_void_p, c_int32, c_int32, c_int32, c_int32, c_void_p, c_void_p]),
    'adn_clamp_range': (C.c_int, [c_void_p, c_void_p, c_int64, c_float, c_void_p, c_void_p]),
}

_lib = None


ABI_VERSION = 3      # adn_version() of the libadn.so these bindings describe (include/adn.h)


def load():
    """Load libadn.so once; raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f'libadn.so not found at {LIB_PATH}: the HIP extension is not built. '
            'Run `make -C audio-depth-estimation_amd/csrc` or `python -c "import __graft_entry__ as g; g.build()"`. '
            'There is no CPU fallback for the product path.')
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.adn_version() != ABI_VERSION:     # the descriptor structs below must match the library's: never mix revisions
        raise RuntimeError(f'{LIB_PATH} is ABI revision {lib.adn_version()}, this package binds revision {ABI_VERSION}: '
                           'rebuild it (`make -C audio-depth-estimation_amd/csrc`)')
    _lib = lib
    return lib


def symbol_names():
    return sorted(_PROTOS)


def check(rc: int, what: str = ''):
    if rc != 0:
        msg = load().adn_last_error().decode('utf-8', 'replace')
        raise RuntimeError(f'libadn {what} failed (rc={rc}): {msg}')


# Launch-plan recording: while RECORD is a list every successful call is appended as (cfunc, args, name) so
# that a fixed-shape step can be replayed with one Python loop over prebuilt ctypes arguments (no descriptor
# rebuilding, ~1.5 us per launch).  Python-side actions (collectives) are recorded with record_py().
RECORD = None


_POISON = bool(os.environ.get('ADN_LDS_POISON'))      # debugging aid: see adn_debug_poison_lds in include/adn.h


def call(name: str, *args):
    """Call an int-returning entry point and raise on error."""
    fn = getattr(load(), name)
    if _POISON:
        import torch
        check(load().adn_debug_poison_lds(C.c_void_p(torch.cuda.current_stream().cuda_stream)), 'adn_debug_poison_lds')
    check(fn(*args), name)
    if RECORD is not None:
        RECORD.append((fn, args, name, {}))


def annotate(**meta):
    """Attach metadata (e.g. algorithmic FLOPs) to the launch that was just recorded."""
    if RECORD:
        RECORD[-1][3].update(meta)


def record_py(fn):
    """Run a Python action now and, while recording, make it part of the launch plan (as ONE entry: library calls the
    action makes itself -- the reducer's per-bucket norm -- are not recorded a second time)."""
    global RECORD
    saved, RECORD = RECORD, None
    try:
        fn()
    finally:
        RECORD = saved
    if RECORD is not None:
        RECORD.append((None, fn, 'py', {}))


def replay(plan):
    for fn, args, name, _ in plan:
        if fn is None:
            args()
        else:
            rc = fn(*args)
            if rc != 0:
                check(rc, name)


def ptr(t):
    """Device/host pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()
