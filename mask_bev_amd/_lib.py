"""ctypes binding of ``libmaskbev_hip.so`` (declared in ``include/maskbev_hip.h``).

The product path has no CPU fallback: if the library is missing or a kernel call fails, this raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_double, c_float, c_int32, c_int64, c_size_t, c_void_p
from typing import Dict, List, Tuple

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG_DIR, 'libmaskbev_hip.so')

ABI_VERSION = 58


class MaskBevHipError(RuntimeError):
    pass


_P = c_void_p
_F = c_float
_I = c_int32
_L = c_int64

# symbol -> (restype, argtypes); mirrors include/maskbev_hip.h one to one
SIGNATURES: Dict[str, Tuple[object, List[object]]] = {
    'mbv_abi_version': (ctypes.c_int, []),
    'mbv_voxelize_workspace_bytes': (c_size_t, [_L, _I, _L]),
    'mbv_voxelize': (ctypes.c_int, [_P, _I, _L, _P, _I, _F, _F, _F, _F, _F, _F, _F, _F, _F, _I, _I, _I, _I, _I, _I, _L,
                                    _P, _P, _P, _P, _P, _P, _P, c_size_t, _P]),
    'mbv_gather_voxels': (ctypes.c_int, [_P, _I, _P, _L, _I, _P, _P]),
    'mbv_pfn_decorate': (ctypes.c_int, [_P, _I, _P, _P, _P, _P, _L, _I, _F, _F, _F, _F, _F, _F, _P, _P, _P]),
    'mbv_pfn_stats': (ctypes.c_int, [_P, _P, _P, _P, _P, _L, _I, _I, _P, _P]),
    'mbv_pfn_bn_finalize': (ctypes.c_int, [_P, c_double, _P, _P, _F, _F, _I, _P, _P, _I, _P, _P, _P, _P, _P]),
    'mbv_pfn_apply_max': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P, _P]),
    'mbv_pfn_bwd_route': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _L, _I, _I, _P, _P, _P]),
    'mbv_pfn_bwd_bn': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_double, _I, _P, _P, _L, _I, _I, _P, _P]),
    'mbv_scatter_layernorm_workspace_bytes': (c_size_t, [_I]),
    'mbv_scatter_layernorm_patch_supported': (ctypes.c_int, [_I, _I, _I, _I]),
    'mbv_scatter_layernorm_fwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, c_size_t,
                                                 _P, _P, _P]),
    'mbv_scatter_layernorm_fwd2': (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _P, _P, _P, c_size_t,
                                                  _P, _P, _P, _P]),
    'mbv_scatter_layernorm_bwd': (ctypes.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _P, _P, _P, _I, _P,
                                                 c_size_t, _P, _P, _P]),
    'mbv_scatter_layernorm_bwd_adamw': (ctypes.c_int, [_P, _I, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _L, _P, _P, _P, _P, _P,
                                                       _P, _P, _I, _F, _F, _F, _F, _F, _L, _I, _P, c_size_t, _P, _P, _P]),
    'mbv_msda_prepare_supported': (ctypes.c_int, [_I, _I]),
    'mbv_pfn_forward_layout': (ctypes.c_int64, [_L, _L, _P, _I, _P]),
    'mbv_pfn_forward': (ctypes.c_int, [_P, _I, _P, _P, _P, _L, _L, _I, _P, _P, _P, _P, _P, _P, _I, _F, _F, _I, _P, _L, _P]),
    'mbv_skinny_gemm_f32_addrows': (ctypes.c_int, [_P, _P, _P, _L, _I, _I, _I, _P, _P, _P]),
    'mbv_skinny_gemm_f32_supported': (ctypes.c_int, [_L, _I, _I]),
    'mbv_skinny_gemm_f32': (ctypes.c_int, [_P, _P, _P, _L, _I, _I, _I, _I, _P]),
    'mbv_msda_prepare_fwd': (ctypes.c_int, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_msda_prepare_fwd_ld': (ctypes.c_int, [_P, _L, _P, _L, _P, _P, _I, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_msda_prepare_bwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_msda_prepare_bwd_ld': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _L, _P, _L, _P]),
    'mbv_ms_deform_attn_fwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mbv_ms_deform_attn_fwd_v': (ctypes.c_int, [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mbv_ms_deform_attn_bwd_locattn': (ctypes.c_int, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_ms_deform_attn_bwd_split': (ctypes.c_int, [_I, _I, _P]),
    'mbv_ms_deform_attn_bwd_value_packed_supported': (ctypes.c_int, [_I, _I, _I, _I, _P]),
    'mbv_ms_deform_attn_bwd_value_packed_workspace_bytes': (c_size_t, [_I, _I, _I, _I]),
    'mbv_ms_deform_attn_bwd_value_packed': (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _L, _P, c_size_t, _P]),
    'mbv_ms_deform_attn_bwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _I,
                                              _P]),
    'mbv_rowchain_max_stages': (ctypes.c_int, []),
    'mbv_rowchain_slots': (ctypes.c_int, []),
    'mbv_rowchain_run': (ctypes.c_int, [_P, _I, _I, _I, _F, _I, _P]),
    'mbv_rowchain_run_split': (ctypes.c_int, [_P, _I, _I, _I, _F, _I, _I, _P]),
    'mbv_transpose_group': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _P]),
    'mbv_copy_group': (ctypes.c_int, [_P, _P, _P, _I, _P]),
    'mbv_fragment_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _P]),
    'mbv_window_attn_lse_elems': (_L, [_I, _I, _I, _I, _I]),
    'mbv_window_attn_fwd': (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_window_attn_bwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    'mbv_point_sample_fwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'mbv_point_sample_bwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _L, _P, _P]),
    'mbv_point_sample_bwd_stack': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _I, _P]),
    'mbv_hungarian': (ctypes.c_int, [_P, _I, _I, _I, _P, _P]),
    'mbv_hungarian_padded': (ctypes.c_int, [_P, _I, _I, _I, _P, _P, _P]),
    'mbv_hungarian_wide_t': (ctypes.c_int, [_P, _I, _I, _I, _P, _P]),
    'mbv_select_uncertain_points': (ctypes.c_int, [_P, _P, _L, _I, _I, _P, _P]),
    'mbv_adamw_step': (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _L, _F, _F, _F, _F, _F, _L, _F, _I, _I, _P, _P, _P, _P]),
    'mbv_grad_nonfinite': (ctypes.c_int, [_P, _L, _P, _P]),
    'mbv_loss_scale_update': (ctypes.c_int, [_P, _P, _P, _F, _F, _I, _P, _P]),
    'mbv_refresh_shadow': (ctypes.c_int, [_P, _P, _I, _L, _P]),
    'mbv_colsum_accum': (ctypes.c_int, [_P, _I, _L, _I, _P, _P]),
    'mbv_act_bwd_colsum': (ctypes.c_int, [_P, _P, _I, _I, _L, _I, _P, _P, _P]),
    'mbv_wgrad_small_f32': (ctypes.c_int, [_P, _P, _I, _I, _I, _P, _P, _P]),
    'mbv_wgrad_small_f32_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _P]),
    'mbv_mask_loss_rows_fwd': (ctypes.c_int, [_P, _P, _L, _I, _P, _P]),
    'mbv_mask_loss_rows_bwd': (ctypes.c_int, [_P, _P, _P, _L, _I, _P, _P]),
    'mbv_dice_bce_reduce': (ctypes.c_int, [_P, _L, _I, _F, _F, _P, _P, _P, _P]),
    'mbv_mask_loss_rows_bwd_coef': (ctypes.c_int, [_P, _P, _P, _P, _I, _P, _I, _L, _I, _I, _P, _P]),
    'mbv_sample_select_uncertain': (ctypes.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _P, _I, _P, _P]),
    'mbv_uniform_points': (ctypes.c_int, [_P, _L, _I, _P, _P]),
    'mbv_add_layernorm_supported': (ctypes.c_int, [_I]),
    'mbv_add_layernorm_bwd_blocks': (_L, [_L, _I]),
    'mbv_add_layernorm_fwd': (ctypes.c_int, [_P, _I, _P, _I, _P, _P, _L, _I, _F, _P, _P, _I, _P, _P, _P]),
    'mbv_add_layernorm_fwd2': (ctypes.c_int, [_P, _I, _P, _I, _L, _P, _P, _L, _I, _F, _P, _P, _I, _P, _I, _P, _P, _P]),
    'mbv_transposed_batch_sum_accum': (ctypes.c_int, [_P, _I, _L, _I, _P, _P]),
    'mbv_add_layernorm_bwd': (ctypes.c_int, [_P, _I, _P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _P, _I, _P, _P,
                                             _I, _P]),
    'mbv_add_layernorm_bwd2': (ctypes.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _P, _I, _P,
                                              _P, _I, _P]),
    'mbv_add_layernorm_bwd3': (ctypes.c_int, [_P, _I, _P, _I, _P, _I, _P, _P, _P, _P, _L, _I, _P, _P, _I, _P, _P, _I, _P,
                                              _P, _I, _P, _P]),
    'mbv_add_layernorm_bwd_direct': (ctypes.c_int, [_L, _I]),
    'mbv_merge_layernorm_supported': (ctypes.c_int, [_I, _I, _I]),
    'mbv_merge_layernorm_fwd': (ctypes.c_int, [_P, _L, _I, _I, _I, _P, _P, _F, _P, _I, _P, _P, _P]),
    'mbv_merge_layernorm_bwd': (ctypes.c_int, [_P, _I, _P, _P, _P, _P, _L, _I, _I, _I, _P, _P, _P, _I, _P, _I, _P]),
    'mbv_colsum_accum_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _P]),
    'mbv_instance_ids': (ctypes.c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'mbv_expand_instance_masks': (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_matched_mask_iou': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_match_cost_terms': (ctypes.c_int, [_P, _L, _I, _I, _I, _P, _P, _P]),
    'mbv_match_cost': (ctypes.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _P]),
    'mbv_match_products_supported': (ctypes.c_int, [_I, _I, _I]),
    'mbv_match_products': (ctypes.c_int, [_P, _P, _L, _I, _I, _I, _I, _P, _P, _P]),
    'mbv_match_cost_split': (ctypes.c_int, [_P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _I, _P, _P]),
    'mbv_cls_loss_fwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P, _P]),
    'mbv_cls_loss_bwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _P, _P]),
    'mbv_packed_mask_words': (_L, [_I, _I]),
    'mbv_pack_binary_masks': (ctypes.c_int, [_P, _L, _I, _I, _P, _P]),
    'mbv_point_sample_packed_fwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    'mbv_mask_logits_fwd': (ctypes.c_int, [_P, _P, _I, _I, _I, _I, _L, _P, _I, _P]),
    'mbv_attn_mask_from_logits': (ctypes.c_int, [_P, _I, _L, _I, _I, _I, _I, _P, _P]),
    'mbv_attn_workspace_bytes': (c_size_t, [_I, _I, _I, _I, _I]),
    'mbv_attn_fwd': (ctypes.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, c_size_t, _P]),
    'mbv_attn_bwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'mbv_attn_fwd_ld': (ctypes.c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, c_size_t, _P]),
    'mbv_gemm16_supported': (ctypes.c_int, [_I, _L, _L, _L]),
    'mbv_gemm16_nt': (ctypes.c_int, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _L, _L, _L, _P]),
    'mbv_gemm16_nt_acc': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _L, _L, _L, _P]),
    'mbv_gemm16_nn_workspace_bytes': (c_size_t, [_L, _L, _I]),
    'mbv_gemm16_nn': (ctypes.c_int, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _L, _L, _L, _P,
                                     c_size_t, _P]),
    'mbv_gemm16_nn_part_rows': (ctypes.c_int64, [_L, _L, _I]),
    'mbv_gemm16_nn_parts': (ctypes.c_int, [_P, _P, _P, _P, _P, c_size_t, _L, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _L,
                                           _L, _L, _P]),
    'mbv_gemm16_tn_workspace_bytes': (c_size_t, [_L, _L, _L]),
    'mbv_gemm16_tn': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _I, _I, _I, _I, _I, _L, _L, _L, _P, c_size_t,
                                     _P]),
    'mbv_attn_bwd_ld': (ctypes.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P, _P, _P, _I, _I, _P]),
    'mbv_msda_query_inputs': (ctypes.c_int, [_P, _P, _L, _L, _I, _I, _P, _P, _P]),
    'mbv_upsample_bilinear_bwd': (ctypes.c_int, [_P, _I, _L, _I, _I, _I, _I, _P, _I, _P]),
    'mbv_groupnorm_supported': (ctypes.c_int, [_I, _I, _I, _I]),
    'mbv_groupnorm_workspace_bytes': (c_size_t, [_L, _I, _I, _I, _I]),
    'mbv_groupnorm_fwd': (ctypes.c_int, [_P, _I, _L, _I, _I, _I, _I, _P, _P, _F, _P, _I, _I, _I, _I, _P, _I, _P, _P, _P,
                                         c_size_t, _P]),
    'mbv_groupnorm_bwd': (ctypes.c_int, [_P, _I, _P, _I, _P, _P, _P, _P, _L, _I, _I, _I, _I, _I, _P, _I, _P, _P, _I, _P,
                                         _P]),
    'mbv_gemm16_tn_group_workspace_bytes': (c_size_t, [_P, _P, _P, _I]),
    'mbv_gemm32s_supported': (ctypes.c_int, [_I, _L, _L, _L]),
    'mbv_f32_absmax_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _I, _P]),
    'mbv_gemm32s_nt': (ctypes.c_int, [_P, _P, _P, _P, _P, _L, _L, _L, _L, _L, _L, _P, _P, _P, _I, _I, _L, _L, _L, _P]),
    'mbv_gemm32s_nn': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _L, _P, _P, _P, _I, _L, _L, _L, _P]),
    'mbv_gemm32s_tn_workspace_bytes': (c_size_t, [_L, _L, _L]),
    'mbv_gemm32s_nn_part_rows': (ctypes.c_int64, [_L, _I]),
    'mbv_gemm32s_nn_act': (ctypes.c_int, [_P, _P, _P, _P, _P, c_size_t, _L, _L, _L, _L, _L, _L, _L, _P, _P, _P, _I, _P]),
    'mbv_gemm32s_tn_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P, c_size_t, _P]),
    'mbv_gemm32s_tn_group_workspace_bytes': (c_size_t, [_P, _P, _P, _I]),
    'mbv_ln_bound_group': (ctypes.c_int, [_P, _P, _P, _P, _I, _P]),
    'mbv_conv_rows': (ctypes.c_int64, [_L, _L, _L]),
    'mbv_conv_pad_rows': (ctypes.c_int, [_P, _P, _L, _L, _L, _L, _I, _P]),
    'mbv_conv_unpad_rows': (ctypes.c_int, [_P, _P, _L, _L, _L, _L, _I, _P]),
    'mbv_conv3x3_gemm32s': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P, _P]),
    'mbv_conv3x3_gemm16': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _I, _I, _P]),
    'mbv_attn_split_supported': (ctypes.c_int, [_I, _I, _I]),
    'mbv_attn_split_fwd_ld': (ctypes.c_int, [_P, _P, _P, _I, _P, _I, _I, _I, _I, _I, _P, _P, _P, c_size_t, _P]),
    'mbv_attn_split_bwd_ld': (ctypes.c_int, [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _I, _P]),
    'mbv_window_attn_split_supported': (ctypes.c_int, [_I, _I, _I]),
    'mbv_window_attn_split_fwd': (ctypes.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P]),
    'mbv_window_attn_split_bwd': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _I, _I,
                                                 _P, _P]),
    'mbv_patch_embed32_supported': (ctypes.c_int, [_L, _L, _L, _L, _L]),
    'mbv_patch_embed32_fwd': (ctypes.c_int, [_P, _P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P]),
    'mbv_patch_embed32_bwd_image': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P]),
    'mbv_patch_embed32_bwd_weight_workspace_bytes': (c_size_t, [_L, _L, _L, _L, _L]),
    'mbv_patch_embed32_bwd_weight': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P, c_size_t, _P]),
    'mbv_gemm32s_tn_acc': (ctypes.c_int, [_P, _P, _P, _L, _L, _L, _L, _L, _P, _P, _P, c_size_t, _P]),
    'mbv_gemm16_tn_group': (ctypes.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P, c_size_t, _P]),
}

_lib = None


class _LibProxy:
    """The bound library.  ``hook`` (None in normal operation) lets bench.py time every C-ABI call of one eager step
    with HIP events on the launch stream: ``hook(name, fn, args)`` must return ``fn(*args)``."""

    def __init__(self, lib: ctypes.CDLL):
        self._lib = lib
        self.hook = None

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        hook = self.__dict__.get('hook')
        if hook is None or not name.startswith('mbv_'):
            return fn
        return lambda *args: hook(name, fn, args)


# What a call's argument list does not say about its algorithmic work (read by workmodel.py under bench.py's hook only):
# e.g. how many DISTINCT source maps / coordinate sets the rows of a point-sampling call share.
WORK_HINT: dict = {}


def load() -> '_LibProxy':
    """Load the shared library, binding every declared symbol; raises if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MaskBevHipError(
            f'{LIB_PATH} is missing. Build it with `python -m mask_bev_amd.build` (needs hipcc). '
            'mask_bev_amd has no CPU fallback.')
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise MaskBevHipError(f'{LIB_PATH} does not export {name}; rebuild it') from e
        fn.restype = res
        fn.argtypes = args
    v = lib.mbv_abi_version()
    if v != ABI_VERSION:
        raise MaskBevHipError(f'libmaskbev_hip.so ABI {v} != expected {ABI_VERSION}; rebuild it')
    _lib = _LibProxy(lib)
    return _lib


def check(rc: int, what: str) -> None:
    if rc == 0:
        return
    if rc < 0:
        names = {-1: 'MBV_ERR_BAD_ARG', -2: 'MBV_ERR_WORKSPACE', -3: 'MBV_ERR_UNSUPPORTED'}
        raise MaskBevHipError(f'{what} failed: {names.get(rc, rc)}')
    raise MaskBevHipError(f'{what} failed: hipError_t {rc}')
