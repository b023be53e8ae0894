"""ctypes loader of libisx.so (the C ABI of include/isx.h).

The library is built in-tree by instance-search_amd/csrc/Makefile (hipcc, gfx950).
torch is imported first so that libisx resolves libamdhip64.so.7 to the HIP runtime
torch already loaded (one runtime per process)."""
import ctypes as C
import os

import torch  # noqa: F401  (must precede the dlopen below)

_CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "csrc")
LIB_PATH = os.environ.get("ISX_LIB") or os.path.join(_CSRC, "libisx.so")      # ISX_LIB: A/B builds of the same ABI

_lib = None

I64, I32, F32, SZ, VP = C.c_int64, C.c_int, C.c_float, C.c_size_t, C.c_void_p

_SIGNATURES = {
    "isx_version": (C.c_int, []),
    "isx_last_error": (C.c_char_p, []),
    "isx_l2norm_rows": (C.c_int, [VP, I64, I64, F32, VP, VP]),
    "isx_l2norm_rows_bwd": (C.c_int, [VP, VP, I64, I64, F32, VP, VP]),
    "isx_l2norm_shift_rows": (C.c_int, [VP, VP, I64, I64, F32, VP, VP]),
    "isx_gap_l2": (C.c_int, [VP, I64, I32, I32, I32, F32, VP, VP]),
    "isx_gap_l2_nhwc": (C.c_int, [VP, I64, I32, I32, I32, F32, VP, VP]),
    "isx_bias_act_inplace": (C.c_int, [VP, VP, VP, I64, I32, I64, I32, VP]),
    "isx_images_u8_to_f32": (C.c_int, [VP, I64, I32, I32, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, I32, VP, VP]),
    "isx_bias_relu_maxpool_nhwc": (C.c_int, [VP, VP, I64, I32, I32, I32, VP, VP]),
    "isx_conv1x1_nhwc": (C.c_int, [VP, I64, I32, VP, I32, VP, VP, I32, VP, VP]),
    "isx_conv1x1_dual_nhwc": (C.c_int, [VP, I32, VP, I64, I32, I32, I32, I32, VP, I32, VP, I32, VP, VP]),
    "isx_stem7x7_pool_nhwc": (C.c_int, [VP, I64, I32, I32, VP, VP, VP, VP]),
    "isx_conv3x3_expand_nhwc": (C.c_int, [VP, I64, I32, I32, I32, VP, VP, I32, VP, I32, VP, VP, I32, VP, VP]),
    "isx_conv3x3_expand_dual_nhwc": (C.c_int, [VP, I64, I32, I32, I32, VP, VP, VP, VP, I32, VP, I32, VP, VP]),
    "isx_conv3x3_nhwc": (C.c_int, [VP, I64, I32, I32, I32, VP, I32, I32, VP, VP, I32, VP, VP]),
    "isx_boxpool_s1": (C.c_int, [VP, I64, I32, I32, I32, I32, I32, VP, VP]),
    "isx_boxpool_s1_nhwc": (C.c_int, [VP, I64, I32, I32, I32, I32, I32, VP, VP]),
    "isx_best_location_desc": (C.c_int, [VP, I64, I32, I32, I32, F32, VP, VP, VP]),
    "isx_best_location_desc_nhwc": (C.c_int, [VP, I64, I32, I32, I32, F32, VP, VP, VP]),
    "isx_region_topk_nhwc": (C.c_int, [VP, I64, I32, I32, I32, I32, VP, VP, VP]),
    "isx_region_gather_l2_nhwc": (C.c_int, [VP, I64, I32, I32, I32, I32, I32, VP, I32, I32, VP, F32, VP, VP]),
    "isx_region_topk": (C.c_int, [VP, I64, I32, I32, I32, I32, VP, VP, VP]),
    "isx_region_gather_l2": (C.c_int, [VP, I64, I32, I32, I32, I32, I32, VP, I32, I32, VP, F32, VP, VP]),
    "isx_cosine_sim": (C.c_int, [VP, I64, VP, I64, I32, VP, VP]),
    "isx_cosine_topk_workspace": (SZ, [I64, I64, I32, I32]),
    "isx_cosine_topk": (C.c_int, [VP, I64, VP, I64, I32, I32, I64, VP, VP, VP, SZ, VP]),
    "isx_rows_to_f16": (C.c_int, [VP, I64, I32, VP, VP, VP, VP]),
    "isx_cosine_sim_f16": (C.c_int, [VP, I64, VP, I64, I32, VP, VP]),
    "isx_gallery_to_f16": (C.c_int, [VP, I64, I32, VP, VP, VP]),
    "isx_cosine_topk_fast_workspace": (SZ, [I64, I64, I32, I32, I32]),
    "isx_cosine_topk_fast_fallback_offset": (SZ, [I64, I64, I32, I32, I32]),
    "isx_cosine_topk_fast": (C.c_int, [VP, I64, VP, I64, I32, I32, I64, VP, VP, VP, VP, VP, SZ, VP]),
    "isx_topk_rows": (C.c_int, [VP, I64, I64, I32, I64, VP, VP, VP]),
    "isx_rank_full_workspace": (SZ, [I64, I64]),
    "isx_rank_full": (C.c_int, [VP, I64, I64, VP, VP, SZ, VP]),
    "isx_average_precision": (C.c_int, [VP, I64, I64, VP, VP, I32, VP, VP]),
    "isx_average_precision_sim": (C.c_int, [VP, I64, I64, VP, VP, I32, VP, VP]),
    "isx_masked_sums": (C.c_int, [VP, I64, I64, VP, VP, VP, VP]),
    "isx_dba_groups": (C.c_int, [VP, I64, I32, VP, VP, VP, I32, I32, VP, VP]),
    "isx_topk_merge": (C.c_int, [VP, VP, I32, I64, I32, VP, VP, VP]),
    "isx_mine_negatives": (C.c_int, [VP, I64, VP, VP, VP, I64, I32, VP, VP]),
    "isx_mine_negatives_rows": (C.c_int, [VP, I64, I64, I64, VP, VP, VP, I64, I32, VP, VP]),
    "isx_triplet_loss_fwd": (C.c_int, [VP, VP, VP, I64, I32, F32, I32, VP, VP]),
    "isx_triplet_loss_bwd": (C.c_int, [VP, VP, VP, VP, I64, I32, F32, I32, VP, VP, VP, VP]),
    "isx_triplet_loss_bwd_dev": (C.c_int, [VP, VP, VP, VP, I64, I32, F32, VP, I32, VP, VP, VP, VP]),
    "isx_conv1x1_dgrad_nhwc": (C.c_int, [VP, I64, I32, VP, I32, VP, VP, VP, VP]),
    "isx_conv3x3_dgrad_nhwc": (C.c_int, [VP, I64, I32, I32, I32, VP, I32, VP, VP, VP]),
    "isx_conv3x3_s2_col2im_nhwc": (C.c_int, [VP, I64, I32, I32, I32, VP, VP, VP]),
    "isx_conv_wgrad_splits": (C.c_int, [I64, I32, I32, I32]),
    "isx_conv_wgrad_nhwc": (C.c_int, [VP, VP, I64, I32, I32, I32, I32, I32, I32, I32, VP, VP, VP]),
    "isx_relu_grad": (C.c_int, [VP, VP, I64, VP, VP]),
    "isx_bn_fold_backward": (C.c_int, [VP, VP, I32, I32, VP, VP, VP, VP, I32, I32, I32, I32, I64, VP, VP, VP, VP]),
    "isx_head_linear_splits": (C.c_int, [I64]),
    "isx_head_linear_fwd": (C.c_int, [VP, I64, I64, I64, VP, I32, VP, VP, VP, SZ, VP]),
    "isx_head_linear_dgrad": (C.c_int, [VP, I64, I32, VP, I64, VP, VP]),
    "isx_head_linear_rows_workspace": (SZ, [I64, I64, I32]),
    "isx_head_linear_fwd_rows": (C.c_int, [VP, I64, I64, VP, I32, VP, VP, VP, SZ, VP]),
    "isx_head_groups": (C.c_int, [I64]),
    "isx_head_linear_dgrad_parts": (C.c_int, [VP, I64, I32, I32, VP, I64, VP, VP]),
    "isx_head_sgd_step": (C.c_int, [VP, VP, I64, I32, I64, VP, VP, I32, F32, F32, F32, F32, I32, VP]),
    "isx_colsum_leaves": (C.c_int, [VP, I32, I32, I64, VP, VP]),
    "isx_tree_sum_rows": (C.c_int, [VP, I32, I64, I64, VP, VP]),
    "isx_ap_shard_max_positives": (C.c_int, []),
    "isx_ap_shard_positives": (C.c_int, [VP, I64, I64, I64, VP, VP, I32, VP, VP, VP]),
    "isx_ap_shard_hist": (C.c_int, [VP, I64, I64, I64, VP, I32, VP, VP]),
    "isx_ap_from_hist": (C.c_int, [VP, I32, VP, I64, I32, VP, VP]),
    "isx_triplet_leaves": (C.c_int, [VP, I32, I32, I32, C.c_float, I32, C.c_float, C.c_float, VP, VP, VP]),
    "isx_comm_unique_id_bytes": (C.c_int, []),
    "isx_comm_unique_id": (C.c_int, [VP]),
    "isx_comm_init_rank": (C.c_int, [C.POINTER(VP), I32, I32, VP]),
    "isx_comm_destroy": (C.c_int, [VP]),
    "isx_shard_topk_allgather": (C.c_int, [VP, VP, VP, I64, I32, VP, VP, VP]),
    "isx_comm_allgather_rows": (C.c_int, [VP, VP, I64, I64, VP, VP]),
}

EXPORTS = tuple(sorted(_SIGNATURES))


class IsxError(RuntimeError):
    pass


def lib():
    """The loaded library; raises (never falls back) when libisx.so is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise IsxError("libisx.so not found at %s -- run `make -C %s` (or __graft_entry__.build()); "
                           "there is no CPU fallback" % (LIB_PATH, _CSRC))
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            f = getattr(h, name)
            f.restype, f.argtypes = res, args
        _lib = h
    return _lib


def check(rc, what):
    if rc != 0:
        raise IsxError("%s failed (%d): %s" % (what, rc, lib().isx_last_error().decode()))
