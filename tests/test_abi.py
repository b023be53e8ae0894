"""CPU-side checks of the drop-in boundary: libisx.so loads and exports exactly what
include/isx.h declares; argument validation works without touching a GPU."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "isx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from isx import _lib
    lib = _lib.lib()
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), "libisx.so does not export %s" % n
    assert sorted(_lib.EXPORTS) == names, "binding table and header disagree"
    assert lib.isx_version() >= 100


def test_argument_validation_without_gpu():
    from isx import _lib
    lib = _lib.lib()
    # bad shapes are rejected on the host before any launch
    assert lib.isx_cosine_sim(None, -1, None, 4, 8, None, None) == -1
    assert b"bad shape" in lib.isx_last_error()
    assert lib.isx_cosine_topk(None, 4, None, 4, 8, 0, 0, None, None, None, 0, None) == -1
    assert lib.isx_region_topk(None, 1, 4, 100, 100, 3, None, None, None) == -1
    assert b"4096" in lib.isx_last_error()
    assert lib.isx_topk_merge(None, None, 8, 4, 1024, None, None, None) == -1
    # round-3 entries: channels-last region kernels, wide stem, DBA groups
    assert lib.isx_boxpool_s1_nhwc(None, 1, 6, 14, 14, 7, 7, None, None) == -1 and b"multiple of 4" in lib.isx_last_error()
    assert lib.isx_boxpool_s1_nhwc(None, 1, 8, 14, 14, 15, 7, None, None) == -1 and b"bad shape" in lib.isx_last_error()
    assert lib.isx_stem7x7_pool_nhwc(None, 1, 8, 900, None, None, None, None) == -1 and b"896" in lib.isx_last_error()
    assert lib.isx_stem7x7_pool_nhwc(None, 0, 8, 448, None, None, None, None) == 0
    assert lib.isx_dba_groups(None, 10, 64, None, None, None, 2000, -1, None, None) == -1 and b"1024" in lib.isx_last_error()
    assert lib.isx_dba_groups(None, 0, 64, None, None, None, 0, -1, None, None) == 0
    assert lib.isx_region_topk_nhwc(None, 1, 4, 100, 100, 3, None, None, None) == -1 and b"4096" in lib.isx_last_error()
    assert lib.isx_region_gather_l2_nhwc(None, 1, 6, 9, 9, 3, 3, None, 2, 7, None, 1e-10, None, None) == -1 and b"multiple of 4" in lib.isx_last_error()
    assert lib.isx_best_location_desc_nhwc(None, 0, 4, 2, 2, 1e-10, None, None, None) == 0
    # round-4 entries: training kernels (backward.hip, head.hip)
    assert lib.isx_conv_wgrad_nhwc(None, None, 4, 1, 7, 7, 100, 64, 1, 1, None, None, None) == -1 and b"multiples of 64" in lib.isx_last_error()
    assert lib.isx_conv_wgrad_nhwc(None, None, 5, 2, 7, 7, 64, 64, 1, 1, None, None, None) == -1 and b"multiple of leaves" in lib.isx_last_error()
    assert lib.isx_conv_wgrad_splits(1176, 512, 2048, 1) == 3 and lib.isx_conv_wgrad_splits(1176, 512, 512, 9) == 2 and lib.isx_conv_wgrad_splits(10, 100, 64, 1) == 0
    assert lib.isx_conv3x3_s2_col2im_nhwc(None, 1, 14, 14, 30, None, None, None) == -1 and b"Cin % 4" in lib.isx_last_error()
    assert lib.isx_conv1x1_dgrad_nhwc(None, 0, 64, None, 64, None, None, None, None) == 0
    assert lib.isx_bn_fold_backward(None, None, 1, 1, None, None, None, None, 64, 64, 5, 0, 0, None, None, None, None) == -1 and b"taps" in lib.isx_last_error()
    assert lib.isx_relu_grad(None, None, 6, None, None) == -1 and b"multiple of 4" in lib.isx_last_error()
    assert lib.isx_head_linear_splits(100352) == 32 and lib.isx_head_linear_splits(640) == 1
    assert lib.isx_head_linear_fwd(None, 24, 60, 100352, None, 2048, None, None, None, 0, None) == -1 and b"Mp % 64" in lib.isx_last_error()
    assert lib.isx_head_linear_dgrad(None, 64, 2048, None, 100, None, None) == -1 and b"K % 64" in lib.isx_last_error()
    assert lib.isx_colsum_leaves(None, 0, 24, 2048, None, None) == 0 and lib.isx_l2norm_rows_bwd(None, None, 0, 16, 1e-10, None, None) == 0
    assert lib.isx_tree_sum_rows(None, 17, 8, 8, None, None) == -1 and b"1 <= L <= 16" in lib.isx_last_error() and lib.isx_tree_sum_rows(None, 4, 8, 0, None, None) == 0
    # round-6 entries: the step's triplet loss in one launch
    assert lib.isx_triplet_leaves(None, 2, 0, 64, 0.1, 1, 1.0, 1.0, None, None, None) == -1 and b"bad shape" in lib.isx_last_error()
    assert lib.isx_triplet_leaves(None, 2, 8, 64, 0.1, 1, 1.0, 1.0, None, None, None) == -1 and b"null pointer" in lib.isx_last_error()
    assert lib.isx_triplet_leaves(None, 0, 8, 64, 0.1, 1, 1.0, 1.0, None, None, None) == 0
    # empty problems are no-ops
    assert lib.isx_l2norm_rows(None, 0, 16, 1e-10, None, None) == 0
    assert lib.isx_cosine_sim(None, 0, None, 0, 8, None, None) == 0
    # workspace queries are pure host arithmetic
    assert lib.isx_cosine_topk_workspace(1000, 10000, 2048, 100) >= 1000 * 100 * 8 + 1000 * 10000 * 4
    assert lib.isx_rank_full_workspace(10, 100) == 256
    assert lib.isx_rank_full_workspace(10, 10000) == 10 * 16384 * 8


def test_ops_refuse_cpu_tensors():
    import torch
    from isx import ops, _lib
    with pytest.raises(_lib.IsxError):
        ops.l2norm_rows(torch.zeros(2, 8))
    with pytest.raises(_lib.IsxError):
        ops.cosine_sim(torch.zeros(2, 8), torch.zeros(3, 8))
