// isx_internal.hpp -- launchers shared between translation units of libisx.
#pragma once
#include "isx_common.hpp"

namespace isx {

// C[m][n] = sum_k Q[m][k] * G[n][k]  (k-ordered fp32 fma chain), C row stride ldc.
int launch_cosine_gemm(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                       hipStream_t st);

// Same GEMM with the filtering epilogue: the scores of a (row, 32-column group) are stored to C only
// when one of them reaches thr[row] (a lower bound of the row's final k-th best score); gflag
// (M, ngrp = ceil(N/32)) bytes say which groups were stored.
int launch_cosine_gemm_filter(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                              const float* thr, uint8_t* gflag, hipStream_t st);

// Per-row running top-k over a score chunk.  sim: (M, Nc) with row stride ld; column j
// of the chunk is gallery row col_base + j.  carry: (M, k) u64 keys (canonical order,
// 0 = empty); read unless `first`, written unless `emit`.  With `emit` the final
// (score, idx_base + index) lists are written instead.
int launch_select(const float* sim, int64_t M, int64_t Nc, int64_t ld, int64_t col_base, int k, uint64_t* carry,
                  bool first, bool emit, int64_t idx_base, float* top_score, int64_t* top_idx, hipStream_t st,
                  float* thr_out = nullptr);

// Running top-k update from a FILTERED chunk: only the flagged 32-column groups are read from sim.  k <= kGroupSelectMaxK.  Updates carry and thr
// (thr[row] = score of the k-th key, -inf while fewer than k).
int launch_select_groups(const float* sim, const uint8_t* gflag, int64_t M, int64_t Nc, int64_t ld, int64_t col_base, int k,
                         uint64_t* carry, float* thr, bool emit, int64_t idx_base, float* top_score, int64_t* top_idx,
                         hipStream_t st);

constexpr int kGroupSelectMaxK = 256;

constexpr int kSelectMaxK = 1024;

}  // namespace isx
