// isx_internal.hpp -- launchers shared between translation units of libisx.
#pragma once
#include <atomic>

#include "isx_common.hpp"

namespace isx {

// C[m][n] = sum_k Q[m][k] * G[n][k]  (k-ordered fp32 fma chain), C row stride ldc.
// m_active (optional, device scalar): tiles whose first row is >= *m_active exit immediately.
int launch_cosine_gemm(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                       hipStream_t st, const int* m_active = nullptr);

// Same GEMM with the filtering epilogue: the scores of a (row, 32-column group) are stored to C only
// when one of them reaches thr[row] (a lower bound of the row's final k-th best score); gflag
// (M, ngrp = ceil(N/32)) bytes say which groups were stored.
int launch_cosine_gemm_filter(const float* Q, int64_t M, const float* G, int64_t N, int D, float* C, int64_t ldc,
                              const float* thr, uint8_t* gflag, hipStream_t st, const int* m_active = nullptr);

// 1x1 convolution over channels-last pixels: the same GEMM, epilogue y = act(. + bias[n] (+ residual)) (used by conv.hip)
int launch_conv1x1_gemm(const float* x, int64_t M, const float* w, int64_t N, int D, float* y, const float* bias, const float* residual, int relu,
                        hipStream_t st);

// Tail of a convolution launch in 128x128 tiles: rows covered by whole rounds of 1024 resident workgroups when the rest of the grid is a
// partial round -- the caller runs the remaining rows as 64x64 tiles in the same grid; 0 = no split.  g_tail_split: debug / A-B switch.
extern std::atomic<int> g_tail_split;      // (the debug / A-B knobs are relaxed atomics: a test thread may flip them while another thread launches)
int64_t gemm_tail_split_rows(int64_t M, int64_t N, int64_t slots = 1024);       // slots: resident 128x128 workgroups of the calling kernel (256 CUs x workgroups per CU)
// Tile shape (0 = 128x128, 1 = 64x128, 2 = 128x64, 3 = 64x64) with the smallest estimated launch time among those in `mask` (rounds of resident
// workgroups + tail + per-CU quantisation, cosine.hip); eff[4]: steady-state efficiency per shape of the calling kernel family.
int pick_tile_cfg(int64_t M, int64_t N, int64_t split, const float* eff, unsigned mask, int wg_per_cu_128 = 4);

// Streaming variant for the HBM-bound Cin = 64 layers (stream1x1.hip): persistent workgroups, weights in registers, pixel tiles by LDS-DMA.
bool conv1x1_stream_applicable(int64_t M, int Cin, int Cout, const float* x, const float* res);
int launch_conv1x1_stream(const float* x, int64_t M, const float* w, int Cin, int Cout, const float* bias, const float* res, int relu, float* y,
                          hipStream_t st);

// fp16-operand variant (fast.hip): approximate scores, fp32 accumulate; gflag == nullptr -> plain GEMM.
int launch_gemm_f16(const _Float16* Q, int64_t M, const _Float16* G, int64_t N, int D, float* C, int64_t ldc, const float* thr,
                    uint8_t* gflag, hipStream_t st);

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
                         uint64_t* carry, float* thr, int mode, int64_t idx_base, float* top_score, int64_t* top_idx,
                         hipStream_t st, const int* m_active = nullptr, const int* row_map = nullptr, const float* win = nullptr, int k_win = 0);
// mode 0: intermediate chunk (carry unsorted, k-th largest key in slot k-1), 1: last chunk, emit sorted lists, 2: last chunk, sorted carry.
// m_active / row_map (optional, device): only rows < *m_active are processed and row r emits to output row row_map[r].
// win / k_win (optional): the caller only needs candidates within win[row] of the final k_win-th best score (k_win < k); the threshold
// written for the next filter GEMM is then max(k-th best, current k_win-th best - win[row]).

constexpr int kGroupSelectMaxK = 256;

constexpr int kSelectMaxK = 1024;

// One chunked running-top-k search (cosine.hip).  fp32 operands (Q, G) or fp16 operands (Qh, Gh); with
// emit the final lists go to top_score / top_idx, otherwise the keys stay in the workspace (carry at
// ws + 0, thr after it).  Workspace layout: [carry M*k u64 | thr M f32 | flags | score chunk].
struct TopkJob {
    const char* who;
    const float* Q; const float* G;
    const _Float16* Qh; const _Float16* Gh;
    int64_t M, N; int D, k;
    int64_t idx_base; float* top_score; int64_t* top_idx; bool emit;
    void* ws; size_t ws_bytes; hipStream_t st;
    const int* m_active; const int* row_map;
    const float* win; int k_win;                  // optional window of interest below the k_win-th best score (see launch_select_groups)
};
int run_topk_chunks(const TopkJob& job);
size_t topk_fixed_bytes(int64_t M, int k);
size_t topk_chunk_bytes(int64_t M, int64_t nc);
int64_t topk_recommended_chunk(int64_t M, int64_t N);

}  // namespace isx
