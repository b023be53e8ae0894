// select.hip -- canonical top-k selection over score rows and the multi-shard merge.
//
// Replaces the reference's  sim.max(1) / sim.kthvalue / sim[i].sort(descending)  consumers
// (utils/metrics.py:10-13,33) for the top of the ranking, with a DEFINED tie-break
// (score desc, index asc).  Every candidate is a single u64 key
// (orderable(score) << 32 | ~index), so "k best" = "k largest keys" and ties cannot occur.
//
// Two kernels:
//   select_groups_kernel (k <= 256, the usual case): ONE WAVE per query row over a materialised or FILTERED score chunk;
//     running list unsorted in LDS, threshold updates by radix select, one sort at the end of the search (see below).
//   select_kernel (k up to 1024): one 256-thread workgroup per row.  Scores stream from HBM/L2 in 16-B loads; candidates
//     beating the current k-th key are appended to a 2048-entry LDS buffer (wave prefix by shuffles + one LDS atomic per
//     wave); when the buffer could overflow it is bitonic-sorted and cut back to k, which also tightens the threshold.
//   topk_merge_kernel: canonical merge of P per-shard lists.
#include "isx_internal.hpp"

namespace isx {

constexpr int SEL_THREADS = 256;
constexpr int SEL_CAP = 2048;
constexpr int SEL_STRIP = SEL_THREADS * 4;

// Exclusive prefix sum over the 64 lanes of a wave (all lanes active) + the wave total: DPP row shifts inside the rows of 16 lanes,
// then the two row broadcasts (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3) -- six VALU adds, where six dependent
// __shfl_up (ds_bpermute through the LDS crossbar) cost ~10^2 cycles each.  A lane without a source keeps `old` = 0.
__device__ __forceinline__ int wave_excl_prefix(int v, int lane, int& total) {
    (void)lane;
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false);      // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false);      // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false);      // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false);      // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1, 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2, 3
    total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
}

template <bool VEC>
__global__ __launch_bounds__(SEL_THREADS) void select_kernel(const float* __restrict__ sim, int64_t Nc, int64_t ld,
                                                             uint32_t col_base, int k, uint64_t* __restrict__ carry,
                                                             int first, int emit, int64_t idx_base,
                                                             float* __restrict__ top_score, int64_t* __restrict__ top_idx,
                                                             float* __restrict__ thr_out) {
    __shared__ __attribute__((aligned(16))) uint64_t buf[SEL_CAP];
    __shared__ int cnt_s, dirty_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row = blockIdx.x;
    const float* r = sim + row * ld;

    for (int i = tid; i < SEL_CAP; i += SEL_THREADS) buf[i] = (!first && i < k) ? carry[row * k + i] : 0ull;
    if (tid == 0) { cnt_s = first ? 0 : k; dirty_s = 0; }
    __syncthreads();
    uint64_t thr = first ? 0ull : buf[k - 1];   // k-th best so far (0 = fewer than k real entries)
    for (int64_t s0 = 0; s0 < Nc; s0 += SEL_STRIP) {
        // make room for a full strip.  cnt_s is stable here (last update precedes the
        // previous barrier); the barrier below keeps this strip's atomics behind every read.
        const int cnt_now = cnt_s;
        __syncthreads();
        if (cnt_now + SEL_STRIP > SEL_CAP) {
            bitonic_sort_desc<SEL_THREADS>(buf, SEL_CAP);
            for (int i = k + tid; i < SEL_CAP; i += SEL_THREADS) buf[i] = 0ull;
            if (tid == 0) cnt_s = k;
            __syncthreads();
            thr = buf[k - 1];
        }
        const int64_t j0 = s0 + (int64_t)tid * 4;
        uint64_t key[4];
        bool take[4];
        int c = 0;
        if (VEC && j0 + 3 < Nc) {
            const float4 v = *reinterpret_cast<const float4*>(r + j0);
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                key[q] = rank_key(vv[q], col_base + (uint32_t)(j0 + q));
                take[q] = key[q] > thr;
                c += take[q] ? 1 : 0;
            }
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool in = (j0 + q < Nc);
                key[q] = in ? rank_key(r[in ? j0 + q : 0], col_base + (uint32_t)(j0 + q)) : 0ull;
                take[q] = in && key[q] > thr;
                c += take[q] ? 1 : 0;
            }
        }
        int total;
        const int pre = wave_excl_prefix(c, lane, total);
        int base = 0;
        if (lane == 0 && total > 0) { base = atomicAdd(&cnt_s, total); dirty_s = 1; }
        base = __shfl(base, 0, 64);
        int off = base + pre;
        // static register indexing only (runtime-indexed arrays would go to scratch)
#pragma unroll
        for (int q = 0; q < 4; ++q) if (take[q]) buf[off++] = key[q];
        __syncthreads();   // appends + cnt_s visible before the next strip (unconditional barrier)
    }
    if (dirty_s != 0 || first) {
        bitonic_sort_desc<SEL_THREADS>(buf, SEL_CAP);
    }
    if (emit) {
        for (int i = tid; i < k; i += SEL_THREADS) {
            const uint64_t kk = buf[i];
            top_score[row * k + i] = kk ? key_score(kk) : -INFINITY;
            top_idx[row * k + i] = kk ? (idx_base + (int64_t)key_idx(kk)) : -1;
        }
    } else {
        for (int i = tid; i < k; i += SEL_THREADS) carry[row * k + i] = buf[i];
    }
    if (thr_out && tid == 0) thr_out[row] = buf[k - 1] ? key_score(buf[k - 1]) : -INFINITY;
}

// ---------------------------------------------------------------------------------------------
// Running top-k update from a chunk produced by the FILTERING GEMM epilogue (cosine.hip, fast.hip): one
// wave (64-thread workgroup) per query row.  gflag marks the 32-column groups the GEMM stored (those
// with a score reaching the row's threshold); nothing else can change the list.  The wave scans the
// flag row 64 groups at a time, then reads up to eight qualifying 128-B segments per step (four
// independent loads per half-wave, so the memory latency of a step is paid once, not four times).
//
// LDS buffer: buf[0, k) = the current list, sorted (canonical order, 0 = empty);  buf[k, cnt) = keys
// that beat the k-th key since the last flush, unsorted.  A flush sorts ONLY the new keys (bitonic
// over next_pow2(m) <= 512 entries) and merges the two sorted runs by rank: every key finds its final
// position with one binary search in the other run, positions >= k are dropped.  (Sorting the whole
// buffer instead made this kernel LDS-write-bound: 55 compare-exchange stages over 1024 keys per flush.)
constexpr int GS_NEW = 512;                        // capacity of the unsorted run
constexpr int GS_BUF = kGroupSelectMaxK + GS_NEW;  // 768 keys = 6 KB

// ---- flush: keep the k largest of the n = cnt keys in buf, by RADIX SELECT (no sort) -----------------------
// The list is kept UNSORTED between flushes; only its k-th largest key T (the admission threshold) is needed.
// T is found most-significant byte first: a 256-bin LDS histogram of the current byte over the keys that still match
// the prefix, a suffix scan for the bin that holds the rank-k key, repeat on the next byte; bytes on which all keys
// agree are skipped (wave AND / OR), and the search stops as soon as the bin holds one key.  Survivors (key >= T) are
// compacted to buf[0, k) with ballot prefix sums.  ~1 k wave instructions per flush, where sorting the new run and
// rank-merging it cost ~3.4 k: the bitonic networks were instruction-bound.  One sort of the k keys remains,
// at the end of the kernel.
constexpr int GS_R = GS_BUF / 64;                  // keys per lane held in registers during a flush (12)

__device__ __forceinline__ uint64_t wave_or_u64(uint64_t v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) {
        const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)v, s, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(v >> 32), s, 64);
        v |= ((uint64_t)hi << 32) | lo;
    }
    return v;
}

// The want-th largest of the cnt keys a wave holds in registers (lane l: keys l, 64 + l, ...; slots >= cnt are zero): radix descent, most
// significant byte first.  Uniform control flow; the barriers are workgroup barriers of the one-wave workgroup.  0 when want > #keys.
template <int R>
__device__ __forceinline__ uint64_t radix_kth(const uint64_t (&v)[R], int cnt, int want, uint32_t* hist, int lane) {
    uint64_t all_and = ~0ull, all_or = 0ull;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = r * 64 + lane;
        if (r * 64 < cnt) { all_or |= v[r]; all_and &= (i < cnt) ? v[r] : ~0ull; }
    }
    all_or = wave_or_u64(all_or);
    all_and = ~wave_or_u64(~all_and);
    const uint64_t diff = all_and ^ all_or;        // bits on which the keys differ (uniform)
    uint64_t prefix = 0ull, mask = 0ull, T = 0ull;
    bool done = false;
#pragma unroll 1
    for (int shift = 56; shift >= 0 && !done; shift -= 8) {
        if (((diff >> shift) & 255ull) == 0ull) {  // every key has the same byte here
            prefix |= ((all_and >> shift) & 255ull) << shift;
            mask |= 255ull << shift;
            continue;
        }
        for (int i = lane; i < 256; i += 64) hist[i] = 0u;
        __syncthreads();
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (r * 64 + lane < cnt && (v[r] & mask) == prefix) atomicAdd(&hist[(unsigned)(v[r] >> shift) & 255u], 1u);
        __syncthreads();
        // lane l owns bins 4l .. 4l+3; suffix sums from bin 255 downwards
        const uint4 c = reinterpret_cast<const uint4*>(hist)[lane];
        const int lane_sum = (int)(c.x + c.y + c.z + c.w);
        int total;
        const int below_incl = wave_excl_prefix(lane_sum, lane, total) + lane_sum;     // bins of lanes <= lane
        int cum = total - below_incl;                                                  // keys in bins of higher lanes
        const int cj[4] = {(int)c.x, (int)c.y, (int)c.z, (int)c.w};
        int digit = -1, above = 0, inbin = 0;
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            if (digit < 0 && cum < want && cum + cj[j] >= want) { digit = 4 * lane + j; above = cum; inbin = cj[j]; }
            cum += cj[j];
        }
        const unsigned long long who = __ballot(digit >= 0);
        if (who == 0ull) {                         // want > number of keys in the class: cannot happen (cnt >= want)
            T = 0ull;
            done = true;
            break;
        }
        const int src = __ffsll((long long)who) - 1;
        digit = __shfl(digit, src, 64);
        above = __shfl(above, src, 64);
        inbin = __shfl(inbin, src, 64);
        want -= above;
        prefix |= (uint64_t)digit << shift;
        mask |= 255ull << shift;
        if (inbin == 1) {                          // a single key left in the class: it is T
            uint64_t f = 0ull;
#pragma unroll
            for (int r = 0; r < R; ++r)
                if (r * 64 + lane < cnt && (v[r] & mask) == prefix) f |= v[r];
            T = wave_or_u64(f);
            done = true;
        }
    }
    if (!done) T = prefix;                         // all eight bytes fixed
    return T;
}

// Returns the new threshold key (k-th largest; 0 while the list holds fewer than k keys).  cnt > k on entry; on exit
// buf[0, k) holds the survivors (unsorted, zero padded).  Uniform control flow; all 64 lanes call it.
__device__ __forceinline__ uint64_t gs_flush(uint64_t* buf, uint32_t* hist, int k, int cnt, int lane) {
    __syncthreads();                               // appended keys visible
    uint64_t v[GS_R];
#pragma unroll
    for (int r = 0; r < GS_R; ++r) {
        const int i = r * 64 + lane;
        v[r] = (i < cnt) ? buf[i] : 0ull;          // slots beyond cnt behave like empty (zero) entries
    }
    const uint64_t T = radix_kth<GS_R>(v, cnt, k, hist, lane);
    // compaction: the non-empty keys >= T (distinct keys: exactly k of them when T != 0)
    __syncthreads();
    int base = 0;
#pragma unroll
    for (int r = 0; r < GS_R; ++r) {
        if (r * 64 < cnt) {                        // uniform
            const bool keep = (v[r] != 0ull) && (v[r] >= T);
            const unsigned long long km = __ballot(keep);
            if (keep) buf[base + __popcll(km & ((1ull << lane) - 1ull))] = v[r];
            base += __popcll(km);
        }
    }
    __syncthreads();
    for (int i = base + lane; i < k; i += 64) buf[i] = 0ull;
    __syncthreads();
    if (base < k) return 0ull;                     // list not full yet: slot k-1 is empty (0), everything is admitted
    // keep the threshold key in slot k-1 (the unsorted carry format: buf[k-1] is always the k-th largest)
    for (int i = lane; i < k; i += 64)
        if (buf[i] == T) { buf[i] = buf[k - 1]; buf[k - 1] = T; }      // exactly one lane matches (distinct keys)
    __syncthreads();
    return T;
}

__global__ __launch_bounds__(64) void select_groups_kernel(const float* __restrict__ sim, const uint8_t* __restrict__ gflag,
                                                           int64_t Nc, int64_t ld, int ngrp, uint32_t col_base, int k,
                                                           uint64_t* __restrict__ carry, float* __restrict__ thr, int mode,
                                                           int64_t idx_base, float* __restrict__ top_score,
                                                           int64_t* __restrict__ top_idx, const int* __restrict__ m_active,
                                                           const int* __restrict__ row_map, const float* __restrict__ win, int k_win) {
    // win / k_win (optional, the approximate pass of fast.hip): only candidates with score >= a - win[row] can matter to the caller, a = the
    // k_win-th best score of the final list (k_win < k).  a only grows from chunk to chunk, so the threshold handed to the next filter GEMM
    // is max(k-th best, CURRENT k_win-th best - win[row]) instead of the k-th best alone: fewer stored groups, same final window.
    // mode 0: intermediate chunk, the carry stays UNSORTED with the k-th largest key in slot k-1; 1: last chunk, emit the sorted
    // lists; 2: last chunk, leave the sorted keys in the carry (fast.hip re-scores them)
    __shared__ __attribute__((aligned(16))) uint64_t buf[GS_BUF];
    __shared__ __attribute__((aligned(16))) uint32_t hist[256];
    const int lane = threadIdx.x, l31 = lane & 31, half = lane >> 5;
    const int64_t row = blockIdx.x;
    if (m_active && row >= *m_active) return;
    const float* r = sim + row * ld;
    const uint8_t* gf = gflag ? gflag + row * (int64_t)ngrp : nullptr;     // null = every group present
    for (int i = lane; i < k; i += 64) buf[i] = carry ? carry[row * k + i] : 0ull;   // null carry = empty list
    __syncthreads();
    int kp2 = 1;
    while (kp2 < k) kp2 <<= 1;
    uint64_t thr_key = buf[k - 1];
    int cnt = k;                                   // uniform

    float thr_f = thr_key ? key_score(thr_key) : -INFINITY;      // score of the threshold key; a list that is not full admits everything
    if (!gf) {
        // every group present (isx_topk_rows, bootstrap chunk): plain streaming scan, 256 scores per step.  Whole steps of 16-B aligned
        // rows run two loads deep (registers A / B, unrolled: the wave waits for a step while the next one is in flight) behind a cheap
        // reject: v < thr in float order (neither a NaN) implies key < thr_key, so the wave builds keys and a prefix sum only when some
        // lane may hold a candidate.  The ragged end of the row and unaligned rows take the scalar loop below.
        const bool vec = ((((uintptr_t)r) & 15) == 0);
        const int64_t nfull = vec ? (Nc >> 8) : 0;
        auto consume = [&](int64_t s, const float4& S) {
            if (__ballot(!(S.x < thr_f && S.y < thr_f && S.z < thr_f && S.w < thr_f)) == 0ull) return;
            const float vv[4] = {S.x, S.y, S.z, S.w};
            const uint32_t j = col_base + (uint32_t)((s << 8) + lane * 4);
            uint64_t key[4];
            bool take[4];
            int c = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                key[q] = rank_key(vv[q], j + (uint32_t)q);
                take[q] = key[q] > thr_key;
                c += take[q] ? 1 : 0;
            }
            int total;
            int off = cnt + wave_excl_prefix(c, lane, total);
#pragma unroll
            for (int q = 0; q < 4; ++q) if (take[q]) buf[off++] = key[q];
            cnt += total;
        };
        auto fetch = [&](int64_t s, float4& S) {       // past the end: a valid address, the step is never consumed
            S = *reinterpret_cast<const float4*>(r + ((s < nfull ? s : nfull - 1) << 8) + lane * 4);
        };
        auto flush_if_full = [&]() {
            if (cnt - k > GS_NEW - 256) {              // room for a full step
                thr_key = gs_flush(buf, hist, k, cnt, lane);
                thr_f = thr_key ? key_score(thr_key) : -INFINITY;
                cnt = k;
            }
        };
        if (nfull > 0) {
            float4 A, B;
            fetch(0, A); fetch(1, B);
            for (int64_t s = 0; s < nfull; s += 2) {
                flush_if_full();
                consume(s, A); fetch(s + 2, A);
                if (s + 1 < nfull) {
                    flush_if_full();
                    consume(s + 1, B);
                }
                fetch(s + 3, B);
            }
        }
        for (int64_t j0 = nfull << 8; j0 < Nc; j0 += 256) {
            flush_if_full();
            const int64_t j = j0 + (int64_t)lane * 4;
            uint64_t key[4];
            bool take[4];
            int c = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                key[q] = (j + q < Nc) ? rank_key(r[j + q], col_base + (uint32_t)(j + q)) : 0ull;
                take[q] = key[q] > thr_key;
                c += take[q] ? 1 : 0;
            }
            int total;
            int off = cnt + wave_excl_prefix(c, lane, total);
#pragma unroll
            for (int q = 0; q < 4; ++q) if (take[q]) buf[off++] = key[q];
            cnt += total;
        }
    } else if (ngrp > 0) {
        // filtered chunk.  The flag row is read 256 groups at a time -- four coalesced byte loads per lane in flight together, the next
        // four issued before this block's segments are fetched -- instead of one dependent load per 64 groups.
        auto flag_at = [&](int g) -> unsigned { return gf[g < ngrp ? g : ngrp - 1]; };     // clamped: the loads stay unconditional
        unsigned fb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) fb[b] = flag_at(b * 64 + lane);
        for (int g0 = 0; g0 < ngrp; g0 += 256) {
            unsigned long long m[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) m[b] = __ballot((g0 + b * 64 + lane < ngrp) && fb[b] != 0u);
            if (g0 + 256 < ngrp) {
#pragma unroll
                for (int b = 0; b < 4; ++b) fb[b] = flag_at(g0 + 256 + b * 64 + lane);
            }
#pragma unroll 1
            for (int b = 0; b < 4; ++b) {
                unsigned long long mask = b == 0 ? m[0] : b == 1 ? m[1] : b == 2 ? m[2] : m[3];
                const int gbase = g0 + b * 64;
                while (mask) {
                    if (cnt - k > GS_NEW - 256) {          // this step can add up to 4 x 64 keys
                        thr_key = gs_flush(buf, hist, k, cnt, lane);
                        thr_f = thr_key ? key_score(thr_key) : -INFINITY;
                        cnt = k;
                    }
                    // up to eight qualifying groups per step: sub-step u gives one to each half-wave
                    int64_t col[4];
                    float v[4];
                    bool in[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        int ga = -1, gb = -1;
                        if (mask) { ga = __ffsll((long long)mask) - 1; mask &= mask - 1; }
                        if (mask) { gb = __ffsll((long long)mask) - 1; mask &= mask - 1; }
                        const int gsel = half ? gb : ga;
                        col[u] = (int64_t)(gbase + gsel) * 32 + l31;
                        in[u] = (gsel >= 0) && (col[u] < Nc);
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) v[u] = in[u] ? r[col[u]] : 0.0f;      // four loads in flight
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (__ballot(in[u] && !(v[u] < thr_f)) == 0ull) continue;     // v < thr (float order) implies key < thr_key
                        const uint64_t key = in[u] ? rank_key(v[u], col_base + (uint32_t)col[u]) : 0ull;
                        const bool take = in[u] && key > thr_key;
                        const unsigned long long tm = __ballot(take);
                        if (tm) {
                            const int pos = cnt + __popcll(tm & ((1ull << lane) - 1ull));
                            if (take) buf[pos] = key;
                            cnt += __popcll(tm);
                        }
                    }
                }
            }
        }
    }
    if (cnt > k) gs_flush(buf, hist, k, cnt, lane);
    __syncthreads();
    if (mode != 0) {                               // the list leaves the search sorted (canonical order)
        for (int i = k + lane; i < kp2; i += 64) buf[i] = 0ull;
        bitonic_sort_desc<64>(buf, kp2);
    }
    const int emit = (mode == 1);
    if (emit) {
        const int64_t orow = row_map ? (int64_t)row_map[row] : row;
        for (int i = lane; i < k; i += 64) {
            const uint64_t kk = buf[i];
            top_score[orow * k + i] = kk ? key_score(kk) : -INFINITY;
            top_idx[orow * k + i] = kk ? (idx_base + (int64_t)key_idx(kk)) : -1;
        }
    } else {
        for (int i = lane; i < k; i += 64) carry[row * k + i] = buf[i];
        if (thr) {
            float t = buf[k - 1] ? key_score(buf[k - 1]) : -INFINITY;
            if (win && mode == 0 && k_win < k && buf[k - 1] != 0ull) {      // full list (uniform): its k_win-th best key by one more radix descent
                uint64_t v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (r * 64 + lane < k) ? buf[r * 64 + lane] : 0ull;
                const uint64_t kw = radix_kth<4>(v, k, k_win, hist, lane);
                if (kw != 0ull) {
                    const float a = key_score(kw);
                    // rounded DOWN past the caller's own evaluation of a_final - win - 2e-7 |a_final| (a <= a_final; fast.hip, rescore_kernel)
                    t = fmaxf(t, a - win[row] - 4e-7f * fabsf(a));
                }
            }
            if (lane == 0) thr[row] = t;
        }
    }
}

int launch_select_groups(const float* sim, const uint8_t* gflag, int64_t M, int64_t Nc, int64_t ld, int64_t col_base, int k,
                         uint64_t* carry, float* thr, int mode, int64_t idx_base, float* top_score, int64_t* top_idx,
                         hipStream_t st, const int* m_active, const int* row_map, const float* win, int k_win) {
    if (M == 0) return ISX_OK;
    if (M >= (1ll << 31) || k > kGroupSelectMaxK) { isx_set_error("select_groups: unsupported M=%lld k=%d", (long long)M, k); return ISX_ERR_ARG; }
    const int ngrp = (int)((Nc + 31) / 32);
    hipLaunchKernelGGL(select_groups_kernel, dim3((unsigned)M), dim3(64), 0, st, sim, gflag, Nc, ld, ngrp, (uint32_t)col_base, k, carry, thr,
                       mode, idx_base, top_score, top_idx, m_active, row_map, win, k_win);
    ISX_CHECK_LAUNCH("select_groups");
    return ISX_OK;
}

int launch_select(const float* sim, int64_t M, int64_t Nc, int64_t ld, int64_t col_base, int k, uint64_t* carry,
                  bool first, bool emit, int64_t idx_base, float* top_score, int64_t* top_idx, hipStream_t st, float* thr_out) {
    if (M == 0) return ISX_OK;
    if (M >= (1ll << 31)) { isx_set_error("select: too many rows"); return ISX_ERR_ARG; }
    const bool vec = sim && ((uintptr_t)sim % 16 == 0) && (ld % 4 == 0);
    if (vec) hipLaunchKernelGGL(select_kernel<true>, dim3((unsigned)M), dim3(SEL_THREADS), 0, st, sim, Nc, ld, (uint32_t)col_base, k,
                                carry, first ? 1 : 0, emit ? 1 : 0, idx_base, top_score, top_idx, thr_out);
    else hipLaunchKernelGGL(select_kernel<false>, dim3((unsigned)M), dim3(SEL_THREADS), 0, st, sim, Nc, ld, (uint32_t)col_base, k,
                            carry, first ? 1 : 0, emit ? 1 : 0, idx_base, top_score, top_idx, thr_out);
    ISX_CHECK_LAUNCH("select");
    return ISX_OK;
}

// Merge of P per-shard lists: one workgroup per query, keys rebuilt from (score, global
// index), bitonic sort of next_pow2(P*k) <= 4096 keys in LDS.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* __restrict__ scores, const int64_t* __restrict__ idx, int P,
                                                         int64_t M, int k, int n2, float* __restrict__ out_s,
                                                         int64_t* __restrict__ out_i) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int64_t m = blockIdx.x;
    const int T = P * k;
    for (int t = threadIdx.x; t < n2; t += 256) {
        uint64_t key = 0;
        if (t < T) {
            const int p = t / k, j = t - p * k;
            const int64_t o = ((int64_t)p * M + m) * k + j;
            const int64_t gi = idx[o];
            if (gi >= 0) key = rank_key(scores[o], (uint32_t)gi);
        }
        keys[t] = key;
    }
    bitonic_sort_desc<256>(keys, n2);
    for (int i = threadIdx.x; i < k; i += 256) {
        const uint64_t kk = keys[i];
        out_s[m * k + i] = kk ? key_score(kk) : -INFINITY;
        out_i[m * k + i] = kk ? (int64_t)key_idx(kk) : -1;
    }
}

}  // namespace isx

using namespace isx;

ISX_API int isx_topk_rows(const float* sim, int64_t M, int64_t N, int k, int64_t idx_base, float* top_score,
                          int64_t* top_idx, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && N <= 0x7FFFFFFFll, "isx_topk_rows: bad shape M=%lld N=%lld", (long long)M, (long long)N);
    ISX_REQUIRE(k >= 1 && k <= kSelectMaxK, "isx_topk_rows: k=%d outside [1,%d]", k, kSelectMaxK);
    ISX_REQUIRE(idx_base >= 0 && idx_base + N <= 0xFFFFFFFFll, "isx_topk_rows: gallery indices must stay below 2^32");
    ISX_REQUIRE((top_score && top_idx && (sim || N == 0)) || M == 0, "isx_topk_rows: null pointer");
    if (k <= kGroupSelectMaxK && M > 0)     // wave-per-row kernel: all groups present, empty carry
        return launch_select_groups(sim, nullptr, M, N, N, 0, k, nullptr, nullptr, 1, idx_base, top_score, top_idx, (hipStream_t)stream);
    return launch_select(sim, M, N, N, 0, k, nullptr, true, true, idx_base, top_score, top_idx, (hipStream_t)stream);
}

ISX_API int isx_topk_merge(const float* scores, const int64_t* idx, int P, int64_t M, int k, float* out_s, int64_t* out_i,
                           isx_stream_t stream) {
    ISX_REQUIRE(P >= 1 && M >= 0 && k >= 1 && M < (1ll << 31), "isx_topk_merge: bad shape P=%d M=%lld k=%d", P, (long long)M, k);
    ISX_REQUIRE((int64_t)P * k <= 4096, "isx_topk_merge: P*k=%lld exceeds 4096", (long long)P * k);
    ISX_REQUIRE((scores && idx && out_s && out_i) || M == 0, "isx_topk_merge: null pointer");
    if (M == 0) return ISX_OK;
    const int n2 = next_pow2(P * k < 2 ? 2 : P * k);
    hipLaunchKernelGGL(topk_merge_kernel, dim3((unsigned)M), dim3(256), (size_t)n2 * 8, (hipStream_t)stream, scores, idx, P, M, k, n2,
                       out_s, out_i);
    ISX_CHECK_LAUNCH("isx_topk_merge");
    return ISX_OK;
}
