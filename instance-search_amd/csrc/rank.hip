// rank.hip -- full canonical ranking of every query row, Oxford-style average precision
// and the label-masked similarity sums of the evaluator.
//
// Reference behaviour restated:
//   utils/metrics.py:33      `_, ranked_list = sim[i].sort(dim=0, descending=True)`
//   utils/metrics.py:25-45   avg_precision (trapezoid AP in float64, rank by rank)
//   utils/train_siamese.py:74-76  sum_pos / sum_neg statistics
//
// Ranking = bitonic sort of u64 keys (orderable(score) << 32 | ~index), descending: the
// canonical (score desc, index asc) order with no ties left.  Rows up to 4096 columns
// are sorted entirely in LDS by one workgroup; longer rows sort 4096-key tiles in LDS
// and finish the large strides with coalesced global compare-exchange passes.
#include <type_traits>

#include "isx_common.hpp"

namespace isx {

constexpr int RK_TILE = 4096;
constexpr int RK_THREADS = 256;

// strides < tile inside a tile whose first key has global position gbase; sizes from
// size_lo to size_hi (inclusive), for size > tile only strides <= tile/2 are done.
__device__ __forceinline__ void bitonic_tile(uint64_t* keys, int tile, int64_t gbase, int64_t size_lo, int64_t size_hi) {
    for (int64_t size = size_lo; size <= size_hi; size <<= 1) {
        int s0 = (int)(size >> 1 < tile ? size >> 1 : tile >> 1);
        for (int stride = s0; stride > 0; stride >>= 1) {
            __syncthreads();
            for (int t = threadIdx.x; t < (tile >> 1); t += RK_THREADS) {
                const int lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
                const bool desc = (((gbase + lo) & size) == 0);
                const uint64_t a = keys[lo], b = keys[hi];
                if ((a < b) == desc) { keys[lo] = b; keys[hi] = a; }
            }
        }
    }
    __syncthreads();
}

// N2 <= RK_TILE: one workgroup per row does everything.
__global__ __launch_bounds__(RK_THREADS) void rank_small_kernel(const float* __restrict__ sim, int64_t N, int N2,
                                                                int64_t* __restrict__ ranked) {
    extern __shared__ __attribute__((aligned(16))) uint64_t keys[];
    const int64_t row = blockIdx.x;
    for (int j = threadIdx.x; j < N2; j += RK_THREADS) keys[j] = (j < N) ? rank_key(sim[row * N + j], (uint32_t)j) : 0ull;
    bitonic_tile(keys, N2, 0, 2, N2);
    for (int j = threadIdx.x; j < N; j += RK_THREADS) ranked[row * N + j] = key_idx(keys[j]);
}

// Large rows, phase 1: build keys and sort each RK_TILE tile (alternating directions).
__global__ __launch_bounds__(RK_THREADS) void rank_tile_sort_kernel(const float* __restrict__ sim, int64_t N, int64_t N2,
                                                                    uint64_t* __restrict__ gkeys) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[RK_TILE];
    const int64_t row = blockIdx.y, gbase = (int64_t)blockIdx.x * RK_TILE;
    for (int j = threadIdx.x; j < RK_TILE; j += RK_THREADS) {
        const int64_t g = gbase + j;
        keys[j] = (g < N) ? rank_key(sim[row * N + g], (uint32_t)g) : 0ull;
    }
    bitonic_tile(keys, RK_TILE, gbase, 2, RK_TILE);
    uint64_t* dst = gkeys + row * N2 + gbase;
    for (int j = threadIdx.x; j < RK_TILE; j += RK_THREADS) dst[j] = keys[j];
}

// One global compare-exchange pass at distance `stride` (>= RK_TILE) of merge level `size`.
__global__ __launch_bounds__(RK_THREADS) void rank_global_step_kernel(uint64_t* __restrict__ gkeys, int64_t N2, int64_t size,
                                                                      int64_t stride) {
    const int64_t row = blockIdx.y;
    uint64_t* k = gkeys + row * N2;
    const int64_t t = (int64_t)blockIdx.x * RK_THREADS + threadIdx.x;   // pair index < N2/2
    const int64_t lo = 2 * t - (t & (stride - 1)), hi = lo + stride;
    const bool desc = ((lo & size) == 0);
    const uint64_t a = k[lo], b = k[hi];
    if ((a < b) == desc) { k[lo] = b; k[hi] = a; }
}

// Remaining strides (< RK_TILE) of merge level `size`, tile by tile in LDS.
__global__ __launch_bounds__(RK_THREADS) void rank_tile_finish_kernel(uint64_t* __restrict__ gkeys, int64_t N2, int64_t size) {
    __shared__ __attribute__((aligned(16))) uint64_t keys[RK_TILE];
    const int64_t row = blockIdx.y, gbase = (int64_t)blockIdx.x * RK_TILE;
    uint64_t* src = gkeys + row * N2 + gbase;
    for (int j = threadIdx.x; j < RK_TILE; j += RK_THREADS) keys[j] = src[j];
    bitonic_tile(keys, RK_TILE, gbase, size, size);
    for (int j = threadIdx.x; j < RK_TILE; j += RK_THREADS) src[j] = keys[j];
}

__global__ __launch_bounds__(RK_THREADS) void rank_emit_kernel(const uint64_t* __restrict__ gkeys, int64_t N, int64_t N2,
                                                               int64_t* __restrict__ ranked) {
    const int64_t row = blockIdx.y;
    const int64_t j = (int64_t)blockIdx.x * RK_THREADS + threadIdx.x;
    if (j < N) ranked[row * N + j] = key_idx(gkeys[row * N2 + j]);
}

// ------------------------------------------------------------ average precision --
constexpr int AP_PER_THREAD = 8;
constexpr int AP_CHUNK = RK_THREADS * AP_PER_THREAD;

__global__ __launch_bounds__(RK_THREADS) void average_precision_kernel(const int64_t* __restrict__ ranked, int64_t N,
                                                                       const int32_t* __restrict__ qlab,
                                                                       const int32_t* __restrict__ glab, int kth,
                                                                       double* __restrict__ ap_out) {
    __shared__ double terms[AP_CHUNK];
    __shared__ int wsum[RK_THREADS / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t row = blockIdx.x;
    const int32_t q = qlab[row];
    const int64_t* rk = ranked + row * N;

    // n_pos = #gallery items with the query's label - (kth - 1)     (metrics.py:27-28)
    int cnt = 0;
    for (int64_t j = tid; j < N; j += RK_THREADS) cnt += (glab[j] == q);
    cnt = wave_sum(cnt);
    if (lane == 0) wsum[wave] = cnt;
    __syncthreads();
    int64_t n_pos = 0;
    for (int i = 0; i < RK_THREADS / 64; ++i) n_pos += wsum[i];
    n_pos -= (kth - 1);
    if (n_pos <= 0) {                                        // metrics.py:29-30 -> None
        if (tid == 0) ap_out[row] = __longlong_as_double(0x7FF8000000000000ll);
        return;
    }
    const double dn = (double)n_pos;
    double ap = 0.0;                                         // thread 0's sequential accumulator
    int64_t hits_before = 0;                                 // uniform: hits in earlier chunks
    const int64_t skip = kth - 1;                            // ranks n < skip are ignored entirely

    for (int64_t c0 = skip; c0 < N; c0 += AP_CHUNK) {
        const int64_t n0 = c0 + (int64_t)tid * AP_PER_THREAD;
        bool hit[AP_PER_THREAD];
        int c = 0;
#pragma unroll
        for (int e = 0; e < AP_PER_THREAD; ++e) {
            const int64_t n = n0 + e;
            hit[e] = (n < N) && (glab[rk[n < N ? n : 0]] == q);
            c += hit[e] ? 1 : 0;
        }
        // block exclusive scan of c
        int incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { int t = __shfl_up(incl, o, 64); if (lane >= o) incl += t; }
        __syncthreads();                                     // previous chunk's terms consumed
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int wbase = 0, tot = 0;
#pragma unroll
        for (int i = 0; i < RK_THREADS / 64; ++i) { if (i < wave) wbase += wsum[i]; tot += wsum[i]; }
        int pos = wbase + incl - c;                          // hits before this thread inside the chunk
#pragma unroll
        for (int e = 0; e < AP_PER_THREAD; ++e) {
            if (hit[e]) {
                const int64_t j = n0 + e - skip;             // position in the counted list (metrics.py:43)
                const int64_t h = hits_before + pos;         // intersect_size before this rank
                const double recall = (double)(h + 1) / dn;
                const double old_recall = (double)h / dn;
                const double precision = (double)(h + 1) / ((double)j + 1.0);
                const double old_precision = (j == 0) ? 1.0 : (double)h / (((double)j - 1.0) + 1.0);
                terms[pos] = (recall - old_recall) * ((old_precision + precision) / 2.0);
                ++pos;
            }
        }
        __syncthreads();
        if (tid == 0) for (int i = 0; i < tot; ++i) ap += terms[i];   // rank order, like the Python loop
        hits_before += tot;
    }
    if (tid == 0) ap_out[row] = ap;
}

// ----------------------------------------------- average precision without the sort --
// AP only depends on the RANKS OF THE POSITIVES: the reference's loop (utils/metrics.py:34-44) adds
// (recall - old_recall) * (...) at every rank, which is exactly 0 unless the rank holds a positive.
// Per query row: (1) the canonical keys of the P positives are collected and sorted (descending) in LDS; (2) ONE pass over the
// score row: a gallery key x below the smallest positive key cannot precede any positive and is dropped after one compare (on
// retrieval data that is nearly every x); any other x finds b(x) = #{positives > x} by binary search in the sorted keys and counts
// into a histogram H[b]; (3) rank of the i-th positive (in rank order) = sum_{b <= i} H[b] - 1 (itself), a prefix sum; (4) the same
// float64 terms as the reference, added in rank order by one thread -> bit-identical to the sorted path and to the reference.
// O(N log P) compares per row instead of the O(N P) of the first sort-free kernel (1 k x 100 k, 100 positives per query: 3.2 ms ->
// see BASELINE.md; 900 positives were SLOWER than the full sort).  Up to AP_MAXP positives per query (more -> ap = -1, the caller
// uses rank_full + average_precision for that row).
constexpr int AP_MAXP = 2048;            // positives per query the sort-free kernel handles; 16 KB keys (later: terms) + 8 KB histogram

// NT threads per query row; VEC: 16-B loads of the score row and the label array (N % 4 == 0, both 16-B aligned).  One workgroup per row is
// latency-bound when there are few rows: few-row launches run 1024 threads per row with four elements per load.
template <int NT, bool VEC>
__global__ __launch_bounds__(NT) void average_precision_sim_kernel(const float* __restrict__ sim, int64_t N,
                                                                   const int32_t* __restrict__ qlab,
                                                                   const int32_t* __restrict__ glab, int kth,
                                                                   double* __restrict__ ap_out) {
    __shared__ __attribute__((aligned(16))) uint64_t pkey[AP_MAXP];     // positive keys, sorted descending; re-used for the float64 terms
    __shared__ int hist[AP_MAXP];                                         // H[b], then the rank of the b-th positive
    __shared__ int wsum[NT / 64];
    __shared__ int npos_s, skip_s;
    const int tid = threadIdx.x, lane = tid & 63;
    const int64_t row = blockIdx.x;
    const int32_t q = qlab[row];
    const float* r = sim + row * N;
    if (tid == 0) { npos_s = 0; skip_s = 0; }
    __syncthreads();
    auto found = [&](int64_t j, float v) {
        const int slot = atomicAdd(&npos_s, 1);
        if (slot < AP_MAXP) pkey[slot] = rank_key(v, (uint32_t)j);
    };
    if (VEC) {
        const int4* g4 = reinterpret_cast<const int4*>(glab);
        const float4* r4 = reinterpret_cast<const float4*>(r);
        for (int64_t j4 = tid; j4 < (N >> 2); j4 += NT) {
            const int4 l = g4[j4];
            if (l.x == q || l.y == q || l.z == q || l.w == q) {          // rare: the scores are fetched only then
                const float4 v = r4[j4];
                if (l.x == q) found(4 * j4, v.x);
                if (l.y == q) found(4 * j4 + 1, v.y);
                if (l.z == q) found(4 * j4 + 2, v.z);
                if (l.w == q) found(4 * j4 + 3, v.w);
            }
        }
    } else {
        for (int64_t j = tid; j < N; j += NT)
            if (glab[j] == q) found(j, r[j]);
    }
    __syncthreads();
    const int n_lab = npos_s;
    const int64_t n_pos = (int64_t)n_lab - (kth - 1);
    if (n_pos <= 0) { if (tid == 0) ap_out[row] = __longlong_as_double(0x7FF8000000000000ll); return; }
    if (n_lab > AP_MAXP) { if (tid == 0) ap_out[row] = -1.0; return; }
    // (1) sort the positives (the collection order above is arbitrary); padding keys 0 sort last
    int n2 = 2;
    while (n2 < n_lab) n2 <<= 1;
    for (int p = n_lab + tid; p < n2; p += NT) pkey[p] = 0ull;
    for (int p = tid; p < n_lab; p += NT) hist[p] = 0;
    __syncthreads();
    bitonic_sort_desc<NT>(pkey, n2);
    __syncthreads();
    // (2) one pass over the row
    const uint64_t pmin = pkey[n_lab - 1];
    auto count = [&](uint64_t x) {
        if (x < pmin) return;                                    // precedes no positive
        int lo = 0, hi = n_lab;                                  // b = #{i : pkey[i] > x}  (x >= pmin: b <= n_lab - 1)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pkey[mid] > x) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&hist[lo], 1);
    };
    if (VEC) {
        // four 16-B loads in flight per thread (the loop is latency-bound otherwise: few rows, two workgroups per CU), and a float compare
        // in front of the key compare: v < the smallest positive SCORE implies key < pmin (neither a NaN)
        const float4* r4 = reinterpret_cast<const float4*>(r);
        const float smin = key_score(pmin);
        const int64_t n4 = N >> 2;
        auto take4 = [&](int64_t j4, const float4& v) {
            if (v.x < smin && v.y < smin && v.z < smin && v.w < smin) return;
            count(rank_key(v.x, (uint32_t)(4 * j4)));
            count(rank_key(v.y, (uint32_t)(4 * j4 + 1)));
            count(rank_key(v.z, (uint32_t)(4 * j4 + 2)));
            count(rank_key(v.w, (uint32_t)(4 * j4 + 3)));
        };
        int64_t j4 = tid;
        for (; j4 + 3 * NT < n4; j4 += 4 * NT) {
            const float4 v0 = r4[j4], v1 = r4[j4 + NT], v2 = r4[j4 + 2 * NT], v3 = r4[j4 + 3 * NT];
            take4(j4, v0); take4(j4 + NT, v1); take4(j4 + 2 * NT, v2); take4(j4 + 3 * NT, v3);
        }
        for (; j4 < n4; j4 += NT) take4(j4, r4[j4]);
    } else {
        for (int64_t j = tid; j < N; j += NT) count(rank_key(r[j], (uint32_t)j));
    }
    __syncthreads();
    // (3) inclusive prefix sum of H over the n_lab buckets -> rank of the i-th positive = prefix - 1.  Thread t owns the EPT consecutive
    // buckets t * EPT ...; wave totals through LDS.
    constexpr int EPT = (AP_MAXP + NT - 1) / NT;
    int loc[EPT], tsum = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid * EPT + e;
        loc[e] = (i < n_lab) ? hist[i] : 0;
        tsum += loc[e];
    }
    int incl = tsum;                                             // inclusive scan over the lanes of the wave
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wsum[tid >> 6] = incl;
    __syncthreads();
    int before = incl - tsum;
    for (int w = 0; w < (tid >> 6); ++w) before += wsum[w];
    __syncthreads();                                             // every hist[] value is in registers: the ranks may overwrite them
    int skipped = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int i = tid * EPT + e;
        before += loc[e];
        if (i < n_lab) {
            hist[i] = before - 1;                                // the positive itself sits in bucket i
            skipped += (before - 1 < kth - 1) ? 1 : 0;          // the first kth-1 ranks are skipped entirely (a prefix of the rank order)
        }
    }
    if (skipped) atomicAdd(&skip_s, skipped);
    __syncthreads();
    // (4) terms in rank order
    double* terms = reinterpret_cast<double*>(pkey);             // the keys are no longer needed
    const int i0 = skip_s;
    const double dn = (double)n_pos;
    __syncthreads();
    for (int i = i0 + tid; i < n_lab; i += NT) {
        const int rp = hist[i], h = i - i0;
        const int64_t j = rp - (kth - 1);
        const double recall = (double)(h + 1) / dn, old_recall = (double)h / dn;
        const double precision = (double)(h + 1) / ((double)j + 1.0);
        const double old_precision = (j == 0) ? 1.0 : (double)h / (double)j;
        terms[h] = (recall - old_recall) * ((old_precision + precision) / 2.0);
    }
    __syncthreads();
    if (tid == 0) {
        double ap = 0.0;
        for (int h = 0; h < n_lab - i0; ++h) ap += terms[h];     // ONE thread, rank order: the order of the reference's loop, hence the same float64 bits
        ap_out[row] = ap;
    }
}

// Per-row label-masked sums: out[row*2] = sum_j sim[row][j] * [glab[j]==qlab[row]], out[row*2+1] = sum_j sim[row][j]
__global__ __launch_bounds__(RK_THREADS) void masked_row_sums_kernel(const float* __restrict__ sim, int64_t N,
                                                                     const int32_t* __restrict__ qlab,
                                                                     const int32_t* __restrict__ glab, double* __restrict__ out) {
    __shared__ double red[2][RK_THREADS / 64];
    const int64_t row = blockIdx.x;
    const int32_t q = qlab[row];
    double sp = 0.0, sa = 0.0;
    for (int64_t j = threadIdx.x; j < N; j += RK_THREADS) {
        const double v = (double)sim[row * N + j];
        sa += v;
        if (glab[j] == q) sp += v;
    }
    sp = wave_sum(sp); sa = wave_sum(sa);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = sp; red[1][threadIdx.x >> 6] = sa; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0.0, b = 0.0;
        for (int i = 0; i < RK_THREADS / 64; ++i) { a += red[0][i]; b += red[1][i]; }
        out[row * 2] = a; out[row * 2 + 1] = b;
    }
}

static int64_t pow2_at_least(int64_t v) { int64_t p = 2; while (p < v) p <<= 1; return p; }

// ---- database-side augmentation (DBA), reference test/instance_avg.py:7-33 ------------------------------------------------------
// One 256-thread workgroup per gallery item i.  Only the items of i's own instance (same label: a run of `order`, ascending item index
// inside the run) are ever scored -- the reference builds the whole N x N matrix and then overwrites every other entry with -2.
//   1. score: thread t owns member t (t + 256, ...): acc = fmaf(E[i][d], E[m][d], acc) for d = 0..D-1 from +0 -- the canonical chain of
//      isx_cosine_sim, bit for bit; E[i] is read from LDS (broadcast);
//   2. rank: canonical keys (score desc, item index asc) sorted in LDS (<= 1024 members);
//   3. aggregate in the REFERENCE's order: agg = E[i]; agg += E[best_j] * w_j for j = 0, 1, ... with w_j = (nn - j) / float(nn + 1)
//      (a double quotient rounded to fp32, as Python's) -- sequential per element, so the sums equal the reference loop's bit for bit;
//   4. out = agg / (|agg| + 1e-10): eps OUTSIDE the norm here (:31).
constexpr int DBA_MAX_GROUP = 1024;

__global__ __launch_bounds__(256) void dba_group_kernel(const float* __restrict__ emb, int D, const int32_t* __restrict__ order,
                                                        const int32_t* __restrict__ grp_begin, const int32_t* __restrict__ grp_size, int k,
                                                        float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float dba_lds[];            // [D] E[i]  |  keys
    __shared__ float red[4];
    const int i = blockIdx.x;
    const int n = grp_size[i], begin = grp_begin[i];
    const float* ei = emb + (int64_t)i * D;
    float* o = out + (int64_t)i * D;
    int nn = n - 1;
    if (k >= 0 && k < nn) nn = k;
    if (nn <= 0) {                                                               // singleton instance or k = 0: kept as is (:21-23)
        for (int d = threadIdx.x; d < D; d += 256) o[d] = ei[d];
        return;
    }
    float* e_lds = dba_lds;
    uint64_t* keys = reinterpret_cast<uint64_t*>(dba_lds + ((D + 3) & ~3));
    int n2 = 2;
    while (n2 < n) n2 <<= 1;
    float* wts = reinterpret_cast<float*>(keys + n2);
    const bool vec4 = (D & 3) == 0 && ((uintptr_t)emb & 15) == 0;
    for (int d = threadIdx.x; d < D; d += 256) e_lds[d] = ei[d];
    __syncthreads();
    for (int t = threadIdx.x; t < n2; t += 256) {
        uint64_t key = 0;                                                        // padding and the item itself rank last
        if (t < n) {
            const int m = order[begin + t];
            if (m != i) {
                const float* em = emb + (int64_t)m * D;
                float acc = 0.0f;
                if (vec4) {                                                      // 16-B loads, the chain order untouched
                    const float4* a4 = reinterpret_cast<const float4*>(e_lds);
                    const float4* b4 = reinterpret_cast<const float4*>(em);
                    for (int d = 0; d < (D >> 2); ++d) {
                        const float4 a = a4[d], b = b4[d];
                        acc = fmaf(a.x, b.x, acc); acc = fmaf(a.y, b.y, acc); acc = fmaf(a.z, b.z, acc); acc = fmaf(a.w, b.w, acc);
                    }
                } else {
                    for (int d = 0; d < D; ++d) acc = fmaf(e_lds[d], em[d], acc);
                }
                key = rank_key(acc, (uint32_t)m);
            }
        }
        keys[t] = key;
    }
    bitonic_sort_desc<256>(keys, n2);
    for (int j = threadIdx.x; j < nn; j += 256) wts[j] = (float)((double)(nn - j) / (double)(nn + 1));
    __syncthreads();
    float ss = 0.0f;
    for (int d = threadIdx.x; d < D; d += 256) {
        float agg = e_lds[d];
        for (int j = 0; j < nn; ++j) agg = agg + emb[(int64_t)key_idx(keys[j]) * D + d] * wts[j];

        o[d] = agg;
        ss += agg * agg;
    }
    ss = block_sum<256>(ss, red);
    const float nrm = sqrtf(ss) + 1e-10f;
    for (int d = threadIdx.x; d < D; d += 256) o[d] = o[d] / nrm;               // each thread re-reads only what it wrote
}

}  // namespace isx

using namespace isx;

ISX_API size_t isx_rank_full_workspace(int64_t M, int64_t N) {
    if (M <= 0 || N <= 0) return 256;
    const int64_t N2 = pow2_at_least(N);
    if (N2 <= RK_TILE) return 256;
    return (size_t)M * (size_t)N2 * 8;
}

ISX_API int isx_rank_full(const float* sim, int64_t M, int64_t N, int64_t* ranked, void* ws, size_t ws_bytes,
                          isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && N <= 0x7FFFFFFFll && M < 65536 * 32768ll, "isx_rank_full: bad shape M=%lld N=%lld", (long long)M, (long long)N);
    if (M == 0 || N == 0) return ISX_OK;
    ISX_REQUIRE(sim && ranked, "isx_rank_full: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const int64_t N2 = pow2_at_least(N);
    if (N2 <= RK_TILE) {
        ISX_REQUIRE(M < (1ll << 31), "isx_rank_full: too many rows");
        hipLaunchKernelGGL(rank_small_kernel, dim3((unsigned)M), dim3(RK_THREADS), (size_t)N2 * 8, st, sim, N, (int)N2, ranked);
        ISX_CHECK_LAUNCH("isx_rank_full(small)");
        return ISX_OK;
    }
    if (!ws || ws_bytes < (size_t)M * (size_t)N2 * 8 || ((uintptr_t)ws % 16) != 0) {
        isx_set_error("isx_rank_full: workspace of %zu bytes too small (need %zu)", ws_bytes, (size_t)M * (size_t)N2 * 8);
        return ISX_ERR_WORKSPACE;
    }
    uint64_t* keys = (uint64_t*)ws;
    // grid.y is limited to 65535: walk the rows in slabs
    for (int64_t r0 = 0; r0 < M; r0 += 32768) {
        const unsigned rows = (unsigned)(M - r0 < 32768 ? M - r0 : 32768);
        const float* s = sim + r0 * N;
        uint64_t* kk = keys + r0 * N2;
        const dim3 tgrid((unsigned)(N2 / RK_TILE), rows);
        hipLaunchKernelGGL(rank_tile_sort_kernel, tgrid, dim3(RK_THREADS), 0, st, s, N, N2, kk);
        for (int64_t size = 2 * RK_TILE; size <= N2; size <<= 1) {
            for (int64_t stride = size >> 1; stride >= RK_TILE; stride >>= 1)
                hipLaunchKernelGGL(rank_global_step_kernel, dim3((unsigned)(N2 / 2 / RK_THREADS), rows), dim3(RK_THREADS), 0, st, kk, N2, size, stride);
            hipLaunchKernelGGL(rank_tile_finish_kernel, tgrid, dim3(RK_THREADS), 0, st, kk, N2, size);
        }
        hipLaunchKernelGGL(rank_emit_kernel, dim3((unsigned)((N + RK_THREADS - 1) / RK_THREADS), rows), dim3(RK_THREADS), 0, st, kk, N, N2, ranked + r0 * N);
    }
    ISX_CHECK_LAUNCH("isx_rank_full");
    return ISX_OK;
}

ISX_API int isx_average_precision(const int64_t* ranked, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab,
                                  int kth, double* ap, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && kth >= 1 && M < (1ll << 31), "isx_average_precision: bad shape M=%lld N=%lld kth=%d", (long long)M, (long long)N, kth);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(qlab && ap && ((ranked && glab) || N == 0), "isx_average_precision: null pointer");
    hipLaunchKernelGGL(average_precision_kernel, dim3((unsigned)M), dim3(RK_THREADS), 0, (hipStream_t)stream, ranked, N, qlab, glab, kth, ap);
    ISX_CHECK_LAUNCH("isx_average_precision");
    return ISX_OK;
}

ISX_API int isx_average_precision_sim(const float* sim, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab, int kth,
                                      double* ap, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && kth >= 1 && M < (1ll << 31) && N <= 0xFFFFFFFFll, "isx_average_precision_sim: bad shape M=%lld N=%lld kth=%d", (long long)M, (long long)N, kth);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(qlab && ap && ((sim && glab) || N == 0), "isx_average_precision_sim: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const bool vec = (N % 4 == 0) && ((((uintptr_t)sim | (uintptr_t)glab) % 16) == 0);
    const bool few_rows = M < 8192 && N >= 32768;                // fewer workgroups than the chip holds AND long rows: more threads per row
    if (few_rows && vec) hipLaunchKernelGGL((average_precision_sim_kernel<1024, true>), dim3((unsigned)M), dim3(1024), 0, st, sim, N, qlab, glab, kth, ap);
    else if (few_rows) hipLaunchKernelGGL((average_precision_sim_kernel<1024, false>), dim3((unsigned)M), dim3(1024), 0, st, sim, N, qlab, glab, kth, ap);
    else if (vec) hipLaunchKernelGGL((average_precision_sim_kernel<RK_THREADS, true>), dim3((unsigned)M), dim3(RK_THREADS), 0, st, sim, N, qlab, glab, kth, ap);
    else hipLaunchKernelGGL((average_precision_sim_kernel<RK_THREADS, false>), dim3((unsigned)M), dim3(RK_THREADS), 0, st, sim, N, qlab, glab, kth, ap);
    ISX_CHECK_LAUNCH("isx_average_precision_sim");
    return ISX_OK;
}

ISX_API int isx_masked_sums(const float* sim, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab, double* out,
                            isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && M < (1ll << 31), "isx_masked_sums: bad shape M=%lld N=%lld", (long long)M, (long long)N);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(qlab && out && ((sim && glab) || N == 0), "isx_masked_sums: null pointer");
    hipLaunchKernelGGL(masked_row_sums_kernel, dim3((unsigned)M), dim3(RK_THREADS), 0, (hipStream_t)stream, sim, N, qlab, glab, out);
    ISX_CHECK_LAUNCH("isx_masked_sums");
    return ISX_OK;
}

// DBA over same-instance groups (reference test/instance_avg.py:7-33).  order: the item indices sorted by (label, index); item i's instance is
// order[grp_begin[i] .. grp_begin[i] + grp_size[i]).  Groups above 1024 members are refused (the caller walks those label blocks with
// isx_cosine_sim + isx_rank_full).
ISX_API int isx_dba_groups(const float* emb, int64_t N, int D, const int32_t* order, const int32_t* grp_begin, const int32_t* grp_size,
                           int max_group, int k, float* out, isx_stream_t stream) {
    ISX_REQUIRE(N >= 0 && N < (1ll << 31) && D > 0 && D <= 16384 && max_group >= 0, "isx_dba_groups: bad shape N=%lld D=%d", (long long)N, D);
    ISX_REQUIRE(max_group <= DBA_MAX_GROUP, "isx_dba_groups: an instance of %d items exceeds the %d the kernel ranks in LDS", max_group, DBA_MAX_GROUP);
    if (N == 0) return ISX_OK;
    ISX_REQUIRE(emb && order && grp_begin && grp_size && out && out != emb, "isx_dba_groups: null or aliased pointer");
    int n2 = 2;
    while (n2 < max_group) n2 <<= 1;
    const size_t lds = (size_t)((D + 3) & ~3) * 4 + (size_t)n2 * 12;          // E[i] | keys | weights
    if (lds > 48 * 1024) {                                                        // wide descriptors: raise the dynamic-LDS limit of this kernel (160 KB per CU)
        ISX_REQUIRE(lds <= 150 * 1024, "isx_dba_groups: D=%d with instances of %d items needs %zu B of LDS", D, max_group, lds);
        if (hipFuncSetAttribute((const void*)dba_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
            isx_set_error("isx_dba_groups: cannot raise the dynamic LDS limit to %zu B", lds);
            return ISX_ERR_HIP;
        }
    }
    hipLaunchKernelGGL(dba_group_kernel, dim3((unsigned)N), dim3(256), lds, (hipStream_t)stream, emb, D, order, grp_begin, grp_size, k, out);
    ISX_CHECK_LAUNCH("isx_dba_groups");
    return ISX_OK;
}
