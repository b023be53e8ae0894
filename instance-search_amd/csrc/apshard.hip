// Average precision of a query against a gallery SHARDED by rows over several GPUs (reference utils/metrics.py:25-45 walks the
// full ranked list of one score row).  AP only depends on the ranks of the positives (rank.hip, average_precision_sim_kernel), and the rank of
// a positive is a COUNT -- the gallery keys above it -- which adds over shards:
//
//   1. isx_ap_shard_positives   per shard: the canonical keys (score, GLOBAL index) of its positives of every query      -> all-gather
//   2. isx_ap_shard_hist        per shard: the gathered keys of a query sorted (descending); one pass over the shard's score row counts, for
//                               every gallery key x >= the smallest positive, the bucket b(x) = #{positives > x}            -> all-reduce (sum)
//   3. isx_ap_from_hist         anywhere:  rank of the i-th positive = prefix sum of the summed histogram - 1; the reference's float64 terms
//                               added in rank order by one thread
//
// With ONE shard the three steps are average_precision_sim_kernel cut at its two synchronisation points: the result is the same float64 bits
// as the unsharded kernels and the reference's loop, for any number of shards.  Integer work between the steps: exact whatever the order of
// the all-reduce.  Up to AP_MAXP positives per query over all shards (more: ap = -1, as in the one-GPU kernel).
#include "isx_internal.hpp"

namespace isx {

constexpr int APS_MAXP = 2048;           // == AP_MAXP of rank.hip
constexpr int APS_T = 256;

// keys[row][slot]: the shard's positives of the query, in arbitrary order, 0 = empty; count[row] = how many there were (may exceed cap)
__global__ __launch_bounds__(APS_T) void ap_shard_positives_kernel(const float* __restrict__ sim, int64_t N, int64_t idx_base,
                                                                   const int32_t* __restrict__ qlab, const int32_t* __restrict__ glab, int cap,
                                                                   uint64_t* __restrict__ keys, int32_t* __restrict__ count) {
    __shared__ int n_s;
    const int64_t row = blockIdx.x;
    const int32_t q = qlab[row];
    const float* r = sim + row * N;
    uint64_t* k = keys + row * cap;
    if (threadIdx.x == 0) n_s = 0;
    for (int p = threadIdx.x; p < cap; p += APS_T) k[p] = 0ull;
    __syncthreads();
    for (int64_t j = threadIdx.x; j < N; j += APS_T)
        if (glab[j] == q) {
            const int slot = atomicAdd(&n_s, 1);
            if (slot < cap) k[slot] = rank_key(r[j], (uint32_t)(idx_base + j));
        }
    __syncthreads();
    if (threadIdx.x == 0) count[row] = n_s;
}

// keys_all: (M, W) gathered keys of every shard (0 = empty slot); hist: (M, APS_MAXP) this shard's bucket counts (zero beyond the query's
// positives).  VEC: 16-B loads of the score row (N % 4 == 0, 16-B aligned rows).
template <bool VEC>
__global__ __launch_bounds__(APS_T) void ap_shard_hist_kernel(const float* __restrict__ sim, int64_t N, int64_t idx_base,
                                                              const uint64_t* __restrict__ keys_all, int W, int32_t* __restrict__ hist_out) {
    __shared__ __attribute__((aligned(16))) uint64_t pkey[APS_MAXP];
    __shared__ int hist[APS_MAXP];
    __shared__ int n_s;
    const int tid = threadIdx.x;
    const int64_t row = blockIdx.x;
    const float* r = sim + row * N;
    int32_t* ho = hist_out + row * APS_MAXP;
    if (tid == 0) n_s = 0;
    for (int p = tid; p < APS_MAXP; p += APS_T) hist[p] = 0;
    __syncthreads();
    const uint64_t* ka = keys_all + row * (int64_t)W;
    for (int p = tid; p < W; p += APS_T) {
        const uint64_t key = ka[p];
        if (key) {
            const int slot = atomicAdd(&n_s, 1);
            if (slot < APS_MAXP) pkey[slot] = key;
        }
    }
    __syncthreads();
    const int n_lab = n_s;
    if (n_lab == 0 || n_lab > APS_MAXP) {                       // nothing to rank / over the cap (isx_ap_from_hist reports it): an all-zero histogram
        for (int p = tid; p < APS_MAXP; p += APS_T) ho[p] = 0;
        return;
    }
    int n2 = 2;
    while (n2 < n_lab) n2 <<= 1;
    for (int p = n_lab + tid; p < n2; p += APS_T) pkey[p] = 0ull;
    __syncthreads();
    bitonic_sort_desc<APS_T>(pkey, n2);
    __syncthreads();
    const uint64_t pmin = pkey[n_lab - 1];
    auto count = [&](uint64_t x) {
        if (x < pmin) return;
        int lo = 0, hi = n_lab;                                  // b = #{i : pkey[i] > x}
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (pkey[mid] > x) lo = mid + 1; else hi = mid;
        }
        atomicAdd(&hist[lo], 1);
    };
    if (VEC) {
        const float4* r4 = reinterpret_cast<const float4*>(r);
        const float smin = key_score(pmin);
        const int64_t n4 = N >> 2;
        for (int64_t j4 = tid; j4 < n4; j4 += APS_T) {
            const float4 v = r4[j4];
            if (v.x < smin && v.y < smin && v.z < smin && v.w < smin) continue;
            const int64_t g = idx_base + 4 * j4;
            count(rank_key(v.x, (uint32_t)g));
            count(rank_key(v.y, (uint32_t)(g + 1)));
            count(rank_key(v.z, (uint32_t)(g + 2)));
            count(rank_key(v.w, (uint32_t)(g + 3)));
        }
    } else {
        for (int64_t j = tid; j < N; j += APS_T) count(rank_key(r[j], (uint32_t)(idx_base + j)));
    }
    __syncthreads();
    for (int p = tid; p < APS_MAXP; p += APS_T) ho[p] = hist[p];
}

// hist: (M, APS_MAXP) summed over the shards; n_lab[row]: the query's positives over all shards.  One thread per row: prefix sum, terms, sum --
// the arithmetic of average_precision_sim_kernel steps (3) and (4).
__global__ __launch_bounds__(64) void ap_from_hist_kernel(const int32_t* __restrict__ hist, int ld, const int32_t* __restrict__ n_lab_all, int64_t M, int kth,
                                                          double* __restrict__ ap_out) {
    const int64_t row = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (row >= M) return;
    const int n_lab = n_lab_all[row];
    const int64_t n_pos = (int64_t)n_lab - (kth - 1);
    if (n_pos <= 0) { ap_out[row] = __longlong_as_double(0x7FF8000000000000ll); return; }
    if (n_lab > APS_MAXP || n_lab > ld) { ap_out[row] = -1.0; return; }
    const int32_t* h_ = hist + row * (int64_t)ld;
    const double dn = (double)n_pos;
    double ap = 0.0;
    int before = 0, hh = 0;
    for (int i = 0; i < n_lab; ++i) {
        before += h_[i];
        const int rp = before - 1;                               // 0-based rank of the i-th positive (itself sits in bucket i)
        if (rp < kth - 1) continue;                              // the first kth - 1 ranks are skipped entirely
        const int h = hh++;
        const int64_t j = rp - (kth - 1);
        const double recall = (double)(h + 1) / dn, old_recall = (double)h / dn;
        const double precision = (double)(h + 1) / ((double)j + 1.0);
        const double old_precision = (j == 0) ? 1.0 : (double)h / (double)j;
        ap += (recall - old_recall) * ((old_precision + precision) / 2.0);
    }
    ap_out[row] = ap;
}

}  // namespace isx

using namespace isx;

ISX_API int isx_ap_shard_max_positives(void) { return APS_MAXP; }

ISX_API int isx_ap_shard_positives(const float* sim, int64_t M, int64_t N, int64_t idx_base, const int32_t* qlab, const int32_t* glab, int cap,
                                   uint64_t* keys, int32_t* count, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && M < (1ll << 31) && idx_base >= 0 && idx_base + N <= 0xFFFFFFFFll && cap >= 1 && cap <= APS_MAXP,
                "isx_ap_shard_positives: bad shape M=%lld N=%lld idx_base=%lld cap=%d (global indices below 2^32, 1 <= cap <= %d)", (long long)M, (long long)N,
                (long long)idx_base, cap, APS_MAXP);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(qlab && keys && count && ((sim && glab) || N == 0), "isx_ap_shard_positives: null pointer");
    hipLaunchKernelGGL(ap_shard_positives_kernel, dim3((unsigned)M), dim3(APS_T), 0, (hipStream_t)stream, sim, N, idx_base, qlab, glab, cap, keys, count);
    ISX_CHECK_LAUNCH("isx_ap_shard_positives");
    return ISX_OK;
}

ISX_API int isx_ap_shard_hist(const float* sim, int64_t M, int64_t N, int64_t idx_base, const uint64_t* keys_all, int W, int32_t* hist, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && N >= 0 && M < (1ll << 31) && idx_base >= 0 && idx_base + N <= 0xFFFFFFFFll && W >= 1 && W <= (1 << 20),
                "isx_ap_shard_hist: bad shape M=%lld N=%lld idx_base=%lld W=%d", (long long)M, (long long)N, (long long)idx_base, W);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(keys_all && hist && (sim || N == 0), "isx_ap_shard_hist: null pointer");
    const bool vec = N % 4 == 0 && ((uintptr_t)sim % 16) == 0 && idx_base % 4 == 0;
    if (vec) hipLaunchKernelGGL((ap_shard_hist_kernel<true>), dim3((unsigned)M), dim3(APS_T), 0, (hipStream_t)stream, sim, N, idx_base, keys_all, W, hist);
    else hipLaunchKernelGGL((ap_shard_hist_kernel<false>), dim3((unsigned)M), dim3(APS_T), 0, (hipStream_t)stream, sim, N, idx_base, keys_all, W, hist);
    ISX_CHECK_LAUNCH("isx_ap_shard_hist");
    return ISX_OK;
}

ISX_API int isx_ap_from_hist(const int32_t* hist, int ld, const int32_t* n_lab, int64_t M, int kth, double* ap, isx_stream_t stream) {
    ISX_REQUIRE(M >= 0 && M < (1ll << 31) && kth >= 1 && ld >= 1, "isx_ap_from_hist: bad shape M=%lld ld=%d kth=%d", (long long)M, ld, kth);
    if (M == 0) return ISX_OK;
    ISX_REQUIRE(hist && n_lab && ap, "isx_ap_from_hist: null pointer");
    hipLaunchKernelGGL(ap_from_hist_kernel, dim3((unsigned)((M + 63) / 64)), dim3(64), 0, (hipStream_t)stream, hist, ld, n_lab, M, kth, ap);
    ISX_CHECK_LAUNCH("isx_ap_from_hist");
    return ISX_OK;
}
