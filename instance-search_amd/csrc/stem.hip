// stem.hip -- the ResNet stem as ONE kernel: conv 7x7 / stride 2 / padding 3 (3 -> 64 channels, BN folded into the weights) + bias + ReLU +
// MaxPool2d(3, stride 2, padding 1) on channels-last activations, from the normalised image straight to the (B, H/4, W/4, 64) map.
//
// Why a kernel of its own: as separate passes the stem writes and re-reads its (B, 112, 112, 64) fp32 convolution output (3.3 GB at
// B = 1024) and MIOpen's implicit GEMM zero-fills it first: 2.75 + 0.54 + 0.83 ms of the 70 ms bench step for 1.5 ms of matrix-core work.
// Here the convolution output never leaves the registers:
//   * one persistent 512-thread workgroup per image (or band of an image) walks DOWN the image in steps of 8 convolution rows x 112
//     columns: 896 pixels x 64 channels = 56 accumulator tiles of v_mfma_f32_32x32x2_f32, 7 per wave (wave w: channels 32 (w & 1) ..,
//     column blocks 7 (w >> 1) ..);
//   * the 21 input rows a step touches arrive by LDS-DMA (global_load_lds_dwordx4, exec-masked at the row end) into one of two LDS
//     buffers, a step ahead; rows keep a 4-pixel zero border, rows outside the image are zero-filled: padding needs no test;
//   * K is ordered (kh, kw, c) with the 21 values of a filter row padded to 22 (zero weight): the A operand of k-step s of filter row
//     kh is the LDS word at lane_base + kh * row + 2 s (+ 24 per column block): ONE base register, everything else immediate offsets;
//     the weights sit in LDS as [k][64] for the whole kernel (one ds_read per 7 MFMAs);
//   * an accumulator row block is 4 columns x 8 rows, ordered so that ONE lane holds, for its channel, the 8 rows of two adjacent
//     columns = four complete 2x2 pooling cells: the 3x3/2 max needs only the row above (carried in registers from the previous
//     step) and the column to the left (the other half-wave: ds_bpermute; across waves: 4 KB of LDS), then 28 dword stores per lane.
//   * images wider than 224 (region path: 384 / 448 wide) are walked in COLUMN BANDS of 112 convolution columns INSIDE the workgroup:
//     unit (step t, band cb) stages the 21 rows x 232 columns under its tile (the 4-pixel border now holds the neighbouring band's
//     pixels, or zeros at the image edge), the carry row of every band stays in registers, and the right-most pooled partial column of
//     band cb reaches band cb + 1 through an LDS slot of its own -- nothing is recomputed and nothing but the pooled map is stored.
// Arithmetic: fp32 fma chains over (kh, kw, c) ascending from +0 (padding taps contribute fma(0, w, acc) = acc), one per group of three filter rows
// (kh 0-2, 3-5, 6), summed in order: acc = ((0 + c0) + c1) + c2 (two-level sum, gemm_tile.hpp), y = max(acc + bias, 0),
// out = max over the window: bit-identical to oracle/isx_oracle.c::isxo_stem7x7_pool_nhwc (tests/test_gpu_parity.py).
// Reference: the torchvision ResNet stem (conv1, bn1, relu, maxpool) inside the `features` trunk built by model/ModelDefinition.py and
// split by model/nn_utils.py:56-71; run from model/siamese.py:20,107,151.
#include <stdlib.h>

#include "gemm_tile.hpp"

namespace isx {

constexpr int ST_RSF = 696;                    // floats per staged input row: 12 (4-pixel zero border) + 3 * 224 + 12
constexpr int ST_ROWS = 21;                    // input rows under 8 convolution rows: 2 * 7 + 7
constexpr int ST_BUF_F = ST_ROWS * ST_RSF;     // 58 464 B per buffer
constexpr int ST_K = 154;                      // 7 filter rows x (21 + 1 zero)
constexpr int ST_W_F = ST_K * 64;              // 39 424 B
constexpr int ST_X_F = 5 * 2 * 4 * 32;         // cross-wave column exchange: [column group (+ 2 band hand-over slots)][channel block][pooled row][channel]
constexpr int ST_NSTORE = 28;                  // output stores per lane and step
constexpr int ST_BANDW = 224;                  // input columns per column band (112 convolution columns, 56 pooled columns)
constexpr int ST_MAXCB = 4;                    // column bands per image
constexpr int ST_MAXW = ST_BANDW * ST_MAXCB;
constexpr int ST_ROWCHUNKS = ST_RSF / 4;       // 16-B chunks per staged row: 174

struct StemGeom { int H, W, Hc, Wc, Hp, Wp, steps, bands; };

// (add_tile_inplace, gemm_tile.hpp: a += t on the 16 registers of one MFMA tile IN PLACE.  As vector arithmetic hipcc put the sums into fresh scattered
// register pairs and turned the two old tiles into the next MFMA accumulators -- the file fragments and 135 VGPRs spill; the tied asm operand keeps the sum
// in its own registers: 230 VGPRs, no spill.  The caller provides the MFMA -> VALU wait states.)

// NCB: column bands per image (1: images up to 224 wide, the tile covers the whole width).
template <int NCB>
__global__ __launch_bounds__(512) void stem7x7_pool_kernel(const float* __restrict__ x, StemGeom g, const float* __restrict__ w_ohwi,
                                                           const float* __restrict__ bias, float* __restrict__ out) {
    __shared__ __attribute__((aligned(1024))) float lds[2 * ST_BUF_F + ST_W_F + ST_X_F];      // 161 472 B
    float* const in_lds = lds;
    float* const w_lds = lds + 2 * ST_BUF_F;
    float* const x_lds = w_lds + ST_W_F;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nblk = wave & 1, cg = wave >> 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int img = blockIdx.x / g.bands, band = blockIdx.x - img * g.bands;
    const int t_first = (int)((int64_t)band * g.steps / g.bands), t_end = (int)((int64_t)(band + 1) * g.steps / g.bands);
    const int t_start = t_first > 0 ? t_first - 1 : 0;            // a band below the top runs one step early for its carry row (no stores)
    const float* __restrict__ x_img = x + (int64_t)img * g.H * g.W * 3;
    const int nchunk = g.W * 3 / 4;                                // 16-B chunks per input row

    // ---- prologue: zero the input buffers (borders stay zero for the whole kernel), weights into LDS as [kh * 22 + kw * 3 + c][co]
    for (int e = tid; e < 2 * ST_BUF_F / 4; e += 512) reinterpret_cast<float4*>(in_lds)[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = tid; e < 64 * 147; e += 512) {
        const int co = e / 147, r = e - co * 147, kh = r / 21, kwc = r - kh * 21;
        w_lds[(kh * 22 + kwc) * 64 + co] = w_ohwi[e];
    }
    for (int e = tid; e < 7 * 64; e += 512) w_lds[((e >> 6) * 22 + 21) * 64 + (e & 63)] = 0.0f;
    __syncthreads();

    // input rows 16 t - 3 .. 16 t + 17, image columns 224 cb - 4 .. 224 cb + 227 of unit (t, cb) into buffer `buf`: 63 (row, 64-chunk part) items,
    // wave w takes items w, w + 8, ...  LDS chunk lc of a row holds image chunk 168 cb - 3 + lc.  One band: the chunks outside the image are never
    // written and stay zero from the prologue; several bands: every chunk of the unit is either fetched or zeroed (the buffers change bands).
    auto issue_dma = [&](int t, int cb, int buf) {
        const int r0 = 16 * t - 3;
        const int gc0 = 168 * cb - 3;
#pragma unroll
        for (int i8 = 0; i8 < 8; ++i8) {
            const int item = i8 * 8 + wave;
            if (item < 63) {
                const int row = item / 3, part = item - row * 3;
                const int ir = r0 + row;
                const int lc = part * 64 + lane;
                const int gc = gc0 + lc;
                float* dst_row = in_lds + buf * ST_BUF_F + row * ST_RSF;
                const bool in_cols = (unsigned)gc < (unsigned)nchunk && lc < ST_ROWCHUNKS;
                if ((unsigned)ir < (unsigned)g.H) {
                    if (in_cols)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x_img + (int64_t)ir * g.W * 3 + gc * 4),
                                                         (__attribute__((address_space(3))) void*)(dst_row + part * 256), 16, 0, 0);
                    else if (NCB > 1 && lc < ST_ROWCHUNKS)
                        reinterpret_cast<float4*>(dst_row)[lc] = make_float4(0.f, 0.f, 0.f, 0.f);
                } else if (NCB > 1 ? lc < ST_ROWCHUNKS : in_cols) {
                    reinterpret_cast<float4*>(dst_row)[lc] = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
        }
    };

    // A operand: MFMA row m of a column block is convolution row 2 (m >> 3) + ((m >> 1) & 1), column 4 b + 2 ((m >> 2) & 1) + (m & 1)
    const int am = l31;
    const int a_row = 2 * (am >> 3) + ((am >> 1) & 1), a_col = 28 * cg + 2 * ((am >> 2) & 1) + (am & 1);
    const int a_lane = a_row * 2 * ST_RSF + 6 * a_col + 3 + h;                  // + kh * ST_RSF + 2 s' + 24 b  (immediates)
    const float* const bp = w_lds + h * 64 + 32 * nblk + l31;                   // + (22 kh + 2 s') * 64

    const float bias_v = bias[32 * nblk + l31];
    float carry_all[NCB][7][2];                                                 // convolution row 8 t - 1 of this lane's two columns (after ReLU), per band
#pragma unroll
    for (int j = 0; j < NCB; ++j)
#pragma unroll
        for (int b = 0; b < 7; ++b) carry_all[j][b][0] = carry_all[j][b][1] = 0.0f;

    // output: descriptor of the image's pooled rows; lane offset of (column q = 56 cb + 14 cg + 2 b + h, channel); pooled row and band are the SGPR offset
    const int64_t out_img_bytes = (int64_t)g.Hp * g.Wp * 256;
    const auto out_rs = uniform_rsrc(out + (int64_t)img * g.Hp * g.Wp * 64, out_img_bytes);
    const auto null_rs = uniform_rsrc(out, 0);
    unsigned o_off[7];
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        const int q = 14 * cg + 2 * b + h;
        o_off[b] = q < g.Wp ? (unsigned)((q * 64 + 32 * nblk + l31) * 4) : 0x80000000u;
    }

    issue_dma(t_start, 0, 0);
    int buf = 0, t = t_start, cb = 0;
    bool first = true;
    for (;;) {
        // this unit's rows have landed (everything older than the previous unit's stores), and every wave is done with the other buffer
        if (first) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ST_NSTORE) : "memory");
        first = false;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int tn = t, cbn = cb + 1;
        if (cbn == NCB) { cbn = 0; ++tn; }
        if (tn < t_end) issue_dma(tn, cbn, buf ^ 1);

        float carry[7][2];
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            carry[b][0] = carry_all[0][b][0];
            carry[b][1] = carry_all[0][b][1];
#pragma unroll
            for (int j = 1; j < NCB; ++j) {
                carry[b][0] = cb == j ? carry_all[j][b][0] : carry[b][0];
                carry[b][1] = cb == j ? carry_all[j][b][1] : carry[b][1];
            }
        }
        if (NCB > 1) {
#pragma unroll
            for (int b = 0; b < 7; ++b) {
                const int q = 56 * cb + 14 * cg + 2 * b + h;
                o_off[b] = q < g.Wp ? (unsigned)(((14 * cg + 2 * b + h) * 64 + 32 * nblk + l31) * 4) : 0x80000000u;
            }
        }
        const int wc_left = g.Wc - 112 * cb;                                    // convolution columns of the image from this band's first one on

        // two-level sum (gemm_tile.hpp): the chain restarts after filter rows 2 and 5 (63 + 63 + 21 terms; the zero-weight slot that pads a filter
        // row to 22 adds fma(x, 0, acc) = acc) and the three chunk sums are added in order: ((0 + c0) + c1) + c2.  A second accumulator set next
        // to all seven tiles does not fit (2 x 112 VGPRs + 80): chunk 0 runs on all seven tiles as before and its chains BECOME the running sums
        // (0 + c0 = c0: a chain from +0 is never -0); chunks 1 and 2 then go through the k loop in groups of 3 + 2 + 2 column blocks with a
        // small temporary accumulator set per group, added element by element IN PLACE (v_pk_add_f32 on the register pairs of the tile).
        const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        f32x16 acc[7];
        const float* const ap = in_lds + buf * ST_BUF_F + a_lane;
        {
            float av[2][7], bv[2];
#pragma unroll
            for (int b = 0; b < 7; ++b) av[0][b] = ap[24 * b];
            bv[0] = bp[0];
#pragma unroll
            for (int s = 0; s < 33; ++s) {                                      // filter rows 0-2: 11 k-steps per row
                if (s + 1 < 33) {
                    const int kh = (s + 1) / 11, sp = (s + 1) % 11;
#pragma unroll
                    for (int b = 0; b < 7; ++b) av[(s + 1) & 1][b] = ap[kh * ST_RSF + 2 * sp + 24 * b];
                    bv[(s + 1) & 1] = bp[(22 * kh + 2 * sp) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < 7; ++b) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][b], bv[s & 1], s == 0 ? zero : acc[b], 0, 0, 0);
            }
        }
        auto later_chunks = [&](auto b0_, auto nb_) {
            constexpr int B0 = decltype(b0_)::value, NB = decltype(nb_)::value;
            f32x16 t[NB];
            float av[2][NB], bv[2];
#pragma unroll
            for (int b = 0; b < NB; ++b) av[1][b] = ap[3 * ST_RSF + 24 * (B0 + b)];            // s = 33: buffer 33 & 1
            bv[1] = bp[(22 * 3) * 64];
#pragma unroll
            for (int s = 33; s < 77; ++s) {
                if (s + 1 < 77) {
                    const int kh = (s + 1) / 11, sp = (s + 1) % 11;
#pragma unroll
                    for (int b = 0; b < NB; ++b) av[(s + 1) & 1][b] = ap[kh * ST_RSF + 2 * sp + 24 * (B0 + b)];
                    bv[(s + 1) & 1] = bp[(22 * kh + 2 * sp) * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int b = 0; b < NB; ++b) t[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s & 1][b], bv[s & 1], (s == 33 || s == 66) ? zero : t[b], 0, 0, 0);
                if (s == 65 || s == 76) {                                       // filter rows 3-5 / row 6 done
                    // the adds are inline asm: hipcc's hazard recognizer does not treat them as VALU reads of MFMA results, so the 18 wait states
                    // a 16-pass MFMA needs before a VALU read of its destination are spelled out (the compiler emits s_nop 15 + s_nop 1 itself
                    // in front of an ordinary v_pk_add_f32)
                    asm volatile("s_nop 15\n\ts_nop 2" ::: "memory");
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        if (B0 + b == 0) add_tile_inplace(acc[0], t[b]);
                        if (B0 + b == 1) add_tile_inplace(acc[1], t[b]);
                        if (B0 + b == 2) add_tile_inplace(acc[2], t[b]);
                        if (B0 + b == 3) add_tile_inplace(acc[3], t[b]);
                        if (B0 + b == 4) add_tile_inplace(acc[4], t[b]);
                        if (B0 + b == 5) add_tile_inplace(acc[5], t[b]);
                        if (B0 + b == 6) add_tile_inplace(acc[6], t[b]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        later_chunks(std::integral_constant<int, 0>(), std::integral_constant<int, 3>());
        later_chunks(std::integral_constant<int, 3>(), std::integral_constant<int, 2>());
        later_chunks(std::integral_constant<int, 5>(), std::integral_constant<int, 2>());

        // ---- epilogue: bias + ReLU, 3x3 / stride 2 max.  acc[b][4 gg + 2 r + c] = convolution row 8 t + 2 gg + r, column 112 cb + 28 cg + 4 b + 2 h + c
        float own[7][4], rgt[7][4];
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool ok = (8 * t + 2 * (e >> 2) + ((e >> 1) & 1) < g.Hc) && (28 * cg + 4 * b + 2 * h + (e & 1) < wc_left);
                v[e] = ok ? fmaxf(acc[b][e] + bias_v, 0.0f) : 0.0f;            // 0 = identity of the max (values are >= 0 after the ReLU)
            }
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float top0 = gg ? v[4 * gg - 2] : carry[b][0], top1 = gg ? v[4 * gg - 1] : carry[b][1];
                rgt[b][gg] = fmaxf(fmaxf(top1, v[4 * gg + 1]), v[4 * gg + 3]);
                own[b][gg] = fmaxf(fmaxf(fmaxf(top0, v[4 * gg]), v[4 * gg + 2]), rgt[b][gg]);
            }
#pragma unroll
            for (int j = 0; j < NCB; ++j) {
                carry_all[j][b][0] = (NCB == 1 || cb == j) ? v[14] : carry_all[j][b][0];
                carry_all[j][b][1] = (NCB == 1 || cb == j) ? v[15] : carry_all[j][b][1];
            }
        }
        // the column to the left: the other half-wave's right column (same block for h = 1, the previous block for h = 0); the last column group
        // of a band leaves its right column in slot 3 + (cb & 1) for the first group of the next band
        if (h) {
            const int slot = (NCB > 1 && cg == 3) ? 3 + (cb & 1) : cg;
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) x_lds[((slot * 2 + nblk) * 4 + gg) * 32 + l31] = rgt[6][gg];
        }
        float swp[7][4];
#pragma unroll
        for (int b = 0; b < 7; ++b)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg)
                swp[b][gg] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, rgt[b][gg])));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        float left0[4];
        const int lslot = cg ? cg - 1 : 3 + ((cb - 1) & 1);
        const bool has_left = cg || (NCB > 1 && cb > 0);
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) left0[gg] = has_left ? x_lds[((lslot * 2 + nblk) * 4 + gg) * 32 + l31] : 0.0f;
        const auto rs = t >= t_first ? out_rs : null_rs;
#pragma unroll
        for (int b = 0; b < 7; ++b)
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                const float left = h ? swp[b][gg] : (b ? swp[b > 0 ? b - 1 : 0][gg] : left0[gg]);
                const float y = fmaxf(own[b][gg], left);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, y), rs, o_off[b], (unsigned)(((4 * t + gg) * g.Wp + 56 * cb) * 256), 0);
            }
        t = tn; cb = cbn; buf ^= 1;
        if (t >= t_end) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace isx

// conv 7x7 / stride 2 / padding 3 (3 -> 64) + bias + ReLU + MaxPool2d(3, 2, 1) on a channels-last image batch.
ISX_API int isx_stem7x7_pool_nhwc(const float* x, int64_t B, int H, int W, const float* w_ohwi, const float* bias, float* out,
                                  isx_stream_t stream) {
    ISX_REQUIRE(B >= 0 && H > 0 && W > 0, "isx_stem7x7_pool_nhwc: bad shape B=%lld H=%d W=%d", (long long)B, H, W);
    ISX_REQUIRE(W % 4 == 0 && W <= isx::ST_MAXW, "isx_stem7x7_pool_nhwc: W=%d must be a multiple of 4 and <= %d", W, isx::ST_MAXW);
    ISX_REQUIRE(H <= 32768 && B < (1ll << 24), "isx_stem7x7_pool_nhwc: B=%lld H=%d too large", (long long)B, H);
    if (B == 0) return ISX_OK;
    ISX_REQUIRE(x && w_ohwi && bias && out && out != x, "isx_stem7x7_pool_nhwc: null or aliased pointer");
    ISX_REQUIRE((((uintptr_t)x) % 16) == 0, "isx_stem7x7_pool_nhwc: x must be 16-B aligned");
    isx::StemGeom g;
    g.H = H; g.W = W;
    g.Hc = (H - 1) / 2 + 1; g.Wc = (W - 1) / 2 + 1;
    g.Hp = (g.Hc - 1) / 2 + 1; g.Wp = (g.Wc - 1) / 2 + 1;
    g.steps = (g.Hc + 7) / 8;
    ISX_REQUIRE((int64_t)g.Hp * g.Wp * 256 < (1ll << 31), "isx_stem7x7_pool_nhwc: pooled map of one image above 2 GiB");
    int bands = 1;                                               // few images: split each into bands of steps so that every CU has work
    while (B * bands < 256 && bands * 4 <= g.steps) bands *= 2;
    g.bands = bands;
    const int ncb = (g.Wc + 111) / 112;                         // column bands, walked inside the workgroup
    const dim3 grid((unsigned)(B * bands));
    hipStream_t st = (hipStream_t)stream;
    if (ncb == 1) hipLaunchKernelGGL(isx::stem7x7_pool_kernel<1>, grid, dim3(512), 0, st, x, g, w_ohwi, bias, out);
    else if (ncb == 2) hipLaunchKernelGGL(isx::stem7x7_pool_kernel<2>, grid, dim3(512), 0, st, x, g, w_ohwi, bias, out);
    else if (ncb == 3) hipLaunchKernelGGL(isx::stem7x7_pool_kernel<3>, grid, dim3(512), 0, st, x, g, w_ohwi, bias, out);
    else hipLaunchKernelGGL(isx::stem7x7_pool_kernel<4>, grid, dim3(512), 0, st, x, g, w_ohwi, bias, out);
    ISX_CHECK_LAUNCH("isx_stem7x7_pool_nhwc");
    return ISX_OK;
}
