// stream1x1.hip -- the Cin = 64 1x1 convolutions of the first ResNet stage at 56x56 (64->64, 64->256 + residual), written as a
// STREAMING kernel.  These layers move 0.5-2.3 KB per pixel for 8-33 kFLOP: they are HBM-bound, and the tiled GEMM of cosine.hip
// (isx_conv1x1_nhwc's general path) holds them at ~4.0 TB/s because a tile's loads, MFMAs and stores follow each other inside
// a workgroup and only other workgroups overlap them.  Here one persistent 512-thread workgroup per CU walks the pixel tiles:
//   * the weights live in REGISTERS for the whole kernel (wave w owns 32 output channels: 32 VGPRs = the B operands of the 32
//     v_mfma_f32_32x32x2_f32 steps per 64 input channels: 32 or 128 VGPRs), the bias in one more;
//   * pixel tiles (64 or 128 pixels x 64 channels) arrive by LDS-DMA (global_load_lds_dwordx4) into a three-stage ring, two tiles
//     ahead; the residual of a tile is fetched into registers (buffer loads) before its MFMAs; stores are never waited for (one
//     counted s_waitcnt vmcnt(#stores) per tile retires everything older than the stores just issued);
//   * A operands: ds_read_b128 of the XOR-swizzled [pixel][16 x 16 B] image (lanes 0-31 chunk 2p, lanes 32-63 chunk 2p + 1) + two
//     v_permlane32_swap = the operands of four consecutive MFMA steps; k order untouched.
// Arithmetic = the general path's: acc = k-ordered fp32 fma chain from +0, y = act(acc + bias (+ residual)): bit-identical results
// (tests/test_gpu_parity.py::test_conv1x1_stream_equals_general).  Reference call sites: as isx_conv1x1_nhwc (conv.hip).
#include <stdlib.h>

#include "isx_internal.hpp"

namespace isx {

typedef float s1_f32x16 __attribute__((ext_vector_type(16)));
typedef float s1_f32x4 __attribute__((ext_vector_type(4)));

// NB = Cout / 32 (2, 4 or 8): wave w owns output-channel block w % NB and pixel group w / NB; PW = 32-pixel row blocks per wave;
// KC = Cin / 64: the K extent is walked in chunks of 64 channels, one ring stage per (pixel tile, chunk) STEP, accumulators live
// across the chunks of a tile and the epilogue runs after the last one.
// Residual loads and output stores go through BUFFER instructions: a wave-uniform descriptor of the tile's rows (SGPRs), one constant
// 32-bit lane offset, row offsets as SGPR / immediate offsets -- no per-element address arithmetic and no per-element edge tests
// (rows past M fall outside the descriptor: loads return 0, stores are dropped by the hardware).
template <int NB, int PW, int KC, bool RES, bool RELU>
__global__ __launch_bounds__(512) void conv1x1_stream_kernel(const float* __restrict__ x, int64_t M, const float* __restrict__ w,
                                                             const float* __restrict__ bias, const float* __restrict__ res,
                                                             float* __restrict__ y, int64_t ntiles) {
    constexpr int NG = 8 / NB, PT = NG * PW * 32, N = NB * 32, K = 64 * KC;
    constexpr int STAGE_B = PT * 256, NST = 3;
    constexpr int NDMA = PT / 4 / 8;                       // DMA instructions (1 KiB = 4 pixel rows x 64 channels each) per wave and step
    constexpr int NSTORE = 16 * PW;
    constexpr int ROW_B = N * 4;                           // bytes per output row
    __shared__ __attribute__((aligned(1024))) char lds[NST * STAGE_B];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cb = wave % NB, pg = wave / NB;
    const int l31 = lane & 31, half = lane >> 5;
    const int cout = cb * 32 + l31;

    // weights of this wave's 32 output channels as MFMA B operands: step s holds w[cout][2 s + half]
    float wreg[32 * KC];
#pragma unroll
    for (int s = 0; s < 32 * KC; ++s) wreg[s] = w[(int64_t)cout * K + 2 * s + half];
    const float bias_v = bias[cout];

    // DMA: instruction q of this wave fills image rows 4 (wave * NDMA + q) + lane / 16; slot lane % 16 holds chunk slot ^ (row & 15)
    int drow[NDMA], dchunk[NDMA];
#pragma unroll
    for (int q = 0; q < NDMA; ++q) {
        drow[q] = 4 * (wave * NDMA + q) + (lane >> 4);
        dchunk[q] = ((lane & 15) ^ (drow[q] & 15)) << 2;   // in floats
    }
    auto issue_x = [&](int64_t tile, int kc, int stage) {
        const int64_t p0 = tile * PT;
#pragma unroll
        for (int q = 0; q < NDMA; ++q) {
            int64_t p = p0 + drow[q];
            p = p < M ? p : M - 1;                          // rows past the end are never stored
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(x + p * K + kc * 64 + dchunk[q]),
                                             (__attribute__((address_space(3))) void*)(lds + stage * STAGE_B + (wave * NDMA + q) * 1024), 16, 0, 0);
        }
    };
    // A operand reads: pixel row r = (pg * PW + i) * 32 + l31 of the tile, chunk pair p: chunk 2p + half, swizzled by r & 15 = l31 & 15
    int a_ad[8];
#pragma unroll
    for (int p = 0; p < 8; ++p) a_ad[p] = (pg * PW * 32 + l31) * 256 + (((2 * p + half) ^ (l31 & 15)) << 4);

    // acc[i][e] is tile pixel (pg * PW + i) * 32 + (e & 3) + 8 (e >> 2) + 4 half, channel cout: lane offset inside the tile's rows
    const int lane_off = (pg * PW * 32 + 4 * half) * ROW_B + cout * 4;
    auto tile_rsrc = [&](const float* base, int64_t tile) {
        // descriptor of the tile's PT rows, clipped at row M (wave-uniform by construction: kernel arguments and the tile index only)
        const int64_t p0 = tile * PT;
        const int64_t rows = (M - p0) < PT ? (M - p0) : PT;
        const float* b0 = base + p0 * N;
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)b0);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)b0 >> 32));
        const unsigned nbytes = __builtin_amdgcn_readfirstlane((unsigned)(rows * ROW_B));
        return __builtin_amdgcn_make_buffer_rsrc((void*)(((uintptr_t)hi << 32) | lo), 0, (int)nbytes, 0x00020000);
    };

    // steps u = 0, 1, ... of this workgroup: tile t0 + (u / KC) G, chunk u % KC; stage u % 3; DMA two steps ahead
    const int64_t t0 = blockIdx.x, G = gridDim.x;
    const int64_t my_tiles = t0 < ntiles ? (ntiles - t0 + G - 1) / G : 0;
    const int64_t nsteps = my_tiles * KC;
    auto issue_step = [&](int64_t u, int stage) { issue_x(t0 + (u / KC) * G, (int)(u % KC), stage); };
    if (nsteps > 0) issue_step(0, 0);
    if (nsteps > 1) issue_step(1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int stage = 0;
    int64_t u = 0;
    for (int64_t t = t0; t < ntiles; t += G) {
        float resv[PW][16];
        s1_f32x16 acc[PW];
#pragma unroll
        for (int i = 0; i < PW; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.0f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc, ++u) {
            // prefetch: the step two ahead goes into the stage every wave finished reading before the last barrier
            const bool pre = (u + 2 < nsteps);
            if (pre) issue_step(u + 2, stage == 0 ? 2 : stage - 1);
            if (RES && kc == KC - 1) {
                // residual of this tile: in flight during the MFMAs of its last chunk
                const auto rr = tile_rsrc(res, t);
#pragma unroll
                for (int i = 0; i < PW; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        resv[i][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rr, lane_off + (e & 3) * ROW_B, (i * 32 + 8 * (e >> 2)) * ROW_B, 0));
            }
            const char* sb = lds + stage * STAGE_B;
#pragma unroll
            for (int i = 0; i < PW; ++i) {
                s1_f32x4 a[8];
#pragma unroll
                for (int p = 0; p < 8; ++p) a[p] = *reinterpret_cast<const s1_f32x4*>(sb + a_ad[p] + i * 8192);
#pragma unroll
                for (int p = 0; p < 8; ++p) {
                    // (__builtin_amdgcn_permlane32_swap of hipcc 7.2 returns element 0 twice: inline asm.)  hipcc pads nothing inside or
                    // around an asm statement: the wait states v_permlane32_swap needs after a VALU write of its operands (the register
                    // copies hipcc places in front of the statement; without the leading s_nop 1 the swap read stale registers) and
                    // before the MFMAs that read its results are written out
                    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 1"
                                 : "+v"(a[p].x), "+v"(a[p].y), "+v"(a[p].z), "+v"(a[p].w));
                    // x = k 8p, 8p+1 | z = 8p+2, 8p+3 | y = 8p+4, 8p+5 | w = 8p+6, 8p+7   (lower | upper lanes), k inside the chunk
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p].x, wreg[32 * kc + 4 * p + 0], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p].z, wreg[32 * kc + 4 * p + 1], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p].y, wreg[32 * kc + 4 * p + 2], acc[i], 0, 0, 0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[p].w, wreg[32 * kc + 4 * p + 3], acc[i], 0, 0, 0);
                }
            }
            if (kc == KC - 1) {
                const auto yr = tile_rsrc(y, t);
#pragma unroll
                for (int i = 0; i < PW; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        float v = acc[i][e] + bias_v;
                        if (RES) v += resv[i][e];
                        if (RELU) v = fmaxf(v, 0.0f);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, lane_off + (e & 3) * ROW_B, (i * 32 + 8 * (e >> 2)) * ROW_B, 0);
                    }
            }
            // The youngest operations -- this tile's stores (last chunk only) and, before them, the DMA issued at the top of this
            // step (two steps ahead) -- may stay in flight; everything older, i.e. the DMA of the NEXT step, is retired.  The barrier
            // makes that step's pixels visible to every wave and frees this stage for the DMA issued two steps from now.  (Where no
            // DMA was issued the wait retires fewer operations than it could, never too few.)
            if (u + 1 < nsteps) {
                if (kc == KC - 1) {
                    if (pre) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORE + NDMA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NSTORE) : "memory");
                } else {
                    if (pre) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NDMA) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            }
            stage = stage == NST - 1 ? 0 : stage + 1;
        }
    }
}

template <int NB, int PW, int KC>
static int launch_stream(const float* x, int64_t M, const float* w, const float* bias, const float* res, int relu, float* y, hipStream_t st) {
    constexpr int PT = (8 / NB) * PW * 32;
    const int64_t ntiles = (M + PT - 1) / PT;
    const dim3 grid((unsigned)(ntiles < 256 ? ntiles : 256)), block(512);       // one persistent workgroup per CU
    if constexpr (KC == 1) {
        if (res && relu) { hipLaunchKernelGGL((conv1x1_stream_kernel<NB, PW, KC, true, true>), grid, block, 0, st, x, M, w, bias, res, y, ntiles); ISX_CHECK_LAUNCH("conv1x1_stream"); return ISX_OK; }
        if (res) { hipLaunchKernelGGL((conv1x1_stream_kernel<NB, PW, KC, true, false>), grid, block, 0, st, x, M, w, bias, res, y, ntiles); ISX_CHECK_LAUNCH("conv1x1_stream"); return ISX_OK; }
    }
    if (relu) hipLaunchKernelGGL((conv1x1_stream_kernel<NB, PW, KC, false, true>), grid, block, 0, st, x, M, w, bias, res, y, ntiles);
    else hipLaunchKernelGGL((conv1x1_stream_kernel<NB, PW, KC, false, false>), grid, block, 0, st, x, M, w, bias, res, y, ntiles);
    ISX_CHECK_LAUNCH("conv1x1_stream");
    return ISX_OK;
}

// true when the streaming kernel covers the shape (the caller falls back to the tiled GEMM otherwise): the Cin = 64 layers of the first
// ResNet stage at 56x56 -- 64->64, 64->256 (+ residual) -- with enough pixels to fill the chip.  (Measured and NOT dispatched: the
// KC = 4 instantiations for 256->64 / 256->128 -- 128 weight registers per lane, four barriers per tile -- ran 1.12 / 1.92 ms against
// 1.01 / 1.88 ms for the tiled GEMM at B = 1024; a 128 -> 512 (+ residual) instantiation -- KC = 2, two column halves per pixel tile on
// workgroup pairs of one XCD -- ran 1.09 ms against 1.02 ms: with MFMA and HBM time balanced the per-step barrier of this kernel costs
// more than the tiled GEMM's four independent workgroups per CU.)
bool conv1x1_stream_applicable(int64_t M, int Cin, int Cout, const float* x, const float* res) {
    (void)res;
    return Cin == 64 && (Cout == 64 || Cout == 256) && M >= 16384 && ((((uintptr_t)x) & 15) == 0);
}

int launch_conv1x1_stream(const float* x, int64_t M, const float* w, int Cin, int Cout, const float* bias, const float* res, int relu, float* y,
                          hipStream_t st) {
    (void)Cin;
    if (Cout == 256) return launch_stream<8, 2, 1>(x, M, w, bias, res, relu, y, st);
    return launch_stream<2, 1, 1>(x, M, w, bias, res, relu, y, st);
}

}  // namespace isx
