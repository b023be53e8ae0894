// Probe: what does an LDS operand read cost beside a back-to-back fp32 MFMA stream on the same SIMD?
//   mode 0: 8 waves (2 per SIMD), every wave: NM MFMAs then NR reads, repeated (same-wave interleave), no barriers
//   mode 1: waves 0-3 stream MFMAs only, waves 4-7 (their SIMD partners) issue LDS reads only
//   mode 2: waves 0-3 stream MFMAs only, waves 4-7 idle (MFMA rate alone);  mode 3: reads alone (waves 0-3 idle)
// Reports cycles per MFMA on the MFMA waves and cycles per read instruction on the reading waves (s_memtime, wave 0 / wave 4 of block 0).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int MODE, int RW, int MFMA16>   // RW: bytes per lane per read (4, 8, 16); MFMA16: 1 = v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(512) void probe(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<float*>(lds)[i] = (float)(i & 255) * 0.001f;
    __syncthreads();
    f32x16 acc[4];
    f32x4 acc4[8];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) acc4[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + lane * 0.001f, b = 0.5f;
    float sink = 0.f;
    const bool do_mfma = (MODE == 0) || ((MODE == 1 || MODE == 2) && wave < 4);
    const bool do_read = (MODE == 0) || (MODE == 1 && wave >= 4) || (MODE == 3 && wave >= 4);
    const char* base = lds + lane * 16 + (wave & 3) * 4096;       // conflict-free 16-B slots per lane
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (do_mfma) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MFMA16) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc4[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[i], 0, 0, 0);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
                }
            }
        }
        if (do_read) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (RW == 4) { float v; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)base), "n"(r * 1024 % 4096 + (r / 4) * 4)); sink += v; }
                else if (RW == 8) { f32x2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)base), "n"(r * 1024 % 4096 + (r / 4) * 8 % 16)); sink += v.x; }
                else { f32x4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"((unsigned)(size_t)base), "n"(r * 1024 % 4096)); sink += v.x; }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0) stamps[wave] = t1 - t0;
    float s = sink;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc4[i][0];
    if (s == 123.456f) out[tid] = s;
}

template <int MODE, int RW, int MFMA16>
static void run(const char* name, float* out, unsigned long long* st) {
    const int iters = 2000;
    hipLaunchKernelGGL((probe<MODE, RW, MFMA16>), dim3(256), dim3(512), 0, 0, out, st, iters);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((probe<MODE, RW, MFMA16>), dim3(256), dim3(512), 0, 0, out, st, iters);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
    const double nm = iters * 16.0 * (MFMA16 ? 2 : 1), nr = iters * 16.0;
    printf("%-46s wave0: %7.1f cyc/iter  (%5.1f cyc per MFMA)   wave4: %7.1f cyc/iter (%6.1f cyc per read, %2d B/lane)\n", name,
           (double)h[0] / iters, (double)h[0] / nm, (double)h[4] / iters, (double)h[4] / nr, RW);
}

int main() {
    float* out; unsigned long long* st;
    CK(hipMalloc(&out, 512 * 4)); CK(hipMalloc(&st, 64));
    run<2, 4, 0>("MFMA 32x32x2 alone (1 wave/SIMD)", out, st);
    run<2, 4, 1>("MFMA 16x16x4 alone (1 wave/SIMD)", out, st);
    run<3, 4, 0>("reads b32 alone", out, st);
    run<3, 8, 0>("reads b64 alone", out, st);
    run<3, 16, 0>("reads b128 alone", out, st);
    run<1, 4, 0>("32x32x2 stream | partner reads b32", out, st);
    run<1, 8, 0>("32x32x2 stream | partner reads b64", out, st);
    run<1, 16, 0>("32x32x2 stream | partner reads b128", out, st);
    run<1, 4, 1>("16x16x4 stream | partner reads b32", out, st);
    run<1, 8, 1>("16x16x4 stream | partner reads b64", out, st);
    run<1, 16, 1>("16x16x4 stream | partner reads b128", out, st);
    run<0, 4, 0>("same wave: 16 MFMA 32x32x2 + 16 reads b32", out, st);
    run<0, 8, 0>("same wave: 16 MFMA 32x32x2 + 16 reads b64", out, st);
    run<0, 16, 0>("same wave: 16 MFMA 32x32x2 + 16 reads b128", out, st);
    run<0, 4, 1>("same wave: 32 MFMA 16x16x4 + 16 reads b32", out, st);
    run<0, 16, 1>("same wave: 32 MFMA 16x16x4 + 16 reads b128", out, st);
    return 0;
}
