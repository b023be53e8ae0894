/* chunk_study.c -- CPU study of two-level accumulation for the trunk convolutions (round 5).
 * One fp32 fma chain per output over a CHUNK of the flattened (kh, kw, ci) reduction, chunks added in order into a second
 * accumulator.  Vectorised over the output channels (every lane is an independent chain: same bits as a scalar loop).
 * build: gcc -O3 -march=native -ffp-contract=off -fopenmp -shared -fPIC chunk_study.c -o chunk_study.so */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* x: (B,H,W,Cin); wT: (KH*KW*Cin, Cout) K-major; y: (B,Ho,Wo,Cout).  chunk: K elements per chunk (0 = one chain). */
void conv_chunked(const float* x, int B, int H, int W, int Cin, const float* wT, int Cout, int KH, int KW, int stride, int pad, int chunk,
                  const float* bias, const float* res, int relu, float* y) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int K = KH * KW * Cin;
    if (chunk <= 0) chunk = K;
#pragma omp parallel
    {
        float* acc = (float*)aligned_alloc(64, sizeof(float) * ((Cout + 15) / 16 * 16));
        float* tot = (float*)aligned_alloc(64, sizeof(float) * ((Cout + 15) / 16 * 16));
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (int ho = 0; ho < Ho; ++ho)
                for (int wo = 0; wo < Wo; ++wo) {
                    for (int c = 0; c < Cout; ++c) acc[c] = 0.0f, tot[c] = 0.0f;
                    int k = 0;
                    for (int kh = 0; kh < KH; ++kh)
                        for (int kw = 0; kw < KW; ++kw) {
                            const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
                            const int ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
                            const float* xp = x + (((int64_t)b * H + (ok ? hi : 0)) * W + (ok ? wi : 0)) * Cin;
                            for (int ci = 0; ci < Cin; ++ci, ++k) {
                                const float xv = ok ? xp[ci] : 0.0f;
                                const float* wr = wT + (int64_t)k * Cout;
                                if (ok)
                                    for (int c = 0; c < Cout; ++c) acc[c] = fmaf(xv, wr[c], acc[c]);
                                if ((k + 1) % chunk == 0 || k + 1 == K) {
                                    for (int c = 0; c < Cout; ++c) tot[c] += acc[c], acc[c] = 0.0f;
                                }
                            }
                        }
                    const int64_t m = ((int64_t)b * Ho + ho) * Wo + wo;
                    for (int c = 0; c < Cout; ++c) {
                        float v = tot[c] + bias[c];
                        if (res) v += res[m * Cout + c];
                        y[m * Cout + c] = relu ? fmaxf(v, 0.0f) : v;
                    }
                }
        free(acc);
        free(tot);
    }
}
