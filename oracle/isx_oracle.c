/*
 * isx_oracle.c -- CPU restatement of the reference's descriptor-extraction +
 * retrieval arithmetic (maxgreat/Instance-Search), in plain scalar C99.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, the
 * __graft_entry__.smoke() check and bench.py's cpu_baseline leg may load it.
 * The product path (instance-search_amd/) never links, imports or calls it.
 *
 * Parity status: PINNED.  Every function below is checked in
 * tests/test_oracle_golden.py against fixtures produced by running the
 * reference's own Python (imported in place from /root/reference with the
 * harness-side shims of oracle/gen_golden.py) -- see tests/golden/.
 * torch.mm / sort / topk / AvgPool2d live in PyTorch (un-pinned by the
 * reference); at that boundary the oracle is pinned against torch 2.10 CPU fp32
 * with the tolerances written in the tests.
 *
 * Conventions shared with the HIP library (include/isx.h):
 *   - all matrices row-major contiguous, fp32; indices int64; labels int32
 *   - canonical ranking order = (score descending, index ascending); -0.0 == +0.0
 *   - canonical dot product = k-ordered fp32 fmaf chain starting from +0.0f
 *     (this is what v_mfma_f32_32x32x2_f32 computes bit-for-bit)
 *   - canonical convolution sum (trunk convolutions, round 5): the flattened (kh, kw, ci)
 *     reduction in chunks of ISXO_CONV_CHUNK terms -- that chain inside a chunk, the chunk
 *     sums added in order into a second fp32 accumulator (conv_sum_t below)
 *
 * Citations are file:line under /root/reference.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ISXO_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------- keys ---- */

/* Monotone map float -> uint32 (larger float => larger key); -0.0 folded to +0.0
 * so that equal floats give equal keys.  NaNs are not produced on this path. */
static inline uint32_t f32_orderable(float f) {
    uint32_t u;
    if (f == 0.0f) f = 0.0f; /* -0 -> +0 */
    memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

/* 64-bit ranking key: larger key = ranked earlier.  idx < 2^32. */
static inline uint64_t rank_key(float score, uint64_t idx) {
    return ((uint64_t)f32_orderable(score) << 32) | (uint64_t)(0xFFFFFFFFu - (uint32_t)idx);
}

typedef struct { uint64_t key; float s; } kv_t;

static int cmp_key_desc(const void* a, const void* b) {
    uint64_t x = ((const kv_t*)a)->key, y = ((const kv_t*)b)->key;
    return (x < y) - (x > y);
}

/* ----------------------------------------------------- NormalizeL2 / Shift -- */

/* model/custom_modules.py:52-57  NormalizeL2Fun.forward:
 *   norm2 = input.pow(2).sum(1).add_(eps); norm = norm2.pow(0.5); out = input / norm
 * eps is added to the SQUARED norm, inside the sqrt.  The row sum is taken in
 * double and rounded once (torch's fp32 CPU summation order is unspecified; any
 * fp32 order lands within a few ulp of this). */
ISXO_API void isxo_l2norm_rows(const float* x, int64_t B, int64_t D, float eps, float* y) {
    for (int64_t b = 0; b < B; ++b) {
        const float* r = x + b * D;
        double ss = 0.0;
        for (int64_t j = 0; j < D; ++j) ss += (double)r[j] * (double)r[j];
        float n = sqrtf((float)ss + eps);
        for (int64_t j = 0; j < D; ++j) y[b * D + j] = r[j] / n;
    }
}

/* model/custom_modules.py:16-18  ShiftFun.forward: input + param (broadcast over rows) */
ISXO_API void isxo_shift_rows(const float* x, const float* param, int64_t B, int64_t F, float* y) {
    for (int64_t b = 0; b < B; ++b)
        for (int64_t j = 0; j < F; ++j) y[b * F + j] = x[b * F + j] + param[j];
}

/* ------------------------------------------------------------- pooling ---- */

/* model/siamese.py:49-54 (TuneClassif.forward with the classifier stripped,
 * train/classif_finetune.py:87-90) followed by NormalizeL2Fun
 * (train/classif_finetune.py:100): AvgPool2d over the whole HxW map (ResNet
 * avgpool 7), view(B,-1), L2.  Pool = sequential fp32 sum (h-major) / (H*W),
 * which is what torch's CPU avg_pool2d does. */
ISXO_API void isxo_gap_l2(const float* fmap, int64_t B, int C, int H, int W, float eps, float* y) {
    const int HW = H * W;
    float* pooled = (float*)malloc(sizeof(float) * (size_t)C);
    for (int64_t b = 0; b < B; ++b) {
        for (int c = 0; c < C; ++c) {
            const float* p = fmap + ((size_t)b * C + c) * HW;
            float s = 0.0f;
            for (int i = 0; i < HW; ++i) s += p[i];
            pooled[c] = s / (float)HW;
        }
        isxo_l2norm_rows(pooled, 1, C, eps, y + (size_t)b * C);
    }
    free(pooled);
}

/* model/siamese.py:67-71  nn.AvgPool2d(feature_size2d, stride=1) on an NCHW map. */
ISXO_API void isxo_boxpool_s1(const float* fmap, int64_t B, int C, int H, int W, int kh, int kw, float* out) {
    const int Ho = H - kh + 1, Wo = W - kw + 1;
    for (int64_t bc = 0; bc < B * (int64_t)C; ++bc) {
        const float* p = fmap + (size_t)bc * H * W;
        float* o = out + (size_t)bc * Ho * Wo;
        for (int i = 0; i < Ho; ++i)
            for (int j = 0; j < Wo; ++j) {
                float s = 0.0f;
                for (int a = 0; a < kh; ++a)
                    for (int b = 0; b < kw; ++b) s += p[(i + a) * W + (j + b)];
                o[i * Wo + j] = s / (float)(kh * kw);
            }
    }
}

/* ------------------------------------------------------- region path ------ */

/* train/classif_regions.py:118-128:
 *   max_pred = out.max(1)            (max over classes)      -> (1,1,H',W')
 *   max_pred1, max_i1 = max_pred.max(2)  (over rows, per column)
 *   _, max_i2 = max_pred1.max(3)     (over columns)
 *   i2 = max_i2[0]; i1 = max_i1[i2];  desc = NormalizeL2(out[:, :, i1, i2])
 * Tie-break (first maximal index, as torch CPU max returns): smallest column,
 * then smallest row inside that column.  loc = {i1 (row), i2 (col)}. */
ISXO_API void isxo_best_location_desc(const float* cls, int K, int Hp, int Wp, float eps, float* desc, int64_t* loc) {
    const int P = Hp * Wp;
    float* mx = (float*)malloc(sizeof(float) * (size_t)P);
    for (int p = 0; p < P; ++p) {
        float m = cls[p];
        for (int c = 1; c < K; ++c) { float v = cls[(size_t)c * P + p]; if (v > m) m = v; }
        mx[p] = m;
    }
    int bi2 = 0, bi1 = 0; float best = 0.0f;
    for (int j = 0; j < Wp; ++j) {
        int i1 = 0; float cm = mx[j];
        for (int i = 1; i < Hp; ++i) if (mx[i * Wp + j] > cm) { cm = mx[i * Wp + j]; i1 = i; }
        if (j == 0 || cm > best) { best = cm; bi2 = j; bi1 = i1; }
    }
    float* v = (float*)malloc(sizeof(float) * (size_t)K);
    for (int c = 0; c < K; ++c) v[c] = cls[(size_t)c * P + bi1 * Wp + bi2];
    isxo_l2norm_rows(v, 1, K, eps, desc);
    loc[0] = bi1; loc[1] = bi2;
    free(v); free(mx);
}

/* model/siamese.py:191-194:  c_maxv = c.max(1).view(-1); k = min(len, self.k); topk(k)
 * flat index is row-major over (H',W').  Returns the number of regions written.
 * Canonical order: (score desc, flat index asc). */
ISXO_API int isxo_region_topk(const float* cls, int K, int Hp, int Wp, int k, int64_t* flat_idx, float* score) {
    const int P = Hp * Wp;
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)P);
    float* mx = (float*)malloc(sizeof(float) * (size_t)P);
    for (int p = 0; p < P; ++p) {
        float m = cls[p];
        for (int c = 1; c < K; ++c) { float v = cls[(size_t)c * P + p]; if (v > m) m = v; }
        mx[p] = m; kv[p].key = rank_key(m, (uint64_t)p);
    }
    qsort(kv, (size_t)P, sizeof(kv_t), cmp_key_desc);
    int kk = k < P ? k : P;
    for (int i = 0; i < kk; ++i) {
        int64_t p = (int64_t)(0xFFFFFFFFu - (uint32_t)(kv[i].key & 0xFFFFFFFFu));
        flat_idx[i] = p; score[i] = mx[p];
    }
    free(kv); free(mx);
    return kk;
}

/* model/siamese.py:199-219: window (row,col) = (idx // W', idx % W');
 *   region = x[:, :, row:row+kh, col:col+kw].contiguous().view(1,-1)   (C,h,w order)
 *   feature_reduc1[0:2] = NormalizeL2 -> Shift        (the Linear follows in the caller)
 * rows is (k, C*kh*kw); shift may be NULL (= zeros, its initial value, custom_modules.py:35-36). */
ISXO_API void isxo_region_gather_l2(const float* fmap, int C, int Hf, int Wf, int kh, int kw,
                                    const int64_t* flat_idx, int k, int Wp,
                                    const float* shift, float eps, float* rows) {
    const size_t F = (size_t)C * kh * kw;
    float* tmp = (float*)malloc(sizeof(float) * F);
    for (int r = 0; r < k; ++r) {
        int row = (int)(flat_idx[r] / Wp), col = (int)(flat_idx[r] % Wp);
        size_t o = 0;
        for (int c = 0; c < C; ++c)
            for (int a = 0; a < kh; ++a)
                for (int b = 0; b < kw; ++b)
                    tmp[o++] = fmap[((size_t)c * Hf + (row + a)) * Wf + (col + b)];
        isxo_l2norm_rows(tmp, 1, (int64_t)F, eps, rows + (size_t)r * F);
        if (shift) for (size_t j = 0; j < F; ++j) rows[(size_t)r * F + j] += shift[j];
    }
    free(tmp);
}

/* ------------------------------------------------------------ retrieval --- */

/* test/classif_finetune_test.py:82  sim = torch.mm(test_emb, ref_emb.t()).
 * Canonical summation: k-ordered fp32 fmaf chain (see header). */
/* target_clones: use the hardware FMA where the host has it (the libm fmaf
 * software path gives the same bits, only slower). */
__attribute__((target_clones("fma", "default")))
static float dot_fma(const float* q, const float* g, int D) {
    float acc = 0.0f;
    for (int k = 0; k < D; ++k) acc = fmaf(q[k], g[k], acc);
    return acc;
}

ISXO_API void isxo_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim) {
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) sim[i * N + j] = dot_fma(Q + i * D, G + j * D, D);
}

/* transforms.ToTensor() + transforms.Normalize(m, s) (test/classif_finetune_test.py:62-73): (B,H,W,3) uint8 -> (B,3,H,W) fp32. */
ISXO_API void isxo_images_u8_to_f32(const uint8_t* img, int64_t B, int H, int W, const float* mean, const float* sd, float* out) {
    const int64_t HW = (int64_t)H * W;
    for (int64_t b = 0; b < B; ++b)
        for (int64_t p = 0; p < HW; ++p)
            for (int c = 0; c < 3; ++c) {
                const float x = (float)img[(b * HW + p) * 3 + c] / 255.0f;
                out[(b * 3 + c) * HW + p] = (x - mean[c]) / sd[c];
            }
}

/* Two-level sum of the trunk convolutions (include/isx.h ISX_CONV_CHUNK; csrc/gemm_tile.hpp fold_chunk): terms arrive in the flattened
 * (kh, kw, ci) order; `acc` is the fma chain of the current chunk (from +0), `tot` the in-order sum of the finished chunks.  One chain over
 * the whole reduction (rounds 1-4) sat 1.2x further from a float64 evaluation of the ResNet-50 / ResNet-152 descriptors than torch's CPU fp32
 * path (the reference's arithmetic; third-party, unpinned by the reference); chunks of 64 sit at 0.7-0.8x.  K <= 64: tot = 0 + chain. */
#define ISXO_CONV_CHUNK 64
typedef struct { float acc, tot; int n, chunk; } conv_sum_t;
static inline void cs_init(conv_sum_t* s, int chunk) { s->acc = 0.0f; s->tot = 0.0f; s->n = 0; s->chunk = chunk; }
static inline void cs_fold(conv_sum_t* s) { s->tot = s->tot + s->acc; s->acc = 0.0f; s->n = 0; }
static inline void cs_term(conv_sum_t* s, float a, float b) {
    s->acc = fmaf(a, b, s->acc);
    if (++s->n == s->chunk) cs_fold(s);
}
static inline float cs_value(conv_sum_t* s) {
    if (s->n) cs_fold(s);                       /* the last, partial chunk */
    return s->tot;
}

/* relu(y + bias) -> MaxPool2d(3, stride 2, padding 1) on an NHWC map (torchvision ResNet stem after the folded BN). */
ISXO_API void isxo_bias_relu_maxpool_nhwc(const float* y, const float* bias, int64_t B, int H, int W, int C, float* out) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    for (int64_t b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo)
                for (int c = 0; c < C; ++c) {
                    float m = -INFINITY;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const int hi = ho * 2 - 1 + kh, wi = wo * 2 - 1 + kw;
                            if (hi < 0 || hi >= H || wi < 0 || wi >= W) continue;
                            const float v = fmaxf(y[((b * H + hi) * W + wi) * (int64_t)C + c] + bias[c], 0.0f);   /* relu first, as the reference does */
                            m = fmaxf(m, v);
                        }
                    out[((b * Ho + ho) * Wo + wo) * (int64_t)C + c] = m;
                }
}

/* 1x1 stride-1 convolution over NHWC pixels with the folded-BN epilogue of the inference trunk
 * (torchvision Bottleneck conv1 / conv3 / downsample as used through model/nn_utils.py:56-71 extract_layers):
 * y[m][co] = act(sum_ci x[m][ci] * w[co][ci] (two-level sum, ci ascending) + bias[co] (+ res[m][co])). */
ISXO_API void isxo_conv1x1_nhwc(const float* x, int64_t M, int Cin, const float* w, int Cout, const float* bias,
                                const float* res, int relu, float* y) {
    for (int64_t m = 0; m < M; ++m)
        for (int co = 0; co < Cout; ++co) {
            conv_sum_t cs;
            cs_init(&cs, ISXO_CONV_CHUNK);
            for (int ci = 0; ci < Cin; ++ci) cs_term(&cs, x[m * Cin + ci], w[(int64_t)co * Cin + ci]);
            float v = cs_value(&cs) + bias[co];
            if (res) v += res[m * Cout + co];
            y[m * Cout + co] = relu ? fmaxf(v, 0.0f) : v;
        }
}

/* conv3 + 1x1 projection shortcut of a bottleneck block as one two-level sum per output: t's K1 channels, then the K2 channels
 * of the strided block input (torchvision Bottleneck: out = conv3(t) + downsample(x), BN folded, then ReLU). */
ISXO_API void isxo_conv1x1_dual_nhwc(const float* t, int K1, const float* x, int64_t B, int H, int W, int K2, int stride,
                                     const float* w, int Cout, const float* bias, int relu, float* y) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (int64_t b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                const int64_t m = (b * Ho + ho) * Wo + wo;
                const float* tp = t + m * K1;
                const float* xp = x + ((b * H + (int64_t)ho * stride) * W + (int64_t)wo * stride) * K2;
                for (int co = 0; co < Cout; ++co) {
                    const float* wp = w + (int64_t)co * (K1 + K2);
                    conv_sum_t cs;
                    cs_init(&cs, ISXO_CONV_CHUNK);
                    for (int c = 0; c < K1; ++c) cs_term(&cs, tp[c], wp[c]);
                    for (int c = 0; c < K2; ++c) cs_term(&cs, xp[c], wp[K1 + c]);
                    const float v = cs_value(&cs) + bias[co];
                    y[m * Cout + co] = relu ? fmaxf(v, 0.0f) : v;
                }
            }
}

/* 3x3 convolution, padding 1, stride 1|2, NHWC, weights (Cout,3,3,Cin), folded-BN epilogue (conv2 of the torchvision
 * Bottleneck / BasicBlock convolutions inside `features`): two-level sum over (kh, kw, ci) in that order; padding taps
 * contribute fma(0, w, acc) and count as terms. */
ISXO_API void isxo_conv3x3_nhwc(const float* x, int64_t B, int H, int W, int Cin, const float* w, int Cout, int stride,
                                const float* bias, const float* res, int relu, float* y) {
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    for (int64_t b = 0; b < B; ++b)
        for (int ho = 0; ho < Ho; ++ho)
            for (int wo = 0; wo < Wo; ++wo) {
                const int64_t m = (b * Ho + ho) * Wo + wo;
                for (int co = 0; co < Cout; ++co) {
                    conv_sum_t cs;
                    cs_init(&cs, ISXO_CONV_CHUNK);
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const int hi = ho * stride - 1 + kh, wi = wo * stride - 1 + kw;
                            const int ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
                            const float* xp = x + ((b * H + (ok ? hi : 0)) * W + (ok ? wi : 0)) * (int64_t)Cin;
                            const float* wp = w + (((int64_t)co * 3 + kh) * 3 + kw) * Cin;
                            for (int ci = 0; ci < Cin; ++ci) cs_term(&cs, ok ? xp[ci] : 0.0f, wp[ci]);
                        }
                    float v = cs_value(&cs) + bias[co];
                    if (res) v += res[m * Cout + co];
                    y[m * Cout + co] = relu ? fmaxf(v, 0.0f) : v;
                }
            }
}

/* The torchvision ResNet stem on a channels-last image batch (conv1 7x7 / stride 2 / padding 3 with bn1 folded, relu, maxpool 3/2/1;
 * the first four modules of the `features` trunk of model/ModelDefinition.py as split by model/nn_utils.py:56-71):
 * conv = two-level sum over (kh, kw, c) ascending in chunks of three filter rows, padding taps contribute fma(0, w, acc); y = relu(conv + bias); out = max over the
 * in-bounds 3x3 window.  x: (B,H,W,3), w: (64,7,7,3), out: (B,Hp,Wp,64). */
ISXO_API void isxo_stem7x7_pool_nhwc(const float* x, int64_t B, int H, int W, const float* w, const float* bias, float* out) {
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    float* y = (float*)malloc((size_t)Hc * Wc * 64 * sizeof(float));
    for (int64_t b = 0; b < B; ++b) {
        for (int ho = 0; ho < Hc; ++ho)
            for (int wo = 0; wo < Wc; ++wo)
                for (int co = 0; co < 64; ++co) {
                    conv_sum_t cs;
                    cs_init(&cs, 63);                      /* three filter rows of 21 terms per chunk: 63 + 63 + 21 */
                    for (int kh = 0; kh < 7; ++kh)
                        for (int kw = 0; kw < 7; ++kw) {
                            const int hi = ho * 2 - 3 + kh, wi = wo * 2 - 3 + kw;
                            const int ok = hi >= 0 && hi < H && wi >= 0 && wi < W;
                            const float* xp = x + ((b * H + (ok ? hi : 0)) * W + (ok ? wi : 0)) * 3;
                            const float* wp = w + (((int64_t)co * 7 + kh) * 7 + kw) * 3;
                            for (int c = 0; c < 3; ++c) cs_term(&cs, ok ? xp[c] : 0.0f, wp[c]);
                        }
                    y[((int64_t)ho * Wc + wo) * 64 + co] = fmaxf(cs_value(&cs) + bias[co], 0.0f);
                }
        for (int po = 0; po < Hp; ++po)
            for (int qo = 0; qo < Wp; ++qo)
                for (int co = 0; co < 64; ++co) {
                    float m = -INFINITY;
                    for (int kh = 0; kh < 3; ++kh)
                        for (int kw = 0; kw < 3; ++kw) {
                            const int hi = po * 2 - 1 + kh, wi = qo * 2 - 1 + kw;
                            if (hi < 0 || hi >= Hc || wi < 0 || wi >= Wc) continue;
                            m = fmaxf(m, y[((int64_t)hi * Wc + wi) * 64 + co]);
                        }
                    out[(((b * Hp) + po) * Wp + qo) * 64 + co] = m;
                }
    }
    free(y);
}

/* Fused form of  torch.mm -> (sort | topk): per query the k best gallery rows in
 * canonical order, global index = idx_base + local row.  Entries beyond N are
 * (-inf, -1). */
ISXO_API void isxo_cosine_topk(const float* Q, int64_t M, const float* G, int64_t N, int D, int k,
                               int64_t idx_base, float* top_score, int64_t* top_idx) {
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(N > 0 ? N : 1));
    float* s = (float*)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < M; ++i) {
        for (int64_t j = 0; j < N; ++j) { s[j] = dot_fma(Q + i * D, G + j * D, D); kv[j].key = rank_key(s[j], (uint64_t)j); }
        qsort(kv, (size_t)N, sizeof(kv_t), cmp_key_desc);
        for (int t = 0; t < k; ++t) {
            if (t < N) {
                int64_t j = (int64_t)(0xFFFFFFFFu - (uint32_t)(kv[t].key & 0xFFFFFFFFu));
                top_score[i * k + t] = s[j]; top_idx[i * k + t] = idx_base + j;
            } else { top_score[i * k + t] = -INFINITY; top_idx[i * k + t] = -1; }
        }
    }
    free(kv); free(s);
}

/* utils/metrics.py:33  `_, ranked_list = sim[i].sort(dim=0, descending=True)` made
 * deterministic: full ranking of every row in canonical order. */
ISXO_API void isxo_rank_full(const float* sim, int64_t M, int64_t N, int64_t* ranked) {
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < M; ++i) {
        for (int64_t j = 0; j < N; ++j) kv[j].key = rank_key(sim[i * N + j], (uint64_t)j);
        qsort(kv, (size_t)N, sizeof(kv_t), cmp_key_desc);
        for (int64_t j = 0; j < N; ++j) ranked[i * N + j] = (int64_t)(0xFFFFFFFFu - (uint32_t)(kv[j].key & 0xFFFFFFFFu));
    }
    free(kv);
}

/* Same ranking restricted to the first k entries (what torch.topk / max / kthvalue
 * consumers need): utils/metrics.py:10-13. */
ISXO_API void isxo_topk_rows(const float* sim, int64_t M, int64_t N, int k, int64_t idx_base,
                             float* top_score, int64_t* top_idx) {
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (size_t)(N > 0 ? N : 1));
    for (int64_t i = 0; i < M; ++i) {
        for (int64_t j = 0; j < N; ++j) kv[j].key = rank_key(sim[i * N + j], (uint64_t)j);
        qsort(kv, (size_t)N, sizeof(kv_t), cmp_key_desc);
        for (int t = 0; t < k; ++t) {
            if (t < N) {
                int64_t j = (int64_t)(0xFFFFFFFFu - (uint32_t)(kv[t].key & 0xFFFFFFFFu));
                top_score[i * k + t] = sim[i * N + j]; top_idx[i * k + t] = idx_base + j;
            } else { top_score[i * k + t] = -INFINITY; top_idx[i * k + t] = -1; }
        }
    }
    free(kv);
}

/* utils/metrics.py:25-45  avg_precision (Oxford-buildings trapezoid AP):
 *   n_pos = #gallery with the query's label - (kth-1); None (here NaN) if <= 0
 *   walk EVERY rank n; the first kth-1 ranks are skipped entirely;
 *   recall = hits/float(n_pos); precision = hits/(j+1.0)
 *   ap += (recall-old_recall) * ((old_precision+precision)/2.0), old_precision0 = 1
 * All in float64, same operation order as the Python. */
ISXO_API void isxo_average_precision(const int64_t* ranked, int64_t M, int64_t N,
                                     const int32_t* qlab, const int32_t* glab, int kth, double* ap_out) {
    for (int64_t i = 0; i < M; ++i) {
        int64_t n_pos = 0;
        for (int64_t j = 0; j < N; ++j) n_pos += (glab[j] == qlab[i]);
        n_pos -= (kth - 1);
        if (n_pos <= 0) { ap_out[i] = NAN; continue; }
        double old_recall = 0.0, old_precision = 1.0, ap = 0.0;
        int64_t hits = 0, j = 0;
        for (int64_t n = 0; n < N; ++n) {
            if (n + 1 < kth) continue;
            if (glab[ranked[i * N + n]] == qlab[i]) hits += 1;
            double recall = (double)hits / (double)n_pos;
            double precision = (double)hits / ((double)j + 1.0);
            ap += (recall - old_recall) * ((old_precision + precision) / 2.0);
            old_recall = recall; old_precision = precision;
            j += 1;
        }
        ap_out[i] = ap;
    }
}

/* Multi-GPU merge (no reference counterpart: the reference is single-device,
 * test/classif_finetune_test.py:82).  P per-shard lists (P,M,k), each already in
 * canonical order with GLOBAL indices -> the k best of the union, canonical
 * order; (-inf,-1) padding entries sort last. */
ISXO_API void isxo_topk_merge(const float* scores, const int64_t* idx, int P, int64_t M, int k,
                              float* out_s, int64_t* out_i) {
    const size_t T = (size_t)P * k;
    kv_t* kv = (kv_t*)malloc(sizeof(kv_t) * (T ? T : 1));
    for (int64_t m = 0; m < M; ++m) {
        size_t n = 0;
        for (int p = 0; p < P; ++p)
            for (int t = 0; t < k; ++t) {
                size_t o = ((size_t)p * M + m) * k + t;
                if (idx[o] < 0) continue;
                kv[n].key = rank_key(scores[o], (uint64_t)idx[o]); kv[n].s = scores[o]; ++n;
            }
        qsort(kv, n, sizeof(kv_t), cmp_key_desc);
        for (int t = 0; t < k; ++t) {
            if ((size_t)t < n) {
                out_s[m * k + t] = kv[t].s;
                out_i[m * k + t] = (int64_t)(0xFFFFFFFFu - (uint32_t)(kv[t].key & 0xFFFFFFFFu));
            } else { out_s[m * k + t] = -INFINITY; out_i[m * k + t] = -1; }
        }
    }
    free(kv);
}

/* utils/train_siamese.py:74-76  sum_pos / sum_neg statistics of test_descriptor_net:
 * sum of sim over label-equal pairs and over the rest (float64 accumulation;
 * the reference adds 0-dim fp32 tensors one by one -- tolerance in the test). */
ISXO_API void isxo_masked_sums(const float* sim, int64_t M, int64_t N, const int32_t* qlab,
                               const int32_t* glab, double* sum_pos, double* sum_all) {
    double sp = 0.0, sa = 0.0;
    for (int64_t i = 0; i < M; ++i)
        for (int64_t j = 0; j < N; ++j) {
            double v = (double)sim[i * N + j];
            sa += v; if (qlab[i] == glab[j]) sp += v;
        }
    *sum_pos = sp; *sum_all = sa;
}

/* ------------------------------------------------------- training step ----- */

/* train/siamese_descriptor.py:94-128: negative for the positive couple (i1,i2):
 *   ind_exl = same label as the anchor  [| similarities[i1] >= sim_pos  while epoch < train_epoch_switch]
 *   all excluded -> None (-1 here; the caller draws a random negative)
 *   else sims[ind_exl] = -2; k = argmax(sims)       (first maximal index) */
ISXO_API void isxo_mine_negatives(const float* sim, int64_t N, const int32_t* lab, const int64_t* i1, const int64_t* i2,
                                  int64_t n_couples, int semi_hard, int64_t* neg) {
    for (int64_t c = 0; c < n_couples; ++c) {
        const float* row = sim + i1[c] * N;
        const float sim_pos = row[i2[c]];
        int64_t best = -1; float bs = 0.0f;
        for (int64_t j = 0; j < N; ++j) {
            int excl = (lab[j] == lab[i1[c]]) || (semi_hard && row[j] >= sim_pos);
            if (excl) continue;
            if (best < 0 || row[j] > bs) { best = j; bs = row[j]; }
        }
        neg[c] = best;
    }
}

/* model/custom_modules.py:153-203 TripletLossFun forward (per-row clamped losses + their sum, divided by B
 * when size_average) and backward (grad_output = 1). */
ISXO_API float isxo_triplet_loss(const float* a, const float* p, const float* n, int64_t B, int D, float margin,
                                 int normalized, int size_average, float* rows, float* ga, float* gp, float* gn) {
    double total = 0.0;
    for (int64_t b = 0; b < B; ++b) {
        double s = 0.0;
        for (int j = 0; j < D; ++j) {
            const double av = a[b * D + j], pv = p[b * D + j], nv = n[b * D + j];
            s += normalized ? (av * nv - av * pv) : ((av - pv) * (av - pv) - (av - nv) * (av - nv));
        }
        float l = normalized ? (float)s + margin : ((float)s + 2.0f * margin) * 0.5f;
        if (l <= 0.0f) l = 0.0f;
        rows[b] = l; total += l;
        const float sc = size_average ? 1.0f / (float)B : 1.0f;
        for (int j = 0; j < D; ++j) {
            const float av = a[b * D + j], pv = p[b * D + j], nv = n[b * D + j];
            const int on = l > 0.0f;
            ga[b * D + j] = on ? (nv - pv) * sc : 0.0f;
            gp[b * D + j] = on ? (normalized ? -av : pv - av) * sc : 0.0f;
            gn[b * D + j] = on ? (normalized ? av : av - nv) * sc : 0.0f;
        }
    }
    return (float)(size_average ? total / (double)B : total);
}


/* ----------------------------------------------------------------- DBA ---- */
/* test/instance_avg.py:7-33: every gallery descriptor is replaced by itself plus the rank-weighted sum of its nearest
 * neighbours WITHIN ITS INSTANCE (same label), renormalised with x / (|x| + 1e-10) -- eps outside the norm here.
 *   sim = torch.mm(E, E.t())                                  (:11)  canonical fma chain here
 *   per item i: num_neighbors = #same-label - 1, capped by k when 0 <= k < num_neighbors (:18-20); <= 0 -> kept (:21-23)
 *   sim[i,i] = -2, other labels = -2, sort descending (:24-26)  -> canonical order (score desc, index asc) over the same-label items
 *   agg = E[i]; for j < num_neighbors: agg += E[best[j]] * ((num_neighbors - j) / float(num_neighbors + 1))   (:27-30)
 *   new[i] = agg / (agg.norm() + 1e-10)                        (:31)
 * emb: (N,D); labels: (N) int32; out: (N,D).  O(N * group) memory: only same-label pairs are ever scored. */
static int cmp_u64_desc(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return (x < y) - (x > y);
}

ISXO_API void isxo_dba(const float* emb, int64_t N, int64_t D, const int32_t* labels, int k, float* out) {
    uint64_t* keys = (uint64_t*)malloc((size_t)(N > 0 ? N : 1) * sizeof(uint64_t));
    float* agg = (float*)malloc((size_t)(D > 0 ? D : 1) * sizeof(float));
    for (int64_t i = 0; i < N; ++i) {
        const float* ei = emb + i * D;
        int64_t n = 0;
        for (int64_t m = 0; m < N; ++m) {
            if (m == i || labels[m] != labels[i]) continue;
            const float* em = emb + m * D;
            float acc = 0.0f;
            for (int64_t d = 0; d < D; ++d) acc = fmaf(ei[d], em[d], acc);
            keys[n++] = rank_key(acc, (uint64_t)m);
        }
        int64_t nn = n;
        if (k >= 0 && k < nn) nn = k;
        if (nn <= 0) { memcpy(out + i * D, ei, (size_t)D * sizeof(float)); continue; }
        qsort(keys, (size_t)n, sizeof(uint64_t), cmp_u64_desc);
        memcpy(agg, ei, (size_t)D * sizeof(float));
        for (int64_t j = 0; j < nn; ++j) {
            const float w = (float)((double)(nn - j) / (double)(nn + 1));
            const float* eb = emb + (int64_t)(0xFFFFFFFFu - (uint32_t)(keys[j] & 0xFFFFFFFFull)) * D;
            for (int64_t d = 0; d < D; ++d) agg[d] = agg[d] + eb[d] * w;
        }
        double ss = 0.0;
        for (int64_t d = 0; d < D; ++d) ss += (double)agg[d] * (double)agg[d];
        const float nrm = (float)sqrt(ss) + 1e-10f;
        for (int64_t d = 0; d < D; ++d) out[i * D + d] = agg[d] / nrm;
    }
    free(keys);
    free(agg);
}
