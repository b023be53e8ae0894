/*
 * isx.h -- C ABI of libisx.so: the MI355X (gfx950) descriptor-pooling and
 * nearest-neighbour retrieval kernels behind the Python surface of
 * maxgreat/Instance-Search (model/custom_modules, model/siamese,
 * train/<approach>.py::get_embeddings, utils/metrics, test/<approach>_test.py).
 *
 * The reference has no FFI of its own: every entry below replaces a run of stock
 * torch calls in the reference's Python; the file:line it replaces is cited per
 * entry (paths relative to the reference repository root).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller, row-major contiguous;
 *     fp32 values, int64 indices, int32 labels, fp64 AP values
 *   - every call only ENQUEUES work on `stream` (a hipStream_t passed as void*) and
 *     returns; no allocation, no synchronisation, no global state in the entry points
 *     declared here: safe to capture in a hipGraph and re-entrant from several host
 *     threads on distinct streams.  (The library also exports four UNDECLARED
 *     process-global A/B hooks -- isx_debug_set_gemm_cfg, isx_debug_set_conv_cfg,
 *     isx_debug_set_f16_tile, isx_debug_fast_fallback_rows -- for tests and
 *     scratch/ timing scripts: they force a tile shape for every later call of
 *     the process (relaxed atomics: flipping one while another host thread
 *     launches is a data-race-free way to get either tile shape), never change a
 *     result, and are not part of this ABI.)
 *   - environment, read once per process, tuning only (no value changes a result):
 *     ISX_TAIL_SPLIT=0 (128x128 grids without the 64x64 tail), ISX_TOPK_CHUNK_MB
 *     (score-chunk budget of the running top-k, default 1024), ISX_TOPK_FIRST
 *     (bootstrap chunk, default 8192 columns), ISX_FAST_KL_PCT (candidates kept per
 *     query by the fp16 filter, in % of k, default 200)
 *   - return 0 = ISX_OK, <0 = error; isx_last_error() gives a thread-local message
 *   - canonical ranking order everywhere: (score DESCENDING, index ASCENDING),
 *     -0.0 == +0.0; gallery indices must be < 2^32
 *   - canonical dot product: k-ordered fp32 fma chain from +0.0f, which is what
 *     v_mfma_f32_32x32x2_f32 computes bit for bit
 *   - canonical convolution sum (trunk kernels, version >= 110): the flattened reduction
 *     (kh, kw, ci) in chunks of ISX_CONV_CHUNK terms -- that chain inside a chunk, the chunk
 *     sums added in order into a second fp32 accumulator (tot = tot + chain_c), then the
 *     epilogue on tot.  The 7x7 stem cuts after filter rows 2 and 5 (63 + 63 + 21 terms).
 */
#ifndef ISX_H_
#define ISX_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISX_CONV_CHUNK 64      /* terms per first-level chain of a trunk convolution (see above) */

#define ISX_OK 0
#define ISX_ERR_ARG (-1)       /* bad shape / pointer / unsupported size */
#define ISX_ERR_WORKSPACE (-2) /* workspace too small */
#define ISX_ERR_HIP (-3)       /* a HIP launch failed */

typedef void* isx_stream_t; /* hipStream_t */

const char* isx_last_error(void);
int isx_version(void);

/* ---- descriptor head ------------------------------------------------------- */

/* model/custom_modules.py:52-57 NormalizeL2Fun.forward: y = x / sqrt(sum_j x_j^2 + eps),
 * eps INSIDE the sqrt.  x, y: (B, D).  In-place (y == x) allowed. */
int isx_l2norm_rows(const float* x, int64_t B, int64_t D, float eps, float* y, isx_stream_t stream);

/* model/siamese.py:110-113 feature_reduc1[0:2] = NormalizeL2 -> Shift
 * (custom_modules.py:52-57 then :16-18): y = x / sqrt(sum x^2 + eps) + shift.
 * shift: (F) or NULL.  x, y: (B, F). */
int isx_l2norm_shift_rows(const float* x, const float* shift, int64_t B, int64_t F, float eps, float* y,
                          isx_stream_t stream);

/* model/siamese.py:49-54 TuneClassif.forward with the classifier stripped
 * (train/classif_finetune.py:87-90) + NormalizeL2Fun (:100): global average pool over
 * the whole HxW map, flatten, L2.  fmap: (B,C,H,W) NCHW; y: (B,C). */
int isx_gap_l2(const float* fmap, int64_t B, int C, int H, int W, float eps, float* y, isx_stream_t stream);

/* Same operation on a channels-last feature map: fmap is (B,H,W,C) in memory (torch
 * memory_format=channels_last of a logical (B,C,H,W) tensor), which is what the NHWC convolution
 * kernels of the backbone produce.  Same summation order, same result.  y: (B,C). */
int isx_gap_l2_nhwc(const float* fmap, int64_t B, int C, int H, int W, float eps, float* y, isx_stream_t stream);

/* Fused epilogue of the inference trunk's convolutions (BatchNorm folded, model/nn_utils.py fold_batch_norm):
 * y[i] = act(y[i] + bias[c(i)] + (residual ? residual[i] : 0)), in place, c(i) = (i / inner) % C
 * (inner = 1 for channels-last memory, H*W for NCHW), act = ReLU when relu != 0.  Replaces the separate
 * bias-add / residual-add / ReLU kernels that follow every convolution (reference blocks:
 * torchvision Bottleneck via model/nn_utils.py:56-71). */
int isx_bias_act_inplace(float* y, const float* bias, const float* residual, int64_t n, int C, int64_t inner, int relu,
                         isx_stream_t stream);

/* Image ingest (next-scope row f4): transforms.ToTensor() + transforms.Normalize(m, s) of the test mains
 * (test/classif_finetune_test.py:62-73, classif_regions_test.py:55-65, siamese_*_test.py) from decoded RGB bytes:
 * out[b][c][h][w] = (img[b][h][w][c] / 255 - mean[c]) / std[c], fp32, same operation order as torch.
 * img: (B,H,W,3) uint8 RGB; out: (B,3,H,W) fp32 in NCHW memory, or channels-last memory ((B,H,W,3)) when channels_last != 0. */
int isx_images_u8_to_f32(const uint8_t* img, int64_t B, int H, int W, float mean0, float mean1, float mean2, float std0, float std1,
                         float std2, int channels_last, float* out, isx_stream_t stream);

/* Stem of the same trunk: relu(conv7x7 + bias) -> MaxPool2d(3, stride 2, padding 1) (torchvision ResNet stem as split
 * by model/nn_utils.py:56-71), epilogue and pooling in ONE pass over the channels-last convolution output:
 * out[b][ho][wo][c] = relu(max_{3x3 window, in bounds} y[b][2ho-1+kh][2wo-1+kw][c] + bias[c]).
 * y: (B,H,W,C), out: (B,Ho,Wo,C), Ho = (H-1)/2 + 1; C % 4 == 0. */
int isx_bias_relu_maxpool_nhwc(const float* y, const float* bias, int64_t B, int H, int W, int C, float* out, isx_stream_t stream);

/* conv2 + conv3 of a torchvision Bottleneck with 64 mid channels (the first ResNet stage; BN folded; model/nn_utils.py:56-71) as ONE kernel:
 *   y = act( W3 . relu(conv3x3(x, W2) + b2) + b3 + (residual ? residual : 0) ),
 * the 64-channel mid activation stays on chip (registers -> LDS -> second MFMA loop).  Same fma chains as isx_conv3x3_nhwc followed by
 * isx_conv1x1_nhwc (bit-identical results).  x: (B,H,W,Cin), Cin % 32 == 0; w2_ohwi: (64,3,3,Cin); b2: (64); w3t: (64,256) = the 1x1
 * weight (256,64) TRANSPOSED; b3: (256); y / residual: (B,Ho,Wo,256); Cout must be 256. */
int isx_conv3x3_expand_nhwc(const float* x, int64_t B, int H, int W, int Cin, const float* w2_ohwi, const float* b2, int stride,
                            const float* w3t, int Cout, const float* b3, const float* residual, int relu, float* y, isx_stream_t stream);

/* The same for the FIRST block of that stage, whose shortcut is a 1x1 projection of the 64-channel block input x2 (stride 1):
 *   y = act( [W3 | Wd] . [relu(conv3x3(t, W2) + b2) ; x2] + bias )  -- one fma chain per output over the 64 mid channels, then the 64 channels
 * of x2, as isx_conv3x3_nhwc followed by isx_conv1x1_dual_nhwc.  t: (B,H,W,Cin); x2: (B,H,W,64); wcat_t: (128,256) = [W3 | Wd] TRANSPOSED;
 * bias: (256) = b3 + bd; y: (B,H,W,256). */
int isx_conv3x3_expand_dual_nhwc(const float* t, int64_t B, int H, int W, int Cin, const float* w2_ohwi, const float* b2, const float* x2,
                                 const float* wcat_t, int Cout, const float* bias, int relu, float* y, isx_stream_t stream);

/* The whole stem as ONE kernel: conv 7x7 / stride 2 / padding 3 (3 -> 64 channels, bn1 folded into w and bias) + ReLU +
 * MaxPool2d(3, stride 2, padding 1) on a channels-last image batch; the convolution output never reaches memory.  Replaces
 * conv1, bn1, relu, maxpool of the torchvision ResNet `features` trunk (model/ModelDefinition.py, split by model/nn_utils.py:56-71,
 * run from model/siamese.py:20,107,151).  conv = fp32 fma chain over (kh, kw, c) ascending (bit-exact vs the oracle).
 * x: (B,H,W,3) fp32, 16-B aligned, W % 4 == 0, W <= 896 (images wider than 224 are walked in column bands of 224 inside the workgroup); w_ohwi: (64,7,7,3); bias: (64); out: (B,Hp,Wp,64) with
 * Hc = (H-1)/2 + 1, Hp = (Hc-1)/2 + 1 (same for W). */
int isx_stem7x7_pool_nhwc(const float* x, int64_t B, int H, int W, const float* w_ohwi, const float* bias, float* out,
                          isx_stream_t stream);

/* 1x1 stride-1 convolution of the inference trunk on channels-last activations, epilogue fused: one
 * fp32-MFMA GEMM over the M = B*H*W pixels,
 *   y[m][co] = act(sum_ci x[m][ci] * w[co][ci] + bias[co] + (residual ? residual[m][co] : 0)),
 * the sum a ci-ordered fp32 fma chain (bit-exact vs the oracle).  Replaces conv1 / conv3 / downsample of the
 * torchvision Bottleneck (the `features` trunk built by model/ModelDefinition.py and split by
 * model/nn_utils.py:56-71; run from model/siamese.py:20,107,151) together with their bias / residual / ReLU
 * passes.  x: (M,Cin), w: (Cout,Cin), y / residual: (M,Cout); y must not alias x or residual. */
int isx_conv1x1_nhwc(const float* x, int64_t M, int Cin, const float* w, int Cout, const float* bias,
                     const float* residual, int relu, float* y, isx_stream_t stream);

/* Last 1x1 convolution of a bottleneck block together with the block's 1x1 projection shortcut (torchvision Bottleneck
 * conv3 + downsample, same call sites) as ONE GEMM over the output pixels:
 *   y[m][co] = act( sum_c t[m][c] w_cat[co][c]  +  sum_c x[pix(m)][c] w_cat[co][K1 + c]  +  bias[co] ),
 * pix(m) = the input pixel (ho*stride, wo*stride) of output pixel m; one fp32 fma chain per output (t's channels first).
 * t: (B,Ho,Wo,K1); x: (B,H,W,K2); w_cat: (Cout, K1 + K2); y: (B,Ho,Wo,Cout); K1 % 32 == K2 % 32 == 0. */
int isx_conv1x1_dual_nhwc(const float* t, int K1, const float* x, int64_t B, int H, int W, int K2, int stride, const float* w_cat,
                          int Cout, const float* bias, int relu, float* y, isx_stream_t stream);

/* 3x3 convolution (padding 1, stride 1 or 2) of the same trunk, channels-last, as an implicit GEMM on the fp32
 * matrix cores: M = B*Ho*Wo output pixels, K = 9*Cin in (kh, kw, ci) order (one fp32 fma chain per output, bit-exact
 * vs the oracle), epilogue act(. + bias[co] + residual) fused.  Replaces conv2 of the torchvision Bottleneck and the
 * convolutions of BasicBlock (same call sites as isx_conv1x1_nhwc).  x: (B,H,W,Cin); w_ohwi: (Cout,3,3,Cin);
 * y / residual: (B,Ho,Wo,Cout), Ho = (H-1)/stride + 1.  Cin % 32 == 0. */
int isx_conv3x3_nhwc(const float* x, int64_t B, int H, int W, int Cin, const float* w_ohwi, int Cout, int stride,
                     const float* bias, const float* residual, int relu, float* y, isx_stream_t stream);

/* model/siamese.py:67-71 nn.AvgPool2d(feature_size2d, stride=1) of TuneClassifSub /
 * RegionDescriptorNet.  fmap: (B,C,H,W); out: (B,C,H-kh+1,W-kw+1). */
int isx_boxpool_s1(const float* fmap, int64_t B, int C, int H, int W, int kh, int kw, float* out,
                   isx_stream_t stream);

/* The same pooling on a channels-last map (what the NHWC trunk produces: no transpose in front of the region path).
 * fmap: (B,H,W,C); out: (B,H-kh+1,W-kw+1,C); C % 4 == 0, H*W*4 floats <= 64 KB.  Same window sums bit for bit. */
int isx_boxpool_s1_nhwc(const float* fmap, int64_t B, int C, int H, int W, int kh, int kw, float* out,
                        isx_stream_t stream);

/* ---- region path ----------------------------------------------------------- */

/* train/classif_regions.py:118-128: class-max map, spatial arg-max (smallest column,
 * then smallest row on ties), gather the K class scores there, L2.
 * cls: (B,K,Hp,Wp); desc: (B,K); loc: (B,2) = {row i1, col i2}. */
int isx_best_location_desc(const float* cls, int64_t B, int K, int Hp, int Wp, float eps, float* desc,
                           int64_t* loc, isx_stream_t stream);

/* model/siamese.py:191-194: c_maxv = c.max(1).view(-1); topk(min(len,k)) in canonical
 * order, for every image of a batch (the reference walks one image per call, :184).
 * cls: (B,K,Hp,Wp); flat_idx, score: (B,k); entries past Hp*Wp are (-1, -inf).  Hp*Wp <= 4096. */
int isx_region_topk(const float* cls, int64_t B, int K, int Hp, int Wp, int k, int64_t* flat_idx, float* score,
                    isx_stream_t stream);

/* model/siamese.py:199-219: for each of the k windows of each image, (row,col) = (idx / Wp, idx % Wp):
 * x[b, :, row:row+kh, col:col+kw] flattened (C,h,w) -> NormalizeL2 -> Shift.
 * fmap: (B,C,Hf,Wf); flat_idx: (B,k); rows: (B, k, C*kh*kw); shift (C*kh*kw) or NULL.
 * Windows with flat_idx < 0 produce zero rows.  The caller applies the Linear ONCE to all B*k rows
 * (the 100352 x D weight is then streamed once per batch, not once per window as in the reference). */
int isx_region_gather_l2(const float* fmap, int64_t B, int C, int Hf, int Wf, int kh, int kw, const int64_t* flat_idx, int k,
                         int Wp, const float* shift, float eps, float* rows, isx_stream_t stream);

/* Channels-last variants of the three region entry points above (same reference lines, same selection and tie-break):
 * cls: (B,Hp,Wp,K) -- the output of the 1x1-convolution classifier on the NHWC trunk --, fmap: (B,Hf,Wf,C).
 * isx_region_gather_l2_nhwc keeps a window in (h,w,C) order: rows (B,k,kh*kw*C), row[(a*kw + b)*C + c] = x[b, c, row+a, col+b];
 * shift_hwc is the Shift parameter permuted the same way (the caller permutes the Linear's weight columns once to match). */
int isx_best_location_desc_nhwc(const float* cls, int64_t B, int K, int Hp, int Wp, float eps, float* desc,
                                int64_t* loc, isx_stream_t stream);
int isx_region_topk_nhwc(const float* cls, int64_t B, int K, int Hp, int Wp, int k, int64_t* flat_idx, float* score,
                         isx_stream_t stream);
int isx_region_gather_l2_nhwc(const float* fmap, int64_t B, int C, int Hf, int Wf, int kh, int kw, const int64_t* flat_idx, int k,
                              int Wp, const float* shift_hwc, float eps, float* rows, isx_stream_t stream);

/* ---- retrieval ------------------------------------------------------------- */

/* test/classif_finetune_test.py:82 (and classif_regions_test.py:73,
 * siamese_descriptor_test.py:77, siamese_regions_test.py:76, utils/train_siamese.py:53,70)
 * sim = torch.mm(Q, G.t()).  Q: (M,D); G: (N,D); sim: (M,N).  fp32 MFMA. */
int isx_cosine_sim(const float* Q, int64_t M, const float* G, int64_t N, int D, float* sim, isx_stream_t stream);

/* Fused torch.mm -> sort/topk/max (same call sites + utils/metrics.py:10-13,33): the k
 * best gallery rows per query in canonical order, never materialising more than a
 * column chunk of the (M,N) matrix.  top_idx = idx_base + row of G.  Entries past N
 * are (-inf, -1).  1 <= k <= 1024.  ws from isx_cosine_topk_workspace (any size >= the
 * minimum it documents works; larger = fewer, bigger chunks). */
size_t isx_cosine_topk_workspace(int64_t M, int64_t N, int D, int k);
int isx_cosine_topk(const float* Q, int64_t M, const float* G, int64_t N, int D, int k, int64_t idx_base,
                    float* top_score, int64_t* top_idx, void* ws, size_t ws_bytes, isx_stream_t stream);

/* ---- exact top-k with a half-precision filter (csrc/fast.hip) -------------------------------------
 * Same call sites and the SAME RESULT, bit for bit, as isx_cosine_topk; the bulk of the arithmetic runs
 * on the fp16 matrix cores (16x the fp32 MFMA rate) and only the candidates that can reach the top-k are
 * re-scored with the exact fp32 fma chain.  Building blocks: */

/* x (B,D) fp32 -> h (B,D) fp16 (round to nearest even), norm2[b] >= sum_j x^2, amax[b] = max_j |x|. */
int isx_rows_to_f16(const float* x, int64_t B, int D, void* h, float* norm2, float* amax, isx_stream_t stream);

/* approximate scores Qh . Gh^T (fp16 operands, fp32 accumulate).  D % 8 == 0, 16-B aligned operands. */
int isx_cosine_sim_f16(const void* Qh, int64_t M, const void* Gh, int64_t N, int D, float* sim, isx_stream_t stream);

/* Gallery preparation, once per shard: Gh (N,D) fp16 = RNE(G * 2^s) with the power of two that brings
 * max|G| into [2^13, 2^14); gstats[4] (device) = {max_j |g_j|^2 (upper bound), max |G|,
 * max_j |g_j - fp16 image of g_j|^2 (what the conversion lost: enters the error bound of the search), 0}. */
int isx_gallery_to_f16(const float* G, int64_t N, int D, void* Gh, float* gstats, isx_stream_t stream);

/* The search.  Gh/gstats: the cached output of isx_gallery_to_f16, or both NULL (converted per call into
 * the workspace).  Output identical to isx_cosine_topk(Q, M, G, N, D, k, idx_base, ...) for every input:
 * rows whose candidate window cannot be proven complete, k > 128, D % 8 != 0, unaligned or tiny galleries
 * and out-of-range magnitudes all run the exact fp32 search. */
size_t isx_cosine_topk_fast_workspace(int64_t M, int64_t N, int D, int k, int have_gallery_f16);
/* Byte offset in that workspace of an int32 holding, after the search has completed, how many query rows took the exact
 * fp32 fallback ((size_t)-1: the whole call runs the fp32 search).  A caller whose data keeps falling back (dense clusters
 * of near-equal scores) should switch to isx_cosine_topk. */
size_t isx_cosine_topk_fast_fallback_offset(int64_t M, int64_t N, int D, int k, int have_gallery_f16);
int isx_cosine_topk_fast(const float* Q, int64_t M, const float* G, int64_t N, int D, int k, int64_t idx_base,
                         const void* Gh, const float* gstats, float* top_score, int64_t* top_idx, void* ws,
                         size_t ws_bytes, isx_stream_t stream);

/* utils/metrics.py:10-13 sim.max(1) / sim.kthvalue(...) on a materialised matrix: the k
 * best columns per row, canonical order.  sim: (M,N).  1 <= k <= 1024. */
int isx_topk_rows(const float* sim, int64_t M, int64_t N, int k, int64_t idx_base, float* top_score,
                  int64_t* top_idx, isx_stream_t stream);

/* utils/metrics.py:33 `_, ranked_list = sim[i].sort(dim=0, descending=True)` for every
 * row, made deterministic (canonical order).  ranked: (M,N) int64. */
size_t isx_rank_full_workspace(int64_t M, int64_t N);
int isx_rank_full(const float* sim, int64_t M, int64_t N, int64_t* ranked, void* ws, size_t ws_bytes,
                  isx_stream_t stream);

/* utils/metrics.py:25-45 avg_precision for every query (Oxford trapezoid AP, float64,
 * same operation order as the Python loop).  ap[i] = NaN where the reference returns
 * None (n_pos <= 0).  ranked: (M,N); qlab: (M); glab: (N). */
int isx_average_precision(const int64_t* ranked, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab,
                          int kth, double* ap, isx_stream_t stream);

/* The same AP values straight from the score matrix, WITHOUT the full sort: AP only depends on the
 * ranks of the query's positives (every other rank adds exactly 0 in utils/metrics.py:34-44), and
 * rank(p) = #{gallery keys above key(p)} is one streaming pass.  Bit-identical to
 * isx_rank_full + isx_average_precision.  Queries with more than 2048 positives get ap = -1.0
 * (use the sorted path for those); NaN where the reference returns None.  sim: (M,N). */
int isx_average_precision_sim(const float* sim, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab,
                              int kth, double* ap, isx_stream_t stream);

/* utils/train_siamese.py:74-76 sum_pos / (sum_neg + sum_pos) of test_descriptor_net, per
 * query row (the host adds the M row values in order, which keeps the result
 * deterministic): out[2*i] = sum_j sim[i][j] over label-equal pairs, out[2*i+1] = sum_j
 * sim[i][j] over all j; float64.  out: (M,2). */
int isx_masked_sums(const float* sim, int64_t M, int64_t N, const int32_t* qlab, const int32_t* glab, double* out,
                    isx_stream_t stream);

/* test/instance_avg.py:7-33 (DBA, database-side augmentation): new[i] = normalise(E[i] + sum_j w_j * E[best_j]) over the nn nearest
 * neighbours of i WITHIN ITS INSTANCE (same label), w_j = (nn - j) / (nn + 1), nn = min(k, group - 1) (k < 0: all), out = agg / (|agg| + 1e-10);
 * singleton instances and k = 0 keep their descriptor.  Only same-instance pairs are scored (the reference builds the N x N matrix and
 * masks it): scores = the canonical fma chain of isx_cosine_sim, ranking canonical, aggregation in the reference's sequential order.
 * emb, out: (N,D); order: (N) item indices sorted by (label, index); grp_begin / grp_size: (N) the run of `order` holding item i's
 * instance; max_group = max(grp_size) <= 1024. */
int isx_dba_groups(const float* emb, int64_t N, int D, const int32_t* order, const int32_t* grp_begin, const int32_t* grp_size,
                   int max_group, int k, float* out, isx_stream_t stream);

/* ---- multi-GPU (no reference counterpart: one torch.mm on one device,
 *      test/classif_finetune_test.py:82; BASELINE config 5 shards the gallery rows) --- */

/* Average precision against a gallery SHARDED by rows (one shard per GPU): the reference walks the full ranked list of ONE score row
 * (utils/metrics.py:25-45); the AP only depends on the ranks of the positives, and a rank is a count of gallery keys -- which adds over
 * shards.  Three steps around two collectives; with one shard they are isx_average_precision_sim cut at its synchronisation points, and for any
 * number of shards the float64 result is the same bits as the unsharded kernels' and the reference loop's.
 *   isx_ap_shard_positives  sim: (M, N) this shard's score rows, idx_base: global index of its first gallery row, glab: (N) its labels ->
 *                           keys: (M, cap) canonical keys (score, global index) of the shard's positives per query, 0 = empty slot, any order;
 *                           count: (M) how many there were (more than cap: the query is over the cap).              [all-gather keys, sum counts]
 *   isx_ap_shard_hist       keys_all: (M, W) the gathered keys of all shards (0 = empty) -> hist: (M, isx_ap_shard_max_positives()) int32: for
 *                           every key x of this shard's rows at or above the smallest positive, bucket #{positives > x} += 1.   [all-reduce sum]
 *   isx_ap_from_hist        hist summed over the shards (rows `ld` ints apart: the buckets past the largest n_lab need not travel), n_lab: (M)
 *                           positives per query over all shards -> ap: (M) float64; NaN where the
 *                           reference returns None (no positive left after kth - 1), -1.0 over isx_ap_shard_max_positives() positives.
 * Global indices below 2^32. */
int isx_ap_shard_max_positives(void);
int isx_ap_shard_positives(const float* sim, int64_t M, int64_t N, int64_t idx_base, const int32_t* qlab, const int32_t* glab, int cap,
                           uint64_t* keys, int32_t* count, isx_stream_t stream);
int isx_ap_shard_hist(const float* sim, int64_t M, int64_t N, int64_t idx_base, const uint64_t* keys_all, int W, int32_t* hist, isx_stream_t stream);
int isx_ap_from_hist(const int32_t* hist, int ld, const int32_t* n_lab, int64_t M, int kth, double* ap, isx_stream_t stream);

/* Merge P per-shard canonical top-k lists (after the RCCL all-gather) into the global
 * top-k.  scores, idx: (P,M,k) with GLOBAL indices, (-inf,-1) padding allowed;
 * out: (M,k).  P*k <= 4096. */
int isx_topk_merge(const float* scores, const int64_t* idx, int P, int64_t M, int k, float* out_s, int64_t* out_i,
                   isx_stream_t stream);

/* RCCL (xGMI) exchange of the per-shard lists: every rank contributes (M,k) scores + (M,k) global
 * indices, every rank receives (P,M,k) of each, rank-major -- the input of isx_topk_merge.  One
 * grouped ncclAllGather pair on `stream`.  RCCL is bound lazily (dlopen), so single-GPU users never
 * load it.  comm: an ncclComm_t, the caller's own or one made by isx_comm_init_rank from a unique
 * id (isx_comm_unique_id on rank 0, isx_comm_unique_id_bytes() bytes, shipped to the other ranks by
 * any means).  isx_comm_* are host-side set-up calls, not stream operations. */
int isx_comm_unique_id_bytes(void);
int isx_comm_unique_id(void* out_bytes);
int isx_comm_init_rank(void** comm, int nranks, int rank, const void* unique_id_bytes);
int isx_comm_destroy(void* comm);
int isx_shard_topk_allgather(void* comm, const float* s_local, const int64_t* i_local, int64_t M, int k, float* s_all,
                             int64_t* i_all, isx_stream_t stream);

/* The OTHER all-gather of the sharded search, on the same communicator: data-parallel extraction leaves every rank with its own
 * (rows, D) fp32 descriptor rows; the search needs the replicated query block (P * rows, D), rank-major (no reference counterpart: one
 * device computes test_embeddings whole, test/classif_finetune_test.py:80).  One ncclAllGather on `stream`.  With the query gather and
 * the result gather on ONE communicator and ONE stream, their order on every rank is program order -- no second communicator whose
 * kernels could be scheduled in a different order on different ranks.  rows_all: (P * rows, D). */
int isx_comm_allgather_rows(void* comm, const float* rows_local, int64_t rows, int64_t D, float* rows_all, isx_stream_t stream);

/* ---- siamese triplet training step (next scope row) ------------------------------- */

/* train/siamese_descriptor.py:94-128 (and siamese_regions.py:94-135): for every positive couple
 * (i1[c], i2[c]) the negative = arg-max over j of sim[i1][j] after excluding same-label items and, in
 * the semi-hard phase (epoch < train_epoch_switch), items with sim >= sim[i1][i2]; ties -> smallest
 * index; neg[c] = -1 when everything is excluded (caller picks a random negative).
 * sim: (N,N); labels: (N) int32; i1, i2, neg: (n_couples) int64. */
int isx_mine_negatives(const float* sim, int64_t N, const int32_t* labels, const int64_t* i1, const int64_t* i2,
                       int64_t n_couples, int semi_hard, int64_t* neg, isx_stream_t stream);

/* The same mining when the N x N matrix is never built whole (it exceeds the memory budget the reference answers with a CPU
 * fallback, utils/train_siamese.py:30-43): sim_rows = rows [row_base, row_base + rows) of the matrix, row-major with
 * stride N; every anchor i1[c] lies inside that range; i1 / i2 / neg are absolute gallery indices. */
int isx_mine_negatives_rows(const float* sim_rows, int64_t N, int64_t row_base, int64_t rows, const int32_t* labels,
                            const int64_t* i1, const int64_t* i2, int64_t n_couples, int semi_hard, int64_t* neg, isx_stream_t stream);

/* model/custom_modules.py:153-171 TripletLossFun.forward, per-row part: loss_rows[b] = max(0, l_b) with
 * l_b = a.n - a.p + margin (normalized) or (|a-p|^2 - |a-n|^2 + 2 margin)/2.  The caller sums the rows
 * (and divides by B for size_average).  anchor, pos, neg: (B,D). */
int isx_triplet_loss_fwd(const float* anchor, const float* pos, const float* neg, int64_t B, int D, float margin,
                         int normalized, float* loss_rows, isx_stream_t stream);

/* model/custom_modules.py:173-203 TripletLossFun.backward: g_a = n - p, g_p = -a (p - a), g_n = a (a - n)
 * on rows with loss_rows > 0, zero elsewhere, times `scale`. */
int isx_triplet_loss_bwd(const float* anchor, const float* pos, const float* neg, const float* loss_rows, int64_t B, int D,
                         float scale, int normalized, float* g_anchor, float* g_pos, float* g_neg, isx_stream_t stream);
/* The same with the incoming gradient as a device scalar: gradients times scale * scale_dev[0] (no host read-back of grad_output). */
int isx_triplet_loss_bwd_dev(const float* anchor, const float* pos, const float* neg, const float* loss_rows, int64_t B, int D,
                             float scale, const float* scale_dev, int normalized, float* g_anchor, float* g_pos, float* g_neg,
                             isx_stream_t stream);

/* The same loss for ALL micro-batches ("leaves") of an optimizer step in ONE launch (utils/train_general.py:51-61 runs criterion + backward once per
 * micro-batch: model/custom_modules.py:153-203 each time).  d: (leaves * 3 k, D), leaf by leaf the k anchor rows, the k positive rows, the k
 * negative rows (the order the reference's batch carries them, train/siamese_descriptor.py:112-128).  loss_leaf[l] = the leaf's row losses added in
 * row order (before any averaging); dd (same shape as d) = the gradient rows, each row as isx_triplet_loss_bwd forms it with
 * scale = scale_a * scale_b (1 / k when the loss is averaged, times the weight of the leaf in the mini-batch). */
int isx_triplet_leaves(const float* d, int leaves, int k, int D, float margin, int normalized, float scale_a, float scale_b, float* loss_leaf,
                       float* dd, isx_stream_t stream);

/* model/custom_modules.py:59-67 NormalizeL2Fun.backward: with n2 = sum_j x_j^2 + eps and c = sum_j x_j dy_j,
 * dx = (n2 dy - x c) / (n2 sqrt(n2)).  x, dy, dx: (B, D). */
int isx_l2norm_rows_bwd(const float* x, const float* dy, int64_t B, int64_t D, float eps, float* dx, isx_stream_t stream);

/* ---- backward pass of the TRAINABLE trunk suffix (siamese training, reference configuration) ---------------------------------
 * The reference trains layer4 of the ResNet (train/siamese_descriptor_p.py:14-17,48 -> model/nn_utils.py:5-23) and leaves the
 * backward pass to torch autograd: `loss.backward()` in utils/train_general.py:51-61.  These entries are that backward pass for
 * the 1x1 / 3x3 convolutions of the residual blocks on channels-last activations, with the (eval-mode) BatchNorm folded into the
 * convolution: forward = isx_conv1x1_nhwc / isx_conv3x3_nhwc / isx_conv1x1_dual_nhwc on w' = w * s, b' = beta - mean * s,
 * s = gamma / sqrt(var + eps).  Every sum has a fixed order: a micro-batch's gradient does not depend on the launch around it. */

/* Gradient of a 1x1 convolution wrt its input: dx = (dz . W' (+ add)) . [mask > 0].  dz: (M, Cout); wt = W'^T as (Cin, Cout)
 * row-major; add (identity-shortcut gradient) and mask (OUTPUT of the ReLU below this convolution): (M, Cin) or NULL; dx: (M, Cin). */
int isx_conv1x1_dgrad_nhwc(const float* dz, int64_t M, int Cout, const float* wt, int Cin, const float* add, const float* mask,
                           float* dx, isx_stream_t stream);

/* Gradient of a 3x3 convolution (padding 1) wrt its input, as a stride-1 3x3 convolution of dz with
 * wt[ci][kh][kw][co] = w'[co][2-kh][2-kw][ci]; a stride-2 layer passes dz zero-upsampled to the input grid.
 * dz: (B,H,W,Cout), wt: (Cin,3,3,Cout), mask / dx: (B,H,W,Cin); mask as above or NULL.  Cout % 32 == 0. */
int isx_conv3x3_dgrad_nhwc(const float* dz, int64_t B, int H, int W, int Cout, const float* wt, int Cin, const float* mask,
                           float* dx, isx_stream_t stream);

/* Gradient of a STRIDE-2 3x3 convolution (padding 1) wrt its input, second half: dcol[p][tap][ci] = sum_co dz[p][co] w'[co][tap][ci] is
 * one GEMM over the OUTPUT pixels (isx_conv1x1_dgrad_nhwc with wt = w' as (9*Cin, Cout)); this entry gathers, for every input pixel, the
 * taps that reach it (1, 2 or 4 of the 9, in (kh, kw) order) and applies the ReLU mask: a quarter of the matrix work of the
 * zero-upsampled form.  dcol: (B*Ho*Wo, 9, Cin); mask (or NULL) / dx: (B,H,W,Cin); Cin % 4 == 0. */
int isx_conv3x3_s2_col2im_nhwc(const float* dcol, int64_t B, int H, int W, int Cin, const float* mask, float* dx, isx_stream_t stream);

/* Weight gradient of `leaves` micro-batches in one launch, each split over its pixels: the B images are `leaves` consecutive groups;
 * dw[l][s][co][tap][ci] = sum over the output pixels p of split s of leaf l of dz[p][co] * x[src(p, tap)][ci] and db[l][s][co] = sum over
 * the same pixels of dz[p][co] (the bias gradient, a by-product of the staged dz tiles), s < S = isx_conv_wgrad_splits(pixels of ONE
 * leaf, Cin, Cout, taps).  A leaf's partials are the same bits whether it is launched alone or with its siblings (its k-tiles start at
 * its first pixel; S and the tile shape depend on the leaf's shape only); isx_bn_fold_backward adds them in split order: a fixed
 * summation tree, no atomics.  taps = 1: 1x1 convolution with `stride` (no padding); taps = 9: 3x3, padding 1, `stride`.
 * x: (B,H,W,Cin), dz: (B,Ho,Wo,Cout), dw: (leaves,S,Cout,taps,Cin) -- per partial the layout of the forward kernels' weights;
 * db: (leaves,S,Cout) or NULL.  Pixels are summed in index order (k-ordered fp32 fma chain).  Cin, Cout % 64 == 0; B % leaves == 0. */
int isx_conv_wgrad_splits(int64_t pixels, int Cin, int Cout, int taps);
int isx_conv_wgrad_nhwc(const float* dz, const float* x, int64_t B, int leaves, int H, int W, int Cin, int Cout, int taps, int stride,
                        float* dw, float* db, isx_stream_t stream);

/* Backward of y = relu(.) at a block output: dz = dy . [y > 0].  n floats (n % 4 == 0, 16-B aligned); dz == dy allowed. */
int isx_relu_grad(const float* dy, const float* y, int64_t n, float* dz, isx_stream_t stream);

/* Chain rule of the BatchNorm fold, per leaf: from the partial gradients (dwp, db) of the folded convolution (as written by
 * isx_conv_wgrad_nhwc; d = sum_s dwp[l][s], b = sum_s db[l][s]) to the gradients of the convolution weight and the BatchNorm affine
 * parameters: gw_l (+)= d * scale, ggamma_l (+)= (<d, w> - mean * b) * istd, gbeta_l (+)= b, with scale = gamma * istd,
 * istd = 1 / sqrt(running_var + eps).  dwp: (leaves,splits,Cout,taps,Cin); db: (leaves,splits,Cout); w: (Cout,Cin,taps) (nn.Conv2d's
 * layout); gw / ggamma / gbeta: the gradient tensors of leaf 0 (parameter layouts), leaf l at + l * leaf_stride floats (one flat gradient
 * buffer per leaf; leaves == 1: plain tensors); scale, mean, istd: (Cout); accumulate != 0 adds into them (gradient accumulation). */
int isx_bn_fold_backward(const float* dwp, const float* db, int leaves, int splits, const float* w, const float* scale, const float* mean,
                         const float* istd, int Cout, int Cin, int taps, int accumulate, int64_t leaf_stride, float* gw, float* ggamma,
                         float* gbeta, isx_stream_t stream);

/* ---- the descriptor head's Linear for all micro-batches of a training step at once (reference model/siamese.py:104-114,
 * Linear(100352 -> 2048); torch / the reference run it once per micro-batch: the 822 MB weight crosses HBM 16 times per step) ---- */

/* y = x . w^T + bias with every row's value independent of how many rows ride along: S = isx_head_linear_splits(K) partial sums per
 * output (k-ordered fp32 fma chains over consecutive K ranges), added in split order.  xT: (K, Mp) = x TRANSPOSED, Mp >= M a multiple
 * of 64 (padding columns: any finite values); w: (N, K) as nn.Linear stores it; bias: (N) or NULL; y: (M, N); ws: S * Mp * N floats.
 * K % 32 == 0, N % 64 == 0. */
int isx_head_linear_splits(int64_t K);
int isx_head_linear_fwd(const float* xT, int64_t M, int64_t Mp, int64_t K, const float* w, int N, const float* bias, float* y,
                        float* ws, size_t ws_bytes, isx_stream_t stream);

/* The same with x as stored, (M, K) row-major (no transposed copy of the activation), any M; ws: isx_head_linear_rows_workspace(M, K, N)
 * bytes.  Bit-identical to isx_head_linear_fwd.  Also the inference path of DescriptorNet / RegionDescriptorNet
 * (model/siamese.py:117-122, 215-220): ONE implementation of the layer, a row's descriptor independent of the batch it rides in. */
size_t isx_head_linear_rows_workspace(int64_t M, int64_t K, int N);
int isx_head_linear_fwd_rows(const float* x, int64_t M, int64_t K, const float* w, int N, const float* bias, float* y, float* ws,
                             size_t ws_bytes, isx_stream_t stream);

/* The input gradient of the same Linear for all rows at once: dx[m][k] = sum_n dy[m][n] w[n][k] whatever M, summed in two levels: the N output
 * features form isx_head_groups(N) consecutive groups (8, or 1 when N % 256 != 0), one k-ordered chain per group, the group sums added in group
 * order.  dyT: (N, Mp) = dy TRANSPOSED, zero-padded to Mp (a multiple of 64); w: (N, K); dx: (Mp, K).  K % 64 == 0.
 * isx_head_linear_dgrad_parts: the chains of `groups` groups of Ng features, not added -- parts (groups, Mp, K) -- for a head sharded by output
 * features across data-parallel ranks (isx/shard_head.py): adding all groups' parts in group order gives isx_head_linear_dgrad's bits. */
int isx_head_groups(int64_t N);
int isx_head_linear_dgrad(const float* dyT, int64_t Mp, int N, const float* w, int64_t K, float* dx, isx_stream_t stream);
int isx_head_linear_dgrad_parts(const float* dyT, int64_t Mp, int Ng, int groups, const float* w, int64_t K, float* parts,
                                isx_stream_t stream);

/* The weight gradient of the same Linear over the R rows of a whole mini-batch AND torch.optim.SGD's update of the weight, as ONE
 * kernel (reference: the optimizer of train/siamese_descriptor.py:136-139 stepped from utils/train_general.py:53 on the 822 MB weight of
 * model/siamese.py:104-114): g[n][k] = sum_r dy[r][n] x[r][k] (one fp32 fma chain over the rows in row order), then per element
 *   g += weight_decay * w;  buf = first ? g : momentum * buf + (1 - dampening) * g;  w -= lr * (nesterov ? g + momentum * buf : buf)
 * dy: (R, N), x: (R, K), w / mom: (N, K) updated in place (mom NULL when momentum == 0).  N % 64 == 0, K % 128 == 0.  No dW tensor:
 * 4 passes over the weight's size per step instead of 7. */
int isx_head_sgd_step(const float* dy, const float* x, int64_t R, int N, int64_t K, float* w, float* mom, int first, float lr,
                      float momentum, float dampening, float weight_decay, int nesterov, isx_stream_t stream);

/* out[l][c] = sum_{r < R} x[l * R + r][c]: column sums of `leaves` consecutive groups of R rows (per-micro-batch bias / Shift
 * gradients), rows added in order.  x: (leaves * R, C); out: (leaves, C). */
int isx_colsum_leaves(const float* x, int leaves, int R, int64_t C, float* out, isx_stream_t stream);

/* out[c] = the sum of rows[0 .. L)[c] in the canonical TREE order of the data-parallel training step (isx/dp.py: a node is its left subtree
 * plus its right subtree, split at L / 2 -- the order in which the per-micro-batch gradients of reference utils/train_general.py:51-74 are
 * added, chosen so that 1, 2, 4, 8 ranks produce the same bits).  rows: L rows of n floats, `stride` floats apart; out may be rows' row 0.
 * 1 <= L <= 16.  One pass over the L rows instead of L - 1 add passes. */
int isx_tree_sum_rows(const float* rows, int L, int64_t stride, int64_t n, float* out, isx_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* ISX_H_ */
