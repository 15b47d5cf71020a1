/*
 * ssad.h -- C ABI of libssad_hip.so, the MI355X (gfx950) kernels behind the
 * self-supervised anomaly-detection hot path.
 *
 * The reference (gabry1998/Self-Supervised-Anomaly-Detection) is pure Python and has no
 * FFI of its own: its numerics are PyTorch / torchvision / scikit-learn calls.  Each entry
 * point below replaces one such call site (cited as path:line relative to the reference
 * root).  Conventions:
 *   - plain pointers + sizes, no torch types; every pointer is a DEVICE pointer unless
 *     the name ends in _host;
 *   - the caller owns every buffer (inputs, outputs, workspaces);
 *   - `stream` is a hipStream_t passed as void*; launches are asynchronous on it and the
 *     library never synchronises;
 *   - return 0 on success, non-zero on error; ssad_last_error() returns a message for
 *     the calling thread.  Never aborts.
 *   - activations are NHWC fp32 inside the library; conv weights are OHWI fp32
 *     ([Cout][KH][KW][Cin]); images enter as NCHW fp32 exactly as the reference's
 *     Dataset emits them (src/self_supervised/datasets.py:394).
 */
#ifndef SSAD_H
#define SSAD_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSAD_VERSION 100

int ssad_version(void);
const char* ssad_last_error(void);

/* ---- weight repacking (checkpoint tensors keep the reference's OIHW shapes, SURVEY s.5) ---- */
/* OIHW -> OHWI.  Linear weights [out][in] are already "OHWI" with KH=KW=1. */
int ssad_repack_oihw_to_ohwi(const float* w_oihw, float* w_ohwi, int O, int I, int KH, int KW, void* stream);
int ssad_repack_ohwi_to_oihw(const float* w_ohwi, float* w_oihw, int O, int I, int KH, int KW, void* stream);
/* conv1 7x7 OIHW [64][3][7][7] -> MFMA K-order [168][64] (kx padded 7->8 with zeros). */
int ssad_pack_stem_weight(const float* w_oihw, float* wk, void* stream);
/* The same pack from an OHWI filter [64][7][7][3]: how the training step's parameter arena holds conv1 (models.py:224 under trainer.fit);
 * spares the step an OIHW copy of the weight per iteration (round 6). */
int ssad_pack_stem_weight_ohwi(const float* w_ohwi, float* wk, void* stream);

/* ---- forward kernels ---- */
/* Replaces: extract_patches (src/self_supervised/functional.py:77-82) + reshape
 * (src/self_supervised/models.py:212-214) + F.interpolate(x, 64, 'nearest') (models.py:217-219)
 * + resnet conv1/bn1/relu (models.py:224; torchvision resnet18 stem), fused: the patch window and
 * the nearest resize are applied in the loader, nothing is materialised.
 *   img      : [B][3][H][W] NCHW fp32
 *   patch_dim/patch_stride : sliding window (32, 8); patch_dim = 0 -> whole image is one sample
 *   Hv, Wv   : size after the nearest resize (== window size when no resize happens)
 *   wk       : ssad_pack_stem_weight output
 *   scale/shift : per-channel affine applied to the conv output (eval-mode BN folded); NULL -> raw conv
 *   out      : [Nsamp][Ho][Wo][64] NHWC, Ho = (Hv-1)/2+1.  Nsamp = B * patches per image.
 *   hwnc     : 1 -> write the position-major layout [Ho][Wo][Nsamp][64] (patch-scoring trunk, see
 *              ssad_conv_igemm_fwd_hwnc); the same flag selects the layout in ssad_maxpool3x3s2_fwd / ssad_gap_fwd.
 */
int ssad_stem_fwd(const float* img, int B, int H, int W, int patch_dim, int patch_stride, int Hv, int Wv,
                  const float* wk, const float* scale, const float* shift, int relu, int hwnc, float* out, void* stream);
/* The same conv1 in training (no patch window, no affine): raw z plus the train-mode statistics of bn1 in one pass
 * (models.py:224 under trainer.fit).  workspace: ssad_stem_stats_rows() * 128 doubles (one [2][64] row per workgroup). */
int ssad_stem_stats_rows(void);
int ssad_stem_fwd_stats(const float* img, int B, int H, int W, int Hv, int Wv, const float* wk, float* out, float eps,
                        float momentum, float* mean, float* invstd, float* running_mean, float* running_var, double* workspace,
                        void* stream);

/* Patch-scoring specialisation of the stem (32x32 windows, exact 2x nearest upsample): the same call sites as
 * ssad_stem_fwd + ssad_maxpool3x3s2_fwd fused into one kernel.  conv7x7/2 over the upsampled window is evaluated as
 * the equivalent 4x4 stride-1 conv over the 32x32 source with pre-summed weights (wf = ssad_pack_stem_weight_folded),
 * BN+ReLU+max-pool are applied in LDS; out = [Nsamp][16][16][64] (or [16][16][Nsamp][64] when hwnc). */
int ssad_pack_stem_weight_folded(const float* w_oihw, float* wf, void* stream);
int ssad_stem_patch_pool_fwd(const float* img, int B, int H, int W, int patch_stride, const float* wf, const float* scale,
                             const float* shift, int hwnc, float* out, void* stream);
/* The same without the pooled positions skip_lo <= py, px <= skip_hi of every patch (neither pooled nor stored): with layer1 shared between
 * overlapping patches the first ring conv (ssad_conv_igemm_fwd_hwnc_ring) reads the pooled map within one position of the outputs it
 * computes -- the interior is read by nobody. */
int ssad_stem_patch_pool_fwd_ring(const float* img, int B, int H, int W, int patch_stride, const float* wf, const float* scale,
                                  const float* shift, int hwnc, int skip_lo, int skip_hi, float* out, void* stream);
/* Only the BORDER of that pooled map -- rows / columns 0, 1 and 15 of every patch, the positions where the patch's own zero padding makes
 * it differ from the per-image pooled map -- written position-major [16][16][Nsamp][64]; the other positions are left untouched
 * (ssad_patch_gather_hwnc_band copies rows / columns 2-3 and 13-14 from the per-image map; nobody reads the rest).  Values bit-identical
 * to ssad_stem_patch_pool_fwd at the positions written. */
int ssad_stem_patch_border_fwd(const float* img, int B, int H, int W, int patch_stride, const float* wf, const float* scale,
                               const float* shift, float* out, void* stream);

/* Replaces nn.MaxPool2d(3, 2, 1) of the torchvision stem (models.py:224).  NHWC, or [H][W][N][C] when hwnc. */
int ssad_maxpool3x3s2_fwd(const float* in, float* out, int64_t N, int H, int W, int C, int hwnc, void* stream);

/* Replaces every nn.Conv2d(3x3 / 1x1, bias=False) + eval BatchNorm2d + residual add + ReLU of the
 * BasicBlocks (models.py:224), and every nn.Linear (+BatchNorm1d, +ReLU) of concatenator /
 * latent_space / classifier (models.py:247-249) with H=W=KH=KW=1.  Implicit GEMM on
 * v_mfma_f32_32x32x2_f32: out[m][co] = act( (sum_k in[m,k] * w[co,k]) * scale[co] + shift[co] + residual[m][co] ).
 * Requires Cin % 32 == 0.  scale/shift/residual may be NULL.
 */
int ssad_conv_igemm_fwd(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                        const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                        int stride, int pad, void* stream);

/* Same contraction with every activation tensor (in, out, residual) stored position-major, [H][W][N][C]: the layout
 * of the patch-scoring trunk (thousands of patches with 16x16 .. 2x2 maps).  A workgroup's 128 rows are 128 patches
 * at one output position: contiguous in HBM for every filter tap, and taps that fall into the zero padding are
 * skipped as whole K-steps (exact: only x*0 products are dropped). */
int ssad_conv_igemm_fwd_hwnc(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                             const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride, int pad, void* stream);

/* Round 6: the layer1 convolutions of the patch-scoring pass without the arithmetic that overlapping patches share.
 * PeraNet.forward in patch mode (models.py:211-224) runs resnet layer1 on 841 windows of 32 x 32 pixels at stride 8 per image
 * (functional.py:77-82); away from a window's own zero-padded border a conv output is the same sum over the same pixels in every
 * window covering them, i.e. the value of the conv run ONCE over the whole image's map.  ssad_conv_igemm_fwd_hwnc_ring computes only
 * the output positions outside the square skip_lo <= oy, ox <= skip_hi (position-major tensors as ssad_conv_igemm_fwd_hwnc; the
 * positions inside are not touched), ssad_patch_gather_hwnc copies the positions lo <= u, v <= hi of every patch from the per-image
 * dense map: out[u][v][n][:] = dense[b][shift * pr + u][shift * pc + v][:], n = (b * prow + pr) * pcol + pc, dense NHWC
 * [B][Hd][Wd][C], out [H][W][N][C].  Exact: no product is approximated, only not repeated. */
int ssad_conv_igemm_fwd_hwnc_ring(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                                  const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                                  int stride, int pad, int skip_lo, int skip_hi, void* stream);
int ssad_patch_gather_hwnc(const float* dense, float* out, int64_t B, int prow, int pcol, int shift, int Hd, int Wd, int C,
                           int H, int W, int lo, int hi, void* stream);
/* The same copy without the inner square ilo <= u, v <= ihi (ilo > ihi: none): the ring conv that follows reads its input within one
 * position of the outputs it computes, the deep interior of an intermediate layer1 map is read by nobody. */
int ssad_patch_gather_hwnc_band(const float* dense, float* out, int64_t B, int prow, int pcol, int shift, int Hd, int Wd, int C,
                                int H, int W, int lo, int hi, int ilo, int ihi, void* stream);

/* Which exact-fp32 instantiation ssad_conv_igemm_fwd (hwnc = 0) / ssad_conv_igemm_fwd_hwnc (hwnc = 1) gives a problem:
 * BM * 100000 + BN * 100 + BK of the workgroup tile, negated when its rows are position-major (hwnc = 2: the tile of a
 * ssad_conv_igemm_fwd_hwnc_ring launch).  Measurement aid only
 * (bench.py names the instantiations its roofline figure sums over); no reference counterpart. */
int ssad_conv_igemm_tile(int64_t N, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int hwnc);

/* Replaces F.adaptive_avg_pool2d(., (1,1)) + flatten + torch.cat (models.py:227-245):
 * out[n*out_stride + out_offset + c] = mean over HW of in[n][hw][c]. */
int ssad_gap_fwd(const float* in, float* out, int64_t N, int HW, int C, int out_stride, int out_offset, int hwnc,
                 void* stream);

/* ---- scoring ---- */
/* Replaces sklearn NearestNeighbors(metric='cosine').kneighbors + torch.mean (models.py:352-370).
 * l2norm: out[i] = x[i] / ||x[i]||.  knn3: given sim[Nq][Nb] = qn . bn, writes mean of the 3 smallest
 * clip(1 - sim, 0, 2) per row. */
int ssad_l2_normalize_rows(const float* x, float* out, int64_t N, int D, void* stream);
int ssad_cosine_knn_mean(const float* sim, float* out, int64_t Nq, int Nb, int k, void* stream);

/* Replaces tools.upsample (src/self_supervised/tools.py:394-399): relu(gaussian_blur(k, sigma=0.15k+0.35,
 * reflect)) then bilinear (align_corners=False) to target x target.  maps [n][h][w] -> out [n][target][target]. */
int ssad_blur_relu_bilinear(const float* maps, float* out, int n, int h, int w, int ksize, int target, void* stream);

/* Grad-CAM contraction (src/self_supervised/gradcam.py:39-43): out[b][pos] = sum_k alpha[b][k] * act[b][pos][k] for the
 * NHWC layer4 activations act [B][HW][C]; alpha rows are alpha_stride floats apart (a slice of the pooled-feature
 * gradient).  ReLU + the bilinear resize to the input size are ssad_blur_relu_bilinear with ksize = 1. */
int ssad_gradcam_map(const float* act, const float* alpha, float* out, int64_t B, int HW, int C, int alpha_stride,
                     void* stream);

/* Replaces sklearn roc_curve + auc as called for pixel / image AUROC (src/self_supervised/metrics.py:49-56,
 * src/self_supervised/tools.py:76-98) when the scores are already on the GPU: radix sort + tie-aware rank sum, exact
 * integer counts reduced in fp64.  labels: uint8, non-zero = anomalous.  out[0] = AUROC, out[1] = #positives. */
int64_t ssad_auroc_workspace(int64_t n);
int ssad_auroc(const float* scores, const uint8_t* labels, int64_t n, void* workspace, int64_t workspace_bytes, double* out,
               void* stream);

/* ---- training step (forward in train mode, backward, update) ---- */
/* Replaces autograd's conv2d/linear input-gradient (loss.backward() inside pl.Trainer.fit, tools.py:270,:303).
 * w_flipT = ssad_flip_transpose_weight(w_ohwi).  dx = dgrad(dy) (+ residual).  Cout % 32 == 0. */
int ssad_flip_transpose_weight(const float* w_ohwi, float* out, int O, int I, int KH, int KW, void* stream);
/* ... for n filters of one arena, one launch per 32 of them: desc[k] = {src offset, dst offset, O, I, KH, KW} (host memory). */
int ssad_flip_transpose_batch(const float* src, float* dst, const int64_t* desc, int n, void* stream);
int ssad_conv_igemm_dgrad(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N, int Hy, int Wy,
                          int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
/* Split-bf16 ("bf16x3") forms of the forward / position-major forward (hwnc != 0) and dgrad contractions: fp32 tensors in
 * and out; every operand x is staged as hi = bf16(x), lo = bf16(x - hi) and every product is hi*hi + hi*lo + lo*hi on
 * the bf16 matrix cores with fp32 accumulation -- ~2^-17 relative error per product at 3/16 of the fp32-MFMA time.
 * Opt-in (SSAD_MATH=bf16x3 / Trainer(precision="bf16x3")); held to the same 1e-4 parity bar as the exact fp32 path. */
int ssad_conv_igemm_fwd_x3(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                           const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int hwnc, void* stream);
int ssad_conv_igemm_dgrad_x3(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N, int Hy,
                             int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
/* Three-way split ("bf16x6"): x = hi + mid + lo (24 significant bits) and the six products of weight >= 2^-18; what is
 * dropped is ~2^-25 of each product, i.e. fp32-faithful products at 6/16 of the fp32-MFMA time. */
int ssad_conv_igemm_fwd_x6(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                           const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                           int stride, int pad, int hwnc, void* stream);
int ssad_conv_igemm_dgrad_x6(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N, int Hy,
                             int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
/* BatchNorm over a few hundred rows (the BatchNorm1d layers of the projection head on a training batch, models.py:65-95) in one
 * launch each way: statistics + running-statistics update + apply (+ ReLU) forward; the two column sums, their use and optionally
 * the bias gradient of the Linear in front (dbias = column sums of dz) backward.  zmask_beta != NULL: the layer's ReLU mask is
 * recomputed from z (no residual in between).  ssad_bn_small_ok(R, C): R <= 512 rows, C a multiple of 32. */
int ssad_bn_small_ok(int64_t R, int C);
int ssad_bn_small_fwd(const float* z, const float* gamma, const float* beta, float* y, float* mean, float* invstd,
                      float* running_mean, float* running_var, int64_t R, int C, float eps, float momentum, int relu, void* stream);
int ssad_bn_small_bwd(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                      const float* zmask_beta, float* dbeta, float* dgamma, float* dbias, float* dz, int64_t R, int C, void* stream);
/* Linear layers over a training batch's few rows (the projection head and classifier, models.py:65-99, :247-252): 1 x 1 layers on
 * 1 x 1 maps with N <= ssad_linear_small_max_rows() rows are served by a dedicated kernel (contraction dealt over the waves of a
 * 32 x 32 output tile) inside ssad_conv_igemm_fwd / _fwd_stats / _dgrad -- those then accept any Cin (Cout for dgrad) % 4 == 0 --
 * and their weight gradient by ssad_linear_wgrad_small: dw[Cout][Cin] (+)= dy[M][Cout]^T x[M][Cin], one launch, no slab.
 * SSAD_LINEAR_SMALL=0 in the environment switches the range off (max_rows 0). */
int ssad_linear_small_max_rows(void);
int ssad_linear_wgrad_small(const float* dy, const float* x, float* dw, int64_t M, int Cin, int Cout, int accumulate, void* stream);
/* the same with its operands rounded to bf16 (round = 1) / fp16 (round = 2) while loaded: the precision-16 step's linear layers */
int ssad_linear_wgrad_small_r(const float* dy, const float* x, float* dw, int64_t M, int Cin, int Cout, int accumulate, int round,
                              void* stream);
/* Replaces autograd's conv2d/linear weight-gradient.  Two launches: partial tiles per pixel split into
 * slab[splits][Cout][KH*KW*Cin], then a fixed-order sum written as OIHW (to_oihw=1, checkpoint layout) or OHWI.
 * Guard convention (round 5): wherever an entry point reads a SECOND tensor over extents it derives from the first one's (dy from
 * x's N, H, W and the filter geometry; dz from the image size; the pooled gradient from z's size), the caller also states how many
 * elements that second buffer holds (`dy_elems`, `dz_elems`, `dpool_elems`); a count that does not match the derived extents is an
 * error return, not an out-of-bounds read. */
int ssad_wgrad_splits(int64_t M, int Cin, int Cout, int KH, int KW);
int ssad_wgrad_splits_bf16(int64_t M, int Cin, int Cout, int KH, int KW);   /* for ssad_conv_wgrad_bf16 / _x3 */
int ssad_conv_wgrad(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                    int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream);
int ssad_conv_wgrad_x3(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream);   /* split-bf16 products, see ssad_conv_igemm_fwd_x3 */
int ssad_conv_wgrad_x6(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                       int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream);   /* three-way split, see ssad_conv_igemm_fwd_x6 */
int ssad_wgrad_reduce(const float* slab, float* dw, int splits, int Cout, int Kpad, int KH, int KW, int Cin, int to_oihw,
                      int accumulate, void* stream);
/* Several such reductions in ONE launch (round 6): desc[6 k ..] = slab pointer, output pointer (as integers), splits, Cout, Kpad,
 * Kreal = KH * KW * Cin.  Nothing on the backward pass's critical path reads a weight gradient, so the training step collects the
 * reductions of its weight-gradient kernels and runs them where the gradients are first needed (optimizer / a gradient bucket).
 * Same order of additions per output as ssad_wgrad_reduce: bit-identical.  OHWI outputs, no accumulation; Kreal % 4 == 0, Kpad % 4 == 0,
 * 16-byte aligned pointers.  Same autograd nodes as ssad_conv_wgrad (Conv2d weight gradients under trainer.fit, models.py:256-277). */
int ssad_wgrad_reduce_batch(const int64_t* desc, int n, void* stream);
/* bf16-operand forms of the three MFMA entry points above (fp32 tensors in HBM; operands rounded to bf16 while staging,
 * fp32 accumulate; v_mfma_f32_32x32x16_bf16).  This is what torch.autocast does to the same Conv2d / Linear call sites
 * under the reference's pl.Trainer(precision=16) (src/self_supervised/tools.py:263, :296). */
int ssad_conv_igemm_fwd_bf16(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                             const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride, int pad, void* stream);
int ssad_conv_igemm_dgrad_bf16(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N, int Hy,
                               int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
int ssad_conv_wgrad_bf16(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                         int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream);
/* Halo-tile direct convolution for the 64 -> 64 channel 3x3 / stride 1 / pad 1 layers (torchvision BasicBlock conv3x3 of
 * ResNet-18 layer1, models.py:224 of the reference under trainer.fit, and their input gradients -- call it with
 * ssad_flip_transpose_weight(w) and dy): out = conv(T(in)) (+ residual), NHWC fp32, OHWI weights, exact fp32 MFMA.
 * res_mask (optional, with residual): nibble mask of ssad_bn_apply_fwd_mask -- only the masked elements of the residual are added.
 * T = identity, or relu((x - tr_mean) * tr_invstd * tr_gamma + tr_beta) per input channel (the train-mode BatchNorm + ReLU of
 * the producing layer, applied while the input tile is staged); `emit` (optional) receives T(in).  stats_ws != NULL: also
 * the train-mode BatchNorm statistics of the output (ssad_conv3x3_c64_stats_rows(N, H, W) * 128 doubles of workspace),
 * mean / invstd / running statistics as ssad_conv_igemm_fwd_stats produces them. */
int64_t ssad_conv3x3_c64_stats_rows(int64_t N, int H, int W);
int ssad_conv3x3_c64(const float* in, const float* w_ohwi, float* out, const float* residual, const uint8_t* res_mask, const float* tr_mean,
                     const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit, int64_t N, int H, int W,
                     double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                     float* running_var, void* stream);
/* The same with op = 1 / 2: bf16 / fp16 OPERANDS (fp32 tensors, rounded while staged; fp32 accumulation, statistics, transform and
 * emit) -- the layer1 convolutions of the reference's fp16-autocast training (pl.Trainer(precision=16), tools.py:263); op = 0 is
 * ssad_conv3x3_c64. */
int ssad_conv3x3_c64_op(const float* in, const float* w_ohwi, float* out, const float* residual, const uint8_t* res_mask,
                        const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit, int64_t N,
                        int H, int W, double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                        float* running_var, int op, void* stream);
/* Inference form of the halo-tile convolution (frozen BatchNorm folded into scale / shift):
 * out = act(conv3x3(in) * scale[co] + shift[co] + residual), 64 -> 64 channels, stride 1, pad 1, exact fp32 MFMA.
 * in_hwnc / out_hwnc / res_hwnc != 0: the input / the output / the residual is position-major [H][W][N][64] (the
 * patch-scoring trunk's layout, see ssad_conv_igemm_fwd_hwnc), otherwise NHWC.  Replaces ssad_conv_igemm_fwd(_hwnc) for the four layer1
 * convolutions of PeraNet.forward in eval mode (torchvision BasicBlock, models.py:224 of the reference). */
int ssad_conv3x3_c64_eval(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                          const float* residual, int relu, int64_t N, int H, int W, int in_hwnc, int out_hwnc, int res_hwnc,
                          void* stream);
/* Weight gradient of the 3x3 / stride 1 / pad 1 convolutions (Cin, Cout multiples of 64) as a halo-tile kernel: a workgroup
 * owns a 64 x 64 (co, ci) block for all nine taps and walks over pixel tiles (csrc/wgrad_halo.hip).  Same contract as
 * ssad_conv_wgrad: slab[splits][Cout][9 * Cin] with splits = ssad_wgrad3x3_halo_splits(...), then ssad_wgrad_reduce. */
int ssad_wgrad3x3_halo_ok(int Cin, int Cout, int KH, int KW, int stride, int pad);   /* 0 no, 1 stride-1 form, 2 stride-2 form */
int ssad_wgrad3x3_halo_splits(int64_t N, int H, int W, int Cin, int Cout);            /* H, W: size of dz (the conv's OUTPUT) */
int ssad_conv_wgrad3x3_halo(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                            int64_t dz_elems, void* stream);
/* 3x3 / stride 2 / pad 1 (the first convolution of layer2-4): dz [N][Ho][Wo][Cout], x [N][H][W][Cin], Ho = (H - 1) / 2 + 1. */
int ssad_conv_wgrad3x3s2_halo(const float* dz, const float* x, float* slab, int splits, int64_t N, int Ho, int Wo, int H, int W,
                              int Cin, int Cout, int64_t dz_elems, void* stream);
/* Stem conv 7x7 / 2 of the precision-16 training step (csrc/stem16.hip): fp16 (f16 != 0) or bf16 operands, fp32 accumulation; the
 * 16-bit counterpart of ssad_stem_fwd_stats for whole images -- resnet.conv1 + bn1 statistics under fp16 autocast
 * (pl.Trainer(precision=16), src/self_supervised/tools.py:263; models.py:224).  wk16: 14 * 64 * 16 halves from ssad_pack_stem_weight16;
 * workspace: ssad_stem_stats_rows() * 128 doubles. */
int ssad_pack_stem_weight16(const float* w_oihw, void* wk16, int f16, void* stream);
int ssad_pack_stem_weight16_ohwi(const float* w_ohwi, void* wk16, int f16, void* stream);   /* OHWI source, as ssad_pack_stem_weight_ohwi */
int ssad_stem_fwd_stats16(const float* img, int B, int H, int W, const void* wk16, float* out, float eps, float momentum, float* mean,
                          float* invstd, float* running_mean, float* running_var, double* workspace, int f16, void* stream);
/* The same halo-tile weight gradient with 16-bit OPERANDS (fp32 tensors in memory, rounded to fp16 -- f16 != 0 -- or bf16 while
 * staged, fp32 accumulation; csrc/wgrad_halo16.hip): the conv2d weight gradient of the reference's fp16-autocast training
 * (pl.Trainer(precision=16), src/self_supervised/tools.py:263, :270, :303) for the 3 x 3 / stride 1 / pad 1 layers.  Same slab contract:
 * splits = ssad_wgrad3x3_halo16_splits(...), then ssad_wgrad_reduce. */
int ssad_wgrad3x3_halo16_ok(int Cin, int Cout, int KH, int KW, int stride, int pad);
int ssad_wgrad3x3_halo16_splits(int64_t N, int H, int W, int Cin, int Cout);
int ssad_conv_wgrad3x3_halo16(const float* dz, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                              int f16, int64_t dz_elems, void* stream);
/* fp16-operand forms: the reference trains under fp16 autocast (pl.Trainer(precision=16), src/self_supervised/tools.py:263,
 * :296), i.e. its Conv2d / Linear products take fp16 operands and accumulate in fp32.  Same contract as the _bf16 forms
 * with v_mfma_f32_32x32x16_f16 (11-bit significands); the slab of ssad_conv_wgrad_f16 is sized by ssad_wgrad_splits_bf16. */
int ssad_conv_igemm_fwd_f16(const float* in, const float* w_ohwi, float* out, const float* scale, const float* shift,
                            const float* residual, int relu, int64_t N, int H, int W, int Cin, int Cout, int KH, int KW,
                            int stride, int pad, void* stream);
int ssad_conv_igemm_dgrad_f16(const float* dy, const float* w_flipT, float* dx, const float* residual, int64_t N, int Hy,
                              int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
int ssad_conv_wgrad_f16(const float* dy, const float* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                        int KH, int KW, int stride, int pad, int64_t dy_elems, void* stream);
/* Stem in training: im2col rows of 160 floats (147 taps + zero pad, nearest resize fused) so that conv1 forward and
 * its weight gradient run on the generic MFMA kernels.  Replaces conv1 of torchvision resnet18 under autograd. */
int ssad_stem_im2col(const float* img, float* col, int64_t B, int H, int W, int Hv, int Wv, void* stream);
/* Weight gradient of resnet conv1 (7x7/2, 3 -> 64; autograd node of models.py:224 inside trainer.fit) straight from the
 * NCHW image -- no im2col buffer.  dz NHWC [B][Ho][Wo][64] with Ho, Wo as ssad_stem_fwd computes them (images below
 * 64x64 go through the same nearest resize).  dw receives 64*7*7*3 floats, OHWI or (to_oihw) OIHW, optionally
 * accumulated.  workspace: ssad_stem_wgrad_workspace(B, H, W) floats (per-workgroup slabs, summed in a fixed order). */
int64_t ssad_stem_wgrad_workspace(int B, int H, int W);
int ssad_stem_wgrad(const float* img, const float* dz, float* dw, int B, int H, int W, int64_t dz_elems, int to_oihw,
                    int accumulate, float* workspace, void* stream);
int ssad_pack_stem_weight_2d(const float* w_oihw, float* out, void* stream);
/* Replaces nn.BatchNorm2d / nn.BatchNorm1d in training mode (models.py:65-95 + torchvision BasicBlock):
 * batch statistics over R rows (biased variance for normalisation, unbiased for running_var, momentum),
 * y = (z-mean)*invstd*gamma+beta (+residual)(ReLU), and the matching backward.  workspace: doubles, see
 * ssad_colreduce_workspace. */
int64_t ssad_colreduce_workspace(int64_t R, int C);
int ssad_bn_stats(const float* z, int64_t R, int C, float eps, float momentum, float* mean, float* invstd,
                  float* running_mean, float* running_var, double* workspace, void* stream);
/* Conv2d (bias-free, as in every BasicBlock: torchvision resnet.py conv3x3/conv1x1) + the batch statistics of the
 * train-mode BatchNorm2d that follows it (models.py:224-243 under trainer.fit) in one pass: the conv kernel leaves
 * per-workgroup double-precision column sums in `workspace` (ssad_conv_stats_workspace(N, Ho, Wo, Cout) doubles), a
 * finalize launch produces mean / invstd and updates the running statistics exactly as ssad_bn_stats does.
 * `out` receives the raw convolution z (NHWC).  bf16: operand mode, 0 fp32, 1 bf16, 2 fp16, 3 / 6 split-bf16. */
int64_t ssad_conv_stats_workspace(int64_t N, int Ho, int Wo, int Cout);
int ssad_conv_igemm_fwd_stats(const float* in, const float* w_ohwi, float* out, int64_t N, int H, int W, int Cin, int Cout,
                              int KH, int KW, int stride, int pad, int bf16, float eps, float momentum, float* mean,
                              float* invstd, float* running_mean, float* running_var, double* workspace, void* stream);
int ssad_bn_apply_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                      const float* residual, float* y, int64_t R, int C, int relu, void* stream);
int ssad_bn_bwd_reduce(const float* dy, const float* yact, const float* z, const float* mean, const float* invstd,
                       float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream);
int ssad_bn_apply_bwd(const float* dy, const float* yact, const float* z, const float* mean, const float* invstd,
                      const float* gamma, const float* dbeta, const float* dgamma, float* dz, float* dres, int64_t R, int C,
                      int eval_mode, void* stream);
/* Residual blocks without the saved-activation reads: ssad_bn_apply_fwd_mask also writes the final ReLU's active set as a
 * nibble mask (one byte per channel quad, bit k = channel 4q+k positive: 1/16 of the activation's bytes); the backward
 * reductions / apply take g = dy * mask from it (mask4 NULL: no ReLU), and the gradient of the identity branch -- dy under the
 * same mask -- is applied by the consumer (ssad_conv_igemm_dgrad_masked, ssad_conv3x3_c64 res_mask) instead of being stored.
 * Replaces the same autograd nodes as ssad_bn_apply_fwd / _bwd_reduce / _apply_bwd (torchvision BasicBlock bn2 + add + relu). */
int ssad_bn_apply_fwd_mask(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                           const float* residual, float* y, uint8_t* mask4, int64_t R, int C, int relu, void* stream);
int ssad_bn_bwd_reduce_mask(const float* dy, const uint8_t* mask4, const float* z, const float* mean, const float* invstd,
                            float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream);
int ssad_bn_apply_bwd_mask(const float* dy, const uint8_t* mask4, const float* z, const float* mean, const float* invstd,
                           const float* gamma, const float* dbeta, const float* dgamma, float* dz, int64_t R, int C, void* stream);
/* The same three over half tensors (the precision-16 step, round 6): a lane's 8 channels take two mask bytes; the mask is that of the
 * STORED half activation.  Same autograd nodes under pl.Trainer(precision=16) (tools.py:263). */
int ssad_bn_apply_fwd_mask_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             const void* residual, void* y, uint8_t* mask4, int64_t R, int C, int relu, void* stream);
int ssad_bn_bwd_reduce_mask_h(const void* dy, const uint8_t* mask4, const void* z, const float* mean, const float* invstd,
                              float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream);
int ssad_bn_apply_bwd_mask_h(const void* dy, const uint8_t* mask4, const void* z, const float* mean, const float* invstd,
                             const float* gamma, const float* dbeta, const float* dgamma, void* dz, int64_t R, int C, void* stream);
int ssad_conv_igemm_dgrad_masked(const float* dy, const float* w_flipT, float* dx, const float* residual, const uint8_t* res_mask,
                                 int64_t N, int Hy, int Wy, int Cout, int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad,
                                 void* stream);
/* The same two calls for a BatchNorm + ReLU with NO residual in between (conv1 of a BasicBlock, the stem, the head's
 * Linear+BN+ReLU): the ReLU mask is recomputed as (z - mean) * invstd * gamma + beta > 0 -- the expression the forward
 * evaluated -- so the saved activation is not read again. */
int ssad_bn_bwd_reduce_zmask(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                             const float* beta, float* dbeta, float* dgamma, int64_t R, int C, double* workspace,
                             void* stream);
int ssad_bn_apply_bwd_zmask(const float* dy, const float* z, const float* mean, const float* invstd, const float* gamma,
                            const float* beta, const float* dbeta, const float* dgamma, float* dz, int64_t R, int C,
                            void* stream);
/* Replaces the backward of nn.MaxPool2d(3,2,1) and of adaptive_avg_pool2d + cat (models.py:224-245). */
int ssad_maxpool3x3s2_bwd(const float* x, const float* dy, float* dx, int64_t N, int H, int W, int C, int64_t dy_elems, void* stream);
/* Same pair with the argmax recorded by the forward pass (one byte per output element: window slot dy*3+dx of the
 * first maximum), so backward compares <= 4 indices per input element instead of rescanning 4 windows. */
int ssad_maxpool3x3s2_fwd_idx(const float* in, float* out, uint8_t* idx, int64_t N, int H, int W, int C, void* stream);
/* Stem tail / head of a TRAINING step, fused around the max-pool (models.py:224 under trainer.fit: bn1, relu, maxpool and
 * their autograd nodes).  Forward: BatchNorm (batch statistics mean / invstd already taken) + ReLU + max-pool 3x3/2 over
 * the raw conv1 output z [N][H][W][C]; only the pooled map and the argmax slots are written.  Backward: from the pooled
 * gradient dpool [N][Ho][Wo][C], the slots and z: dbeta, dgamma and dz (gradient of z) -- the max-pool backward, the
 * ReLU mask (recomputed from z) and both BatchNorm passes; the 128x128 activation and its gradient never exist in HBM.
 * workspace: ssad_colreduce_workspace(N*H*W, C) doubles. */
int ssad_bn_relu_maxpool_fwd(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                             float* out, uint8_t* idx, int64_t N, int H, int W, int C, void* stream);
/* Round 5: the stem's BatchNorm backward reduction over the POOLED tensors.  ..._fwd_win also writes zwin[N][Ho][Wo][C], the raw z of each
 * window's winner; sum over pixels of g and g * xhat = sum over windows of dpool * mask(zwin) and dpool * mask * xhat(zwin) (a window
 * routes its gradient to exactly one pixel), i.e. ssad_bn_bwd_reduce_zmask(dpool, zwin, ...) over N * Ho * Wo rows; ..._bwd_apply is the
 * apply pass alone with dbeta / dgamma as inputs.  Half forms: ssad_bn_relu_maxpool_fwd_win_h, ssad_pool_bn_relu_bwd_apply_h. */
int ssad_bn_relu_maxpool_fwd_win(const float* z, const float* mean, const float* invstd, const float* gamma, const float* beta, float* out,
                                 uint8_t* idx, float* zwin, int64_t N, int H, int W, int C, void* stream);
int ssad_bn_relu_maxpool_fwd_win_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* out,
                                   uint8_t* idx, void* zwin, int64_t N, int H, int W, int C, void* stream);
int ssad_pool_bn_relu_bwd_apply(const uint8_t* idx, const float* dpool, const float* z, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, const float* dbeta, const float* dgamma, float* dz, int64_t N, int H,
                                int W, int C, int64_t dpool_elems, void* stream);
int ssad_pool_bn_relu_bwd_apply_h(const uint8_t* idx, const void* dpool, const void* z, const float* mean, const float* invstd,
                                  const float* gamma, const float* beta, const float* dbeta, const float* dgamma, void* dz, int64_t N, int H,
                                  int W, int C, int64_t dpool_elems, void* stream);
int ssad_pool_bn_relu_bwd(const uint8_t* idx, const float* dpool, const float* z, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float* dbeta, float* dgamma, float* dz, int64_t N, int H,
                          int W, int C, int64_t dpool_elems, double* workspace, void* stream);
int ssad_maxpool3x3s2_bwd_idx(const uint8_t* idx, const float* dy, float* dx, int64_t N, int H, int W, int C, int64_t dy_elems,
                              void* stream);
int ssad_gap_bwd(const float* dpooled, float* dy, int64_t N, int HW, int C, int stride, int offset, int accumulate,
                 void* stream);
/* Replaces F.cross_entropy + torchmetrics accuracy (models.py:261-262) and their backward: loss_acc[0] = mean NLL,
 * loss_acc[1] = accuracy, dlogits[b][0..ldd) = (softmax - onehot) * grad_scale (zero padded to ldd columns). */
int ssad_softmax_ce(const float* logits, const int64_t* labels, int B, int C, float* loss_acc, float* dlogits, int ldd,
                    float grad_scale, void* stream);
/* Replaces torch.optim.SGD(lr, momentum=0.9, weight_decay=5e-4).step() (models.py:337) over a flat parameter arena:
 * m = momentum*m + (g*grad_scale + weight_decay*p); p -= lr*m. */
int ssad_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay,
                  float grad_scale, void* stream);
/* The same update with its hyper-parameters in device memory -- hyper = [lr, momentum, weight_decay, grad_scale] -- so a
 * captured step (hipGraph) never bakes a learning rate in; `scaler` = NULL or the loss-scaler state below. */
int ssad_sgd_step_dev(float* p, const float* g, float* m, int64_t n, const float* hyper, const float* scaler, void* stream);
/* Dynamic loss scaling for the fp16-operand path: what torch.cuda.amp.GradScaler does for the reference under
 * pl.Trainer(precision=16) (tools.py:263).  scaler = 3 device floats [loss_scale, growth_tracker, found_inf]:
 * ssad_scale_by_loss_scale multiplies the loss gradient by loss_scale, ssad_check_finite sets found_inf when a gradient is
 * inf / nan, ssad_sgd_step_dev unscales (and skips the update while found_inf is set), ssad_loss_scaler_update applies
 * backoff / growth and clears found_inf. */
int ssad_scale_by_loss_scale(float* x, int64_t n, const float* scaler, void* stream);
int ssad_check_finite(const float* g, int64_t n, float* scaler, void* stream);
int ssad_loss_scaler_update(float* scaler, float growth_factor, float backoff_factor, int growth_interval, void* stream);

/* ---- synthetic-defect augmentation (batched, GPU resident) ---- */
/* One record per sample, drawn on the host in the reference's order (datasets.py:209-394): the GPU does the pixels,
 * with Pillow's own integer / float32 rules (self_supervised/pil_exact.py) so that the result is byte-identical to the
 * reference's PIL path.  All fields are 32-bit; the Python side mirrors this layout with a numpy structured dtype
 * (augment.py) and checks ssad_aug_params_size(). */
#define SSAD_AUG_MAX_LINE_POINTS 32
typedef struct ssad_aug_params {
    int32_t label;                          /* 0 good, 1 polygon patch, 2 scars, 3 poly-line */
    int32_t crop_left, crop_top;            /* output window origin inside the (affine) source image */
    int32_t aff_on;                         /* 0: no RandomAffine; 1: Image.transform(AFFINE, NEAREST) with the coefficients below */
    int32_t aff_fix[6];                     /* Pillow affine_fixed 16.16 integers: sx = (a2 + y a1 + x a0) >> 16, sy = (a5 + y a4 + x a3) >> 16 */
    int32_t cut_index, cut_left, cut_top, cut_w, cut_h;   /* defect source: index into `cuts` (-1 = the sample's own image) + window */
    int32_t patch_src_left, patch_src_top, patch_w, patch_h, patch_dst_left, patch_dst_top, patch_flat;
    int32_t patch_rgb[3];
    float   patch_bright[2];                /* two successive ImageEnhance.Brightness factors ... */
    int32_t patch_nbright;                  /* ... applied when this is 2 (0: none) */
    int32_t poly_n;
    int32_t poly_xy[16];                    /* up to 8 integer vertices in patch coordinates (ImageDraw.polygon) */
    int32_t scar_src_left, scar_src_top, scar_w, scar_h, scar_flat;
    int32_t scar_rgb[3];
    float   scar_bright[2];
    int32_t scar_nbright;
    int32_t scar_rot;                       /* 0: rotate() took the copy path (angle % 360 == 0); 1: affine_fixed coefficients below */
    int32_t scar_fix[6];
    int32_t scar_rw, scar_rh, scar_n;       /* rotated (expanded) size, number of pasted copies (<= 5) */
    int32_t scar_dst[10];                   /* left, top per copy */
    int32_t line_n;                         /* poly-line points (<= SSAD_AUG_MAX_LINE_POINTS), already (int)-truncated */
    int32_t line_xy[2 * SSAD_AUG_MAX_LINE_POINTS];
    int32_t line_quad[8 * (SSAD_AUG_MAX_LINE_POINTS - 1)];   /* width > 1: ImagingDrawWideLine's four vertices per segment */
    int32_t line_quad_ok[SSAD_AUG_MAX_LINE_POINTS - 1];      /* 0: degenerate segment (a single point) */
    int32_t line_rgb[3];
    int32_t line_width;
    int32_t jit_n;                          /* enabled ColorJitter ops, in application order: */
    int32_t jit_order[3];                   /* 0 brightness, 1 contrast, 2 saturation */
    float   jit_factor[3];                  /* indexed by op */
} ssad_aug_params;

int ssad_aug_params_size(void);
/* Replaces the PIL pixel work of PretextTaskDataset.__getitem__ (src/self_supervised/datasets.py:209-394;
 * dataset_generator.py:42-101, :268-275) for a whole batch: imgs [B][H][W][3] uint8 (and cuts [NC][H][W][3]) ->
 * out [B][3][h][w] fp32 normalised with mean3/std3 (host pointers) and, in `work` (B*h*w*3 bytes), the uint8 HWC image
 * BEFORE ColorJitter.  gray_mean: B floats of scratch. */
int ssad_cutpaste_augment(const uint8_t* imgs, const uint8_t* cuts, const ssad_aug_params* params, uint8_t* work,
                          float* gray_mean, float* out, int B, int H, int W, int h, int w, const float* mean3_host,
                          const float* std3_host, void* stream);
/* AnomalyDetector.predict (models.py:363-370) in one kernel: out[n] = mean of the k (1..3) smallest clip(1 - cos(x_n, bank_r), 0, 2)
 * over the R rows of an L2-normalised bank; the N x R similarity matrix is never written.  Bit-identical to
 * ssad_l2_normalize_rows -> ssad_conv_igemm_fwd (1x1) -> ssad_cosine_knn_mean.  D % 32 == 0. */
int ssad_cosine_knn_fused(const float* x, const float* bank_normalized, float* out, int64_t N, int D, int R, int k, void* stream);
/* HOST function (no GPU work): per-channel integer sums of the window [top, top + h) x [left, left + w) of the NEAREST affine
 * transform of an H x W x 3 uint8 image (fix: the six 16.16 coefficients of ssad_aug_params.aff_fix; NULL: the image itself),
 * zero outside the image.  The sampler's colour-similarity test (datasets.py:300-312) needs this mean between two random
 * draws. */
int ssad_affine_window_sum_u8(const uint8_t* img, int H, int W, const int32_t* fix, int left, int top, int w, int h,
                              int64_t* sum3);
/* The other sort-bound metrics of an evaluation, for maps that are already on the device (same hand-written radix sort as
 * ssad_auroc; no library sort).
 * ssad_pro_curve: the MVTec per-region-overlap curve of metrics.compute_pro (src/self_supervised/metrics.py:58-190): scores fp32
 *   [n]; fp_w uint8 [n] = 1 on defect-free pixels; pro_w fp64 [n] = 1 / (size of the pixel's ground-truth region), 0 elsewhere;
 *   n_ok = number of defect-free pixels, n_regions = number of regions (each at least 1, numpy's max(., 1)).  Writes one curve
 *   point per distinct score in descending order -- fprs fp32, pros fp64, both clipped at 1 -- and their number to count[0]
 *   (device memory; the arrays need room for n entries).  The caller adds the end points (0, 0) and (1, 1).
 * ssad_best_f1_threshold: the threshold maximising F1 over torchmetrics' precision-recall curve (tools.py:141-146), float32
 *   arithmetic as there, the smallest threshold among equal F1.  targets: 1 = positive.  out[0] = threshold, out[1] = F1.
 * ssad_confusion_counts: tp, fp, fn, tn of (scores >= threshold) against (targets != 0), added into out[0..3] (int64, zeroed by
 *   the caller): what compute_f1 / compute_iou (tools.py:131-139) need. */
int64_t ssad_pro_curve_workspace(int64_t n);
int ssad_pro_curve(const float* scores, const uint8_t* fp_w, const double* pro_w, int64_t n, double n_ok, double n_regions,
                   void* workspace, int64_t workspace_bytes, float* fprs, double* pros, int64_t* count, void* stream);
int64_t ssad_best_f1_workspace(int64_t n);
int ssad_best_f1_threshold(const float* scores, const uint8_t* targets, int64_t n, void* workspace, int64_t workspace_bytes, float* out,
                           void* stream);
int ssad_confusion_counts(const float* scores, const uint8_t* targets, int64_t n, float threshold, int64_t* out, void* stream);

/* transforms.ToTensor() on a uint8 HWC batch: -> [B][3][H][W] fp32 in [0,1] (the Dataset's third output). */
int ssad_u8hwc_to_f32chw(const uint8_t* img, float* out, int B, int H, int W, void* stream);
/* The transform of MVTecDataset.__getitem__ (datasets.py:68-80, :102-105: ToTensor, then Normalize(mean, std)) on a uint8 HWC
 * batch that is already on the device: orig (may be NULL) = img / 255, norm = (img / 255 - mean) / std, both [B][3][H][W], each
 * operation a single IEEE fp32 operation as torch evaluates them on the host (bit-identical).  mean3 / std3: host pointers. */
int ssad_u8hwc_to_f32chw_norm(const uint8_t* img, float* orig, float* norm, int B, int H, int W, const float* mean3_host,
                              const float* std3_host, void* stream);

/* Image.resize(size) of Pillow (default filter: BICUBIC) for 8-bit 'L' (C = 1) / 'RGB' (C = 3) batches already on the device:
 * what the reference applies to every file it opens (datasets.py:68, :211-213, :189-200).  in [B][Hin][Win][C] -> out
 * [B][Hout][Wout][C]; tmp [B][Hin][Wout][C] when both extents change (else may be NULL).  bounds_* int32 [out][2] = (first source
 * index, taps) and coef_* int32 [out][ksize_*] (22-bit fixed point) are DEVICE arrays holding libImaging's precompute_coeffs +
 * normalize_coeffs_8bpc tables (self_supervised/pil_exact.resample_coeffs); the axis whose extent does not change takes no table.
 * Bit-exact integer arithmetic (Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc). */
int ssad_resize_bicubic_u8(const uint8_t* in, uint8_t* tmp, uint8_t* out, int B, int Hin, int Win, int C, int Hout, int Wout,
                           const int32_t* bounds_x, const int32_t* coef_x, int ksize_x, const int32_t* bounds_y,
                           const int32_t* coef_y, int ksize_y, void* stream);

/* dataset_generator.obj_mask (dataset_generator.py:27-39: skimage.feature.canny(sigma = 1.5, low 5, high 15) -> binary dilation 3 x 3
 * -> closing 3 x 3 -> fill holes -> erosion 4 x 4 -> largest 8-connected component) for a uint8 RGB batch on the device:
 * rgb [B][H][W][3] -> mask [B][H][W] (0 / 1) and edges [B][H][W] (the Canny edge map, 0 / 1).  gauss_w_host: the 2 * radius + 1
 * normalised Gaussian weights as scipy.ndimage computes them (host memory); low / high: the thresholds on the [0, 1] scale.
 * workspace: ssad_obj_mask_workspace(B, H, W) bytes of device memory.  Bit-exact against the host statement (fp64, scipy's
 * operation order, glibc's hypot restated; fixed points for hysteresis / filling / labelling). */
int64_t ssad_obj_mask_workspace(int B, int H, int W);
int ssad_obj_mask(const uint8_t* rgb, uint8_t* mask, uint8_t* edges, int B, int H, int W, const double* gauss_w_host, int radius,
                  double low, double high, void* workspace, void* stream);

/* ---- half-tensor forms of the precision-16 training step ----
 * pl.Trainer(precision=16) (src/self_supervised/tools.py:263, :296) runs the reference's training_step (models.py:256-277) under
 * torch.autocast(float16): every conv / linear / BatchNorm output of the trunk is an fp16 tensor in memory and so is its gradient.
 * The `_h` entry points below are the SAME kernels as their fp32-tensor namesakes with the activation tensors (void*: N H W C
 * halves) read and written as halves; arithmetic, BatchNorm statistics (those of the stored halves), parameters and parameter
 * gradients stay fp32.  Same call sites, same argument meaning. */
/* one rounded copy of the fp32 master arena per step (autocast's weight cast): src, dst 16-byte aligned */
int ssad_cvt_f32_f16(const float* src, void* dst, int64_t n, void* stream);
/* ssad_flip_transpose_batch writing halves (dgrad operands) */
int ssad_flip_transpose_batch_h(const float* src, void* dst, const int64_t* desc, int n, void* stream);
/* resnet.conv1 with fp16 operands, z stored as halves [B][Ho][Wo][64] (ssad_stem_fwd_stats16) */
int ssad_stem_fwd_stats16_h(const float* img, int B, int H, int W, const void* wk16, void* out, float eps, float momentum,
                            float* mean, float* invstd, float* running_mean, float* running_var, double* workspace, void* stream);
int ssad_bn_relu_maxpool_fwd_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta, void* out,
                               uint8_t* idx, int64_t N, int H, int W, int C, void* stream);
/* conv + train-mode BatchNorm statistics, dgrad: in / out / residual AND the weights (w_ohwi, w_flipT) are halves
 * (ssad_conv_igemm_fwd_stats with bf16 = 2, ssad_conv_igemm_dgrad_f16) */
int ssad_conv_igemm_fwd_stats_h(const void* in, const void* w_ohwi, void* out, int64_t N, int H, int W, int Cin, int Cout, int KH,
                                int KW, int stride, int pad, float eps, float momentum, float* mean, float* invstd,
                                float* running_mean, float* running_var, double* workspace, void* stream);
int ssad_conv_igemm_dgrad_h(const void* dy, const void* w_flipT, void* dx, const void* residual, int64_t N, int Hy, int Wy, int Cout,
                            int Hx, int Wx, int Cin, int KH, int KW, int stride, int pad, void* stream);
/* ssad_conv3x3_c64_op with op = 2 and in / out / residual / emit as halves (the weights are the fp32 OHWI master copy) */
int ssad_conv3x3_c64_h(const void* in, const float* w_ohwi, void* out, const void* residual, const float* tr_mean,
                       const float* tr_invstd, const float* tr_gamma, const float* tr_beta, void* emit, int64_t N, int H, int W,
                       double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                       float* running_var, void* stream);
int ssad_bn_stats_h(const void* z, int64_t R, int C, float eps, float momentum, float* mean, float* invstd, float* running_mean,
                    float* running_var, double* workspace, void* stream);
int ssad_bn_apply_fwd_h(const void* z, const float* mean, const float* invstd, const float* gamma, const float* beta,
                        const void* residual, void* y, int64_t R, int C, int relu, void* stream);
int ssad_gap_fwd_h(const void* in, float* out, int64_t N, int HW, int C, int out_stride, int out_offset, void* stream);
int ssad_gap_bwd_h(const float* dpooled, void* dy, int64_t N, int HW, int C, int stride, int offset, int accumulate, void* stream);
int ssad_bn_bwd_reduce_h(const void* dy, const void* yact, const void* z, const float* mean, const float* invstd, float* dbeta,
                         float* dgamma, int64_t R, int C, double* workspace, void* stream);
int ssad_bn_bwd_reduce_zmask_h(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                               const float* beta, float* dbeta, float* dgamma, int64_t R, int C, double* workspace, void* stream);
int ssad_bn_apply_bwd_h(const void* dy, const void* yact, const void* z, const float* mean, const float* invstd, const float* gamma,
                        const float* dbeta, const float* dgamma, void* dz, void* dres, int64_t R, int C, int eval_mode, void* stream);
int ssad_bn_apply_bwd_zmask_h(const void* dy, const void* z, const float* mean, const float* invstd, const float* gamma,
                              const float* beta, const float* dbeta, const float* dgamma, void* dz, int64_t R, int C, void* stream);
int ssad_pool_bn_relu_bwd_h(const uint8_t* idx, const void* dpool, const void* z, const float* mean, const float* invstd,
                            const float* gamma, const float* beta, float* dbeta, float* dgamma, void* dz, int64_t N, int H, int W,
                            int C, int64_t dpool_elems, double* workspace, void* stream);
/* 3x3 / stride 1 / pad 1 convolution over half tensors (csrc/conv16.hip): the BasicBlock conv3x3 layers of the trunk (models.py:224
 * under pl.Trainer(precision=16)) forward and -- with the flipped filter -- their input gradients.  Halo-tile form with persistent
 * workgroups: the input tile of 64 channels is staged once for its nine taps, optionally through relu(bn(x)) of the producing layer
 * (tr_*: that layer's batch statistics and affine parameters; `emit` then receives the transformed activation for this layer's weight
 * gradient), weight slices two steps ahead, accumulators stored straight from registers.  w_ohwi: halves [Cout][3][3][Cin].
 * stats_ws: ssad_conv3x3_h_stats_rows(...) * 2 * Cout doubles -> mean / invstd / running statistics of the stored output. */
int ssad_conv3x3_h_ok(int Cin, int Cout);
int64_t ssad_conv3x3_h_stats_rows(int64_t N, int H, int W, int Cout);
int ssad_conv3x3_h(const void* in, const void* w_ohwi, void* out, const void* residual, const float* tr_mean, const float* tr_invstd,
                   const float* tr_gamma, const float* tr_beta, void* emit, int64_t N, int H, int W, int Cin, int Cout, double* stats_ws,
                   float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var, void* stream);
/* The same convolution with the filter fed to the matrix cores from registers (csrc/conv16w.hip), for launches that fill the chip
 * (ssad_conv3x3_hw_ok: maps of 16 x 16 blocks or 8 x 8 maps, >= ~200 (tile, channel slab) pairs): a wave owns 128 pixels x 64 channels,
 * the filter is packed once per step in fragment order [Cout/32][tap][Cin/16][2][32][8] halves (ssad_conv3x3_hw_pack_batch, from the
 * fp32 master weights: desc[5 k ..] = source offset in floats, destination offset in halves, Cout, Cin of the conv that runs on it,
 * flip = 1 when the source is the OHWI filter [Cin][3][3][Cout] of the forward conv whose input gradient this is), the halo is staged
 * by four extra waves of the workgroup.  Arguments of ssad_conv3x3_hw as ssad_conv3x3_h, w_packed in place of w_ohwi, plus res_mask
 * (optional, with a residual; round 6): the residual is the identity-branch gradient (dy, nibble mask) of ssad_bn_apply_fwd_mask_h --
 * two mask bytes per 8-half piece -- applied while the residual is staged. */
int ssad_conv3x3_hw_ok(int64_t N, int H, int W, int Cin, int Cout);
int64_t ssad_conv3x3_hw_packed_size(int Cin, int Cout);
int64_t ssad_conv3x3_hw_stats_rows(int64_t N, int H, int W, int Cout);
int ssad_conv3x3_hw_pack_batch(const float* src, void* dst, const int64_t* desc, int n, void* stream);
int ssad_conv3x3_hw(const void* in, const void* w_packed, void* out, const void* residual, const uint8_t* res_mask, const float* tr_mean,
                    const float* tr_invstd, const float* tr_gamma, const float* tr_beta, void* emit, int64_t N, int H, int W, int Cin, int Cout,
                    double* stats_ws, float eps, float momentum, float* mean, float* invstd, float* running_mean, float* running_var,
                    void* stream);
/* The exact-fp32 instantiation of the same kernel (csrc/conv16w.hip, T = float: v_mfma_f32_32x32x2_f32; statistics: packed fp32 sums of 32 values per tile and column, added in
 * double across tiles):
 * the 3 x 3 / stride 1 convs of the fp32 training step (models.py:224) forward and -- with the flipped pack -- their input gradients,
 * whenever the launch fills the chip (ssad_conv3x3_fw_ok; other launches stay on ssad_conv3x3_c64 / ssad_conv_igemm_*).  The filter pack
 * holds floats in fragment order [Cout/32][tap][Cin/8][2][32][4]; res_mask: the residual is the identity-branch gradient (dy, nibble mask)
 * of ssad_bn_apply_fwd_mask, applied while the residual is staged. */
int ssad_conv3x3_fw_ok(int64_t N, int H, int W, int Cin, int Cout);
int ssad_conv3x3_fw_pack_batch(const float* src, float* dst, const int64_t* desc, int n, void* stream);
int ssad_conv3x3_fw(const float* in, const float* w_packed, float* out, const float* residual, const uint8_t* res_mask,
                    const float* tr_mean, const float* tr_invstd, const float* tr_gamma, const float* tr_beta, float* emit, int64_t N,
                    int H, int W, int Cin, int Cout, double* stats_ws, float eps, float momentum, float* mean, float* invstd,
                    float* running_mean, float* running_var, void* stream);
/* Inference form of ssad_conv3x3_fw (the layer1 convs of the patch-scoring pass, models.py:224 in eval mode, replacing ssad_conv3x3_c64_eval
 * on launches that fill the chip): out = act(conv(in) + shift (+ residual)); the folded BatchNorm's SCALE goes into the packed filter
 * (ssad_conv3x3_fw_pack_scaled: w[o][..] * scale[o], fragment order), its shift is added in the epilogue, the residual (NHWC) by the matrix
 * cores.  16 x 16 maps of 64 channels run two maps per tile.  out_hwnc: output written position-major [H][W][N][C]. */
int ssad_conv3x3_fw_eval_ok(int64_t N, int H, int W, int Cin, int Cout);
int ssad_conv3x3_fw_pack_scaled(const float* w_ohwi, const float* scale, float* dst, int Cout, int Cin, void* stream);
int ssad_conv3x3_fw_eval(const float* in, const float* w_packed, float* out, const float* shift, const float* residual, int relu, int64_t N,
                         int H, int W, int Cin, int Cout, int out_hwnc, void* stream);
/* Weight gradient of the 3 x 3 / pad 1 convolutions, stride 1 AND 2, over half tensors (csrc/wgrad16.hip): tiles go to LDS as they lie
 * in memory and the [pixel][channel] -> [channel][pixel] transpose the matrix instruction needs happens in the fragment reads (eight
 * 2-byte LDS reads per operand), so no staging waves, no conversion; same slab contract as ssad_conv_wgrad3x3_halo16:
 * splits = ssad_wgrad3x3_g16_splits(N, Ho, Wo, Cin, Cout, stride), then ssad_wgrad_reduce(slab, dw, splits, Cout, 9 * Cin, 3, 3, Cin, ..).
 * Replaces the conv2d weight-gradient node of every BasicBlock conv3x3 under pl.Trainer(precision=16) (tools.py:263, :270, :303). */
int ssad_wgrad3x3_g16_ok(int Cin, int Cout, int KH, int KW, int stride, int pad);
int ssad_wgrad3x3_g16_splits(int64_t N, int Ho, int Wo, int Cin, int Cout, int stride);
int ssad_conv_wgrad3x3_g16_h(const void* dz, const void* x, float* slab, int splits, int64_t N, int Ho, int Wo, int H, int W, int Cin,
                             int Cout, int stride, int64_t dz_elems, void* stream);
/* weight gradients from half tensors (fp32 slabs, then ssad_wgrad_reduce, as the fp32-tensor forms) */
int ssad_conv_wgrad3x3_halo16_h(const void* dz, const void* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout,
                                int64_t dz_elems, void* stream);
int ssad_conv_wgrad_f16_h(const void* dy, const void* x, float* slab, int splits, int64_t N, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int64_t dy_elems, void* stream);
int ssad_stem_wgrad_h(const float* img, const void* dz, float* dw, int B, int H, int W, int64_t dz_elems, int to_oihw, int accumulate,
                      float* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SSAD_H */
