/*
 * libp2p_hip -- MI355X (gfx950) implementation of the Patch2Pix matching hot path.
 *
 * C ABI, plain pointers and sizes only.  The reference (GrumpyZhou/patch2pix) has no FFI: its
 * boundary for this path is the Python module API (networks/patch2pix.py, utils/eval/model_helper.py).
 * Each entry point below names the reference function(s) it replaces; the Python side
 * (patch2pix_amd/) binds them with ctypes and re-exposes the reference's own names.
 *
 * Conventions
 *   - every function returns 0 on success or a negative P2P_E* code; p2p_last_error() returns a
 *     thread-local message for the last failure on the calling thread;
 *   - "device pointer" arguments are caller-owned HBM buffers (torch tensors); nothing is retained
 *     across calls except the opaque weight handles;
 *   - all launches are asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - threading: every launch function is re-entrant per stream and may be called from several host threads (distinct
 *     streams, or one stream with the caller's own ordering); the weight handles are immutable after creation
 *     (p2p_regressor_set_mode / p2p_ncn_set_tile / p2p_conv_set_tile mutate a handle and must not race with launches that
 *     use it).  The only process-wide state is a per-device "kernel attributes set" flag per launcher (dynamic-LDS size via
 *     hipFuncSetAttribute, compute-unit count): atomic, set after idempotent work, so two threads meeting on a device's
 *     first launch both do that work and neither sees a half-initialised value;
 *   - dense tensors are contiguous fp32 in the layout PyTorch produces (NCHW without the N);
 *   - match rows are (xA, yA, xB, yB).
 */
#ifndef P2P_HIP_H
#define P2P_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define P2P_OK            0
#define P2P_EINVAL       -1   /* bad argument (null pointer, size not supported ...) */
#define P2P_EHIP         -2   /* a HIP runtime call failed */
#define P2P_EUNSUPPORTED -3   /* valid for the reference but not implemented by this library */
#define P2P_ENOMEM       -4   /* workspace too small / allocation failed */

typedef void *p2p_stream_t;                 /* hipStream_t */
typedef struct p2p_ncn p2p_ncn;             /* NCNet consensus filters, device resident */
typedef struct p2p_regressor p2p_regressor; /* one FeatRegressNet, packed + device resident */

int p2p_version(void);
const char *p2p_last_error(void);

/* ---- weights ------------------------------------------------------------------------------- */

/* NeighConsensus(kernel_sizes=[3,3], channels=[16,1]) -- reference networks/ncn/model.py:124-143,
 * built at networks/patch2pix.py:32.  HOST pointers to the checkpoint tensors in their stored
 * (pre-permuted, networks/ncn/conv4d.py:119-120) layout:
 *   w1 [3,16,1,3,3,3]  b1 [16]  w2 [3,1,16,3,3,3]  b2 [1]                                     */
int p2p_ncn_create(const float *w1, const float *b1, const float *w2, const float *b2, p2p_ncn **out);
void p2p_ncn_destroy(p2p_ncn *ncn);
/* Tests and sweeps: force the work-group tile (ta, tb, tc) of the consensus kernel for launches with this handle (0, 0, 0 =
 * automatic; ta = 0 with tb, tc > 0: only the march length is picked; any other partial triple is P2P_EINVAL).  Results do not depend on the tile: every output
 * cell sums its contributions in one fixed order.                                                                   */
int p2p_ncn_set_tile(p2p_ncn *ncn, int ta, int tb, int tc);

/* FeatRegressNet with the released configuration (conv_kers [3,3], conv_strs [2,1],
 * conv_dims [512,512], fc_dims [512,256], feat_comb 'pre', psize 16, feat_idx [0,1,2,3]) --
 * reference networks/modules.py:56-112.  HOST pointers to the state_dict tensors.           */
typedef struct p2p_bn_params {
    const float *weight, *bias, *running_mean, *running_var;
} p2p_bn_params;

typedef struct p2p_regressor_params {
    const float *conv1_w;      /* conv.0.weight [512,518,3,3] */
    p2p_bn_params bn1;         /* conv.1.*      [512]         */
    const float *conv2_w;      /* conv.2.weight [512,512,3,3] */
    p2p_bn_params bn2;         /* conv.3.*      [512]         */
    const float *fc1_w, *fc1_b; /* fc.0 [512,512],[512]       */
    p2p_bn_params bnf1;        /* fc.1.*        [512]         */
    const float *fc2_w, *fc2_b; /* fc.3 [256,512],[256]       */
    p2p_bn_params bnf2;        /* fc.4.*        [256]         */
    const float *fc3_w, *fc3_b; /* fc.6 [5,256],[5]           */
} p2p_regressor_params;

int p2p_regressor_create(const p2p_regressor_params *params, p2p_regressor **out);
void p2p_regressor_destroy(p2p_regressor *reg);

/* Arithmetic used for the two convolutions of a regressor (everything else is fp32 either way; the
 * reference computes them in fp32, networks/modules.py:76-87):
 *   P2P_REGRESS_FP16X2 fp32-equivalent on the fp16 matrix cores: every fp32 operand, scaled by an exact power of
 *                      two into the normal range of fp16, is the sum of two fp16 numbers to within 2^-24 of its
 *                      magnitude; three v_mfma_f32_32x32x16_f16 per product (the dropped term is <= 2^-24 of it),
 *                      fp32 accumulation; the scales are undone exactly.  As accurate as P2P_REGRESS_F32 against
 *                      an fp64 evaluation at 5.3x its matrix-core ceiling;
 *   P2P_REGRESS_F32    v_mfma_f32_32x32x2_f32, bit-identical to an fp32 fma chain;
 *   P2P_REGRESS_FP16X2W (default) the same arithmetic with the second convolution (3x3, stride 1 on the 8x8 map) evaluated as
 *                      Winograd F(2x2, 3x3): 16 batched GEMMs over the transformed tiles of ALL proposals (2.25x fewer
 *                      matrix-core passes; the transforms are exact up to fp32 rounding, transformed filters computed in
 *                      fp64 at pack time) -- three launches per regressor level instead of one, and a larger scratch
 *                      buffer (the transformed conv2 input of a chunk of up to 3328 proposals, 512 KiB each).
 * New handles start in P2P_REGRESS_DEFAULT (the library reads no environment variables; the Python host layer maps
 * P2P_REGRESS_MODE onto p2p_regressor_set_mode).  Only the weight stream of the mode in use is packed and uploaded;
 * p2p_regressor_set_mode builds another mode's on its first selection (host-side packing + one upload).   */
#define P2P_REGRESS_F32     0
#define P2P_REGRESS_FP16X2  3
#define P2P_REGRESS_FP16X2W 4
#define P2P_REGRESS_DEFAULT P2P_REGRESS_FP16X2W
int p2p_regressor_set_mode(p2p_regressor *reg, int mode);
int p2p_regressor_get_mode(const p2p_regressor *reg);

/* ---- coarse stage ---------------------------------------------------------------------------- */

/* Workspace (bytes) p2p_coarse_forward needs for these sizes (per pair). */
size_t p2p_coarse_workspace_bytes(int channels, int hA, int wA, int hB, int wB, int ksize);

/* Patch2Pix.forward_coarse_match -- reference networks/patch2pix.py:120-136:
 * L2Normalize (modules.py:6) -> FeatCorrelation (modules.py:41-53) -> maxpool4d (modules.py:11-34,
 * only if ksize > 1) -> MutualMatching (ncn/model.py:157-176) -> NeighConsensus (ncn/model.py:145-155,
 * conv4d.py:12-74) -> MutualMatching.
 *   featA [C,hA,wA], featB [C,hB,wB]      device, fp32
 *   corr4d_out [hA/k, wA/k, hB/k, wB/k]   device, fp32
 *   delta_out  same shape, uint8, value s = ((di*k+dj)*k+dk)*k+dl of the first maximum in the
 *              reference's slice order (modules.py:13-18); may be NULL; ignored when ksize == 1
 * ksize must be 1 or 2 (the reference default; other values -> P2P_EUNSUPPORTED).               */
int p2p_coarse_forward(const float *featA, const float *featB, int channels, int hA, int wA, int hB, int wB,
                       int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                       void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* The same for the batch axis of the reference's tensors (feat1/feat2 are [B,C,h,w] in
 * networks/patch2pix.py:120-136): `batch` equally sized pairs, contiguous along the leading axis in all four
 * arrays, one launch per kernel for the whole batch.  The workspace must hold at least one pair
 * (p2p_coarse_workspace_bytes); with batch x that size all pairs are processed together, with less they are
 * processed in as many groups as fit.                                                             */
int p2p_coarse_forward_batch(const float *featA, const float *featB, int batch, int channels, int hA, int wA, int hB,
                             int wB, int ksize, const p2p_ncn *ncn, float *corr4d_out, uint8_t *delta_out,
                             void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* NeighConsensus.forward -- reference networks/ncn/model.py:145-155 (kernel_sizes [3,3], channels [16,1]):
 * y = net(x) + T(net(T(x))), net = Conv4d(1->16) + ReLU + Conv4d(16->1) + ReLU (conv4d.py:12-74), T = swap of the A and B axes,
 * on `batch` volumes x [B, hA, wA, hB, wB] (fp32) -> y_out of the same shape.  One kernel on the fp16 matrix cores in
 * fp32-equivalent arithmetic (the fp16 planes are scaled by the volume's largest magnitude, found first); the hidden
 * 16-channel volume stays in LDS (csrc/consensus.hip).  workspace: 4 bytes per volume of device memory.          */
int p2p_neigh_consensus_batch(const float *x, int batch, int hA, int wA, int hB, int wB, const p2p_ncn *ncn, float *y_out,
                              void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* Expand the packed relocalisation byte into the reference's four int64 tensors
 * (max_i, max_j, max_k, max_l of modules.py:24-28); `out` holds 4 consecutive planes of n int64. */
int p2p_delta_unpack(const uint8_t *delta, size_t n, int ksize, int64_t *out, p2p_stream_t stream);

/* Patch2Pix.cal_coarse_matches -- reference networks/patch2pix.py:340-375, i.e. corr_to_matches in
 * both directions (ncn/extract_ncmatches.py:6-94: softmax along one side, max/first-argmax,
 * relocalisation with delta) concatenated B->A first then A->B, scaled to pixels:
 *   matches_out [nB + nA, 4] int64 = upsample * (xA,yA,xB,yB) (+ upsample/2 if center)
 *   scores_out  [nB + nA]    fp32                                    (nA = hA'*wA', nB = hB'*wB')
 * corr4d dims are the pooled ones; delta may be NULL (ksize 1).                                  */
int p2p_coarse_matches(const float *corr4d, const uint8_t *delta, int hA, int wA, int hB, int wB, int ksize,
                       int upsample, int center, int64_t *matches_out, float *scores_out, p2p_stream_t stream);

/* Batch form: corr4d [B, ...], delta [B, ...], matches_out [B, nB + nA, 4], scores_out [B, nB + nA]. */
int p2p_coarse_matches_batch(const float *corr4d, const uint8_t *delta, int batch, int hA, int wA, int hB, int wB,
                             int ksize, int upsample, int center, int64_t *matches_out, float *scores_out,
                             p2p_stream_t stream);

/* filter_coarse -- reference networks/utils.py:38-72 without the `ptmax` sampling (which draws from the host's numpy
 * RNG and therefore stays on the host): per batch item the lexicographically sorted distinct rows of matches [n,4]
 * with the score of their first occurrence; `mutual` keeps rows that occur more than once; an empty selection leaves
 * the list as it was; then rows with score > ncn_thres, again "or everything".
 *   matches [B,n,4] int64, scores [B,n] fp32  ->  out_matches [B,n,4], out_scores [B,n] (first out_counts[b] rows
 *   valid), out_counts [B] int32 on the device; out_counts[b] = -1 if a coordinate is negative or >= 2^15 (the caller
 *   falls back to the host path).
 * Lists of up to 8192 rows are filtered entirely in LDS and need no workspace (NULL, 0).  Longer lists (a 960x1280 pair
 * at ksize 2 has 9600 rows) are sorted in `workspace`, p2p_filter_coarse_workspace_bytes(batch, n) bytes of device
 * memory (0 for n <= 8192; P2P_ENOMEM if it is missing or too small); n <= 2^20.                                  */
size_t p2p_filter_coarse_workspace_bytes(int batch, int n);
int p2p_filter_coarse_batch(const int64_t *matches, const float *scores, int batch, int n, float ncn_thres, int mutual,
                            int64_t *out_matches, float *out_scores, int *out_counts, void *workspace,
                            size_t workspace_bytes, p2p_stream_t stream);

/* The tail of estimate_matches -- reference utils/eval/model_helper.py:92-109 -- for a batch with device-side counts:
 * per item keep the rows with fine score > io_thres (all rows if none passes), in order, and scale refined and coarse
 * coordinates to original-image pixels in float64.
 *   fine [B,stride,4] fp32, scores [B,stride] fp32, coarse [B,stride,4] int64, counts [B] int32 (device; valid rows per
 *   item, -1 passes through), scale [B,4] float64 (device; w1o/w1, h1o/h1, w2o/w2, h2o/h2)
 *   -> out_matches [B,stride,4] f64, out_scores [B,stride] fp32, out_coarse [B,stride,4] f64, out_counts [B] int32.   */
int p2p_match_tail_batch(const float *fine, const float *scores, const int64_t *coarse, const int *counts,
                         const double *scale, int batch, int stride, float io_thres, double *out_matches,
                         float *out_scores, double *out_coarse, int *out_counts, p2p_stream_t stream);

/* ---- fine stage ------------------------------------------------------------------------------ */

/* One image's feature pyramid levels feat_idx [0,1,2,3] (reference networks/resnet.py:138-157 with
 * change_stride): device pointers [3,H,W], [64,H1,W1], [64,H2,W2], [128,H3,W3] with Hj = ceil(H / 2^j)
 * (the extents the backbone's strided layers produce for ANY H, W -- refine_matches loads images without
 * rounding their size, utils/datasets/preprocess.py:7-30); gathered indices are clamped to H // 2^j - 1 like
 * networks/utils.py:22-23.                                                                     */
typedef struct p2p_pyramid {
    const float *level[4];
    int height, width;      /* of level 0 (the network input); 8 <= H, W < 32768 */
} p2p_pyramid;

/* Patch2Pix.forward_fine_match for one batch item -- reference networks/patch2pix.py:157-218:
 * select_local_patch_feats (networks/utils.py:4-36) -> L2Normalize(dim 0) -> FeatRegressNet
 * (modules.py:101-112) -> parse_regressor_out (patch2pix.py:138-155).
 * With reg2 != NULL the second regressor is run on the first one's output inside the same launch
 * (predict_fine's mid -> fine chain, patch2pix.py:259-272).
 *   proposals   [n,4] int64 (is_float == 0) or fp32 (is_float != 0), device
 *   matches1/probs1 [n,4]/[n] fp32 outputs of reg1 (may be NULL when reg2 != NULL)
 *   matches2/probs2 outputs of reg2 (required when reg2 != NULL)
 *   raw1/raw2   optional [n,5] raw regressor outputs (NULL to skip)
 *   workspace   p2p_regress_workspace_bytes(n) bytes of device memory, 128-byte aligned (scratch of the launch: the pooled
 *               convolution features of every proposal wait there for the batched FC tail; contents are meaningless
 *               outside the call)                                                             */
size_t p2p_regress_workspace_bytes(int n);
/* The same for ONE arithmetic mode (p2p_regress_workspace_bytes is the largest of them = the default mode's):
 * P2P_REGRESS_FP16X2W  4 KB per proposal slot + 512 KiB per proposal of a chunk of at most 3328 (<= 1.74 GB: the value
 *                      jumps from ~2 MB per proposal to that cap once n exceeds one chunk -- callers that run another
 *                      mode should size their buffer with this query, not with p2p_regress_workspace_bytes),
 * P2P_REGRESS_FP16X2   4 KB per proposal slot,   P2P_REGRESS_F32   0 (the buffer is ignored).                      */
size_t p2p_regress_workspace_bytes_mode(int n, int mode);
int p2p_regress(const p2p_regressor *reg1, const p2p_regressor *reg2,
                const p2p_pyramid *im1, const p2p_pyramid *im2,
                const void *proposals, int is_float, int n,
                float *matches1, float *probs1, float *raw1,
                float *matches2, float *probs2, float *raw2,
                void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* The same for `nitems` image pairs in one launch (the reference loops over the batch items of its
 * list arguments, patch2pix.py:192): im1/im2 are arrays of nitems pyramids, counts[i] (HOST array) the
 * number of proposals of item i; proposals and every output are the per-item arrays concatenated in
 * item order.  Filling the chip matters here: one proposal occupies one compute unit.  workspace:
 * p2p_regress_workspace_bytes(sum of counts).                                                   */
int p2p_regress_batch(const p2p_regressor *reg1, const p2p_regressor *reg2, int nitems,
                      const p2p_pyramid *im1, const p2p_pyramid *im2, const int *counts,
                      const void *proposals, int is_float,
                      float *matches1, float *probs1, float *raw1,
                      float *matches2, float *probs2, float *raw2,
                      void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* The same with the proposal counts in DEVICE memory (e.g. written by p2p_filter_coarse_batch): every item owns
 * `stride` slots of the proposal and output arrays ([nitems*stride, ...]), of which the first dev_counts[i] are used;
 * the other slots' outputs are left untouched.  Nothing has to come back to the host between the coarse and the fine
 * stage.  workspace: p2p_regress_workspace_bytes(nitems * stride).                                                 */
int p2p_regress_batch_dev(const p2p_regressor *reg1, const p2p_regressor *reg2, int nitems,
                          const p2p_pyramid *im1, const p2p_pyramid *im2, const int *dev_counts, int stride,
                          const void *proposals, int is_float,
                          float *matches1, float *probs1, float *raw1,
                          float *matches2, float *probs2, float *raw2,
                          void *workspace, size_t workspace_bytes, p2p_stream_t stream);

/* ---- feature-pyramid producer (the convolutions of ResNet34 layer1..layer3) ------------------ */

/* One Conv2d(ci, co, ks, stride, padding = ks / 2, bias = False) + BatchNorm2d(co) in eval mode -- reference
 * networks/resnet.py:26-60 (BasicBlock: conv1/bn1, conv2/bn2, downsample) as used by forward_all (:138-157) with the
 * layer3 stride patch (:169-173).  HOST pointers: weight [co,ci,ks,ks], the four BatchNorm vectors [co].  Supported:
 * ks 1 or 3, stride 1 or 2, ci a multiple of 32, co a multiple of 64 (every convolution of layer1..layer3).
 * Arithmetic: fp32-equivalent (operands as two fp16 planes under exact power-of-two scales, three MFMA products per
 * fp32 product, fp32 accumulation).                                                                             */
typedef struct p2p_conv p2p_conv;
int p2p_conv_create(const float *weight, const p2p_bn_params *bn, int ci, int co, int ks, int stride, p2p_conv **out);
void p2p_conv_destroy(p2p_conv *conv);
/* Experiments / tests: force the work-group tile (mt, nt, wn) of this layer's launches -- [32 mt (4 / wn) pixels] x
 * [32 nt wn channels]; (0,0,0) = chosen from the launch size (default).  A tile the layer does not have is ignored.  The
 * result does not depend on the tile (every tile sums an output's K axis in the same order).                          */
int p2p_conv_set_tile(p2p_conv *conv, int mt, int nt, int wn);

/* y = [relu](bn(conv(x)) [+ residual]) for a batch of n images.  Activations are fp32 **NHWC** device arrays
 * (x [n,h,w,ci], y and residual [n,ho,wo,co], ho = (h + 2 (ks/2) - ks) / stride + 1); xmax [n] holds the float bits of
 * max |x| per image (p2p_absmax_batch, or the ymax of the producing call); ymax (optional, [n] int, **zero on entry**:
 * the kernel raises it with atomicMax) receives the same for y.  The three steps of a BasicBlock: conv1 (relu = 1), downsample (relu = 0, when present), conv2 (residual =
 * the block input or the downsample output, relu = 1).                                                           */
int p2p_conv_forward(const p2p_conv *conv, const float *x, const int *xmax, int n, int h, int w, const float *residual,
                     int relu, float *y, int *ymax, p2p_stream_t stream);

/* The stem: Conv2d(3, 64, 7, stride 2, padding 3, bias = False) + BatchNorm2d(64) + ReLU -- reference
 * networks/resnet.py:101-103 as used at :141-143.  HOST pointers: weight [64,3,7,7] + BatchNorm vectors.  image [n,3,h,w]
 * NCHW fp32 (device), imax [n] float bits of max |image| per item; y [n,64,ho,wo] **NCHW** (pyramid level 1 as the fine
 * stage reads it), ho = (h - 1) / 2 + 1.                                                                         */
typedef struct p2p_stem p2p_stem;
int p2p_stem_create(const float *weight, const p2p_bn_params *bn, p2p_stem **out);
void p2p_stem_destroy(p2p_stem *stem);
int p2p_stem_forward(const p2p_stem *stem, const float *image, const int *imax, int n, int h, int w, float *y, p2p_stream_t stream);

/* MaxPool2d(kernel 3, stride 2, padding 1) -- reference networks/resnet.py:104,146: x [n,c,h,w] NCHW -> y [n,hp,wp,c]
 * **NHWC** (hp = (h - 1) / 2 + 1), the input layout of p2p_conv_forward; ymax (optional, zero on entry) as there.  c a multiple of 64. */
int p2p_maxpool_nhwc(const float *x, int n, int c, int h, int w, float *y, int *ymax, p2p_stream_t stream);

/* [n,h,w,c] -> [n,c,h,w]: a pyramid level in the layout p2p_coarse_forward / p2p_regress read.                     */
int p2p_nhwc_to_nchw(const float *x, int n, int h, int w, int c, float *y, p2p_stream_t stream);

/* out[i] = float bits of max |x| over the count values of item i (items are count values apart).                 */
int p2p_absmax_batch(const float *x, size_t count, int items, int *out, p2p_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* P2P_HIP_H */
