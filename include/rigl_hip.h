/* rigl_hip.h -- C ABI of the MI355X (gfx950) RigL hot path.
 *
 * The reference (google-research/rigl) has no native code and no FFI: its hot
 * path sits behind Python signatures and runs inside TensorFlow ops.  This
 * header is the boundary a maintainer binds INSTEAD of those TF ops; every
 * entry point cites the reference call site it replaces.  Rules:
 *   - plain C, `extern "C"`, no torch / C++ types in any signature;
 *   - the caller owns every buffer and passes raw DEVICE pointers + sizes and
 *     the hipStream_t to enqueue on (as void*); nothing is allocated inside,
 *     scratch comes from a caller-provided workspace (`*_workspace_bytes`);
 *   - every call only ENQUEUES work on `stream` (no hidden synchronisation);
 *   - return 0 on success, a negative RIGL_E* code otherwise; the message for
 *     the calling thread is available from rigl_last_error(); never throws;
 *   - re-entrant: no mutable global state besides the loaded code object.
 *
 * Layout conventions (the reference's):
 *   activations  NHWC, bf16 (raw uint16 bit patterns)
 *   weights      HWIO  [kh][kw][cin][cout] fp32 master copy (FC: [in][out]);
 *                flat index = C order -- this is the index space in which the
 *                masks, the top-k tie-break ("lower index first") and the
 *                golden vectors are defined
 *   mask         1 bit / weight, bit (i & 31) of uint32 word (i >> 5), i = flat
 *                HWIO index; bits at positions >= n in the last word are 0
 */
#ifndef RIGL_HIP_H_
#define RIGL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an entry point is removed or a signature changes (2: round 4 dropped
 * rigl_masked_conv2d_bwd_deferred / rigl_wgrad_reduce_pending and two parameters of rigl_masked_conv2d_bwd_bn). */
#define RIGL_ABI_VERSION 2

typedef void* rigl_stream_t; /* hipStream_t */
typedef uint16_t rigl_bf16;  /* raw bfloat16 bits */

enum {
  RIGL_OK = 0,
  RIGL_EINVAL = -1,       /* bad argument (NULL, negative size, misaligned) */
  RIGL_ELAUNCH = -2,      /* HIP runtime / launch failure */
  RIGL_EWORKSPACE = -3,   /* workspace too small */
  RIGL_EUNSUPPORTED = -4  /* shape / mode not supported by this build */
};

int rigl_version(void);
/* Host utility: CRC-32C (Castagnoli) of a host buffer, continuing from `crc`
 * (0 to start) -- the checksum TensorFlow's checkpoint bundles carry
 * (README.md:34-58 checkpoints; rigl_amd/tf_checkpoint.py).                 */
uint32_t rigl_crc32c(const void* data, size_t n, uint32_t crc);
/* Message of the last failing call made by THIS thread ("" if none). */
const char* rigl_last_error(void);

/* ------------------------------------------------------------------------
 * Mask bitmap <-> 0/1 float mask.
 * Replaces: the fp32 `mask` variable of tf.contrib.model_pruning
 * (registered by layers.masked_conv2d, rigl/imagenet_resnet/pruning_layers.py
 * :139-157) -- the bitmap is the HBM-resident form, the float form exists
 * only at the API surface (get_masks()).
 * ---------------------------------------------------------------------- */
int rigl_mask_pack(const float* mask01, uint32_t* bits, int64_t n,
                   rigl_stream_t stream);
int rigl_mask_unpack(const uint32_t* bits, float* mask01, int64_t n,
                     rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * K2: fused magnitude-prune / gradient-regrow mask update.
 * Replaces: SparseRigLOptimizerBase.generic_mask_update
 *   (rigl/sparse_optimizers_base.py:523-538) + _get_update_op (:276-343)
 *   + get_grow_tensor (:355-400, :540-553) + reset_momentum (:555-564), i.e.
 *   the two full-length nn_ops.top_k sorts, two scatter_nd, where/assign.
 * Also serves SET / Static / Momentum (sparse_optimizers.py:109-214) through
 * the explicit-score pointers.
 *
 * Result (bit-exact with the reference semantics on identical inputs):
 *   n_ones  = popcount(mask);  n_prune = (int32)((float)n_ones * drop_fraction)
 *   n_keep  = n_ones - n_prune
 *   mask1   = the n_keep entries with the largest drop score, equal scores
 *             resolved by LOWER flat index first (tf.nn.top_k order)
 *   lifted  = mask1 ? (min(grow score) - 1) : grow score
 *   mask2   = the n_prune entries with the largest lifted score (same ties)
 *   new     = mask2 & (reinit_when_same ? 1 : ~mask_old)
 *   w[new]  = grow value;  momentum[new] = reset value;  mask = mask1 | mask2
 * drop score = |mask*w| (+ drop_noise[i], one fp32 add) unless score_drop is
 * given; grow score = |dense_grad| unless score_grow is given.
 * ---------------------------------------------------------------------- */
typedef struct RiglPruneRegrowLayer {
  int64_t n;               /* number of weights in this tensor (< 2^31)     */
  float* w;                /* in/out fp32 [n]                                */
  float* momentum;         /* in/out fp32 [n], NULL = no slot                */
  uint32_t* mask_bits;     /* in/out ceil(n/32) words                        */
  const float* dense_grad; /* fp32 [n] dense dL/d(mask*W); may be NULL only  */
                           /* if score_grow is given and modes need no grad  */
  const float* drop_noise; /* fp32 [n] or NULL (noise_std = 0)               */
  const float* score_drop; /* fp32 [n] or NULL: explicit drop scores         */
  const float* score_grow; /* fp32 [n] or NULL: explicit grow scores         */
  const float* grow_values;/* fp32 [n] or NULL: explicit grown-weight values */
} RiglPruneRegrowLayer;

enum {
  RIGL_GROW_ZEROS = 0,      /* grow_init='zeros'          (:372-373)         */
  RIGL_GROW_GRAD_SCALE = 1, /* 'grad_scale_d': g / d      (:542-545)         */
  RIGL_GROW_GRAD_SIGN = 2,  /* 'grad_sign_d': sign(g) / d (:546-549)         */
  RIGL_GROW_EXPLICIT = 3    /* values from grow_values    (random_* / initial_dist_* drawn by the host) */
};
enum {
  RIGL_MOMRESET_ZEROS = 0,  /* SET: zeros                 (:345-353)         */
  RIGL_MOMRESET_GRAD = 1    /* RigL: dense_grad * initial_acc_scale (:555-564) */
};

typedef struct RiglPruneRegrowParams {
  float drop_fraction;
  int32_t grow_init_mode;
  float grow_init_div;
  int32_t momentum_reset_mode;
  float initial_acc_scale;
  int32_t reinit_when_same;
} RiglPruneRegrowParams;

#define RIGL_COUNTS_PER_LAYER 8
/* out_counts[l*8 + ..]: 0 n_ones, 1 n_prune, 2 n_keep, 3 n_new_connections,
 * 4 overlap (!=0 <=> the reference's Assert(mask1*mask2==0) would fire),
 * 5 ties admitted at the drop threshold, 6 ties admitted at the grow
 * threshold, 7 popcount of the new mask.                                   */
size_t rigl_prune_regrow_workspace_bytes(const int64_t* n_per_layer,
                                         int32_t n_layers);
int rigl_prune_regrow(const RiglPruneRegrowLayer* layers /* host */,
                      int32_t n_layers, const RiglPruneRegrowParams* params,
                      int32_t* out_counts /* device, 8*n_layers, nullable */,
                      void* workspace /* device */, size_t workspace_bytes,
                      rigl_stream_t stream);

/* The same update on ONE tensor, with its two selections read back (north
 * star: "masks and top-k indices bit-exact"; SURVEY.md 8(b) sketch
 * `out_topk_idx`): out_mask1_bits = the kept set mask1, out_mask2_bits = the
 * grown set mask2 (bitmaps, either may be NULL) and -- both or neither --
 * out_idx1 / out_idx2 [n] int32: the indices in tf.nn.top_k order of the drop
 * score resp. the lifted grow score (sparse_optimizers_base.py:293, :311),
 * i.e. larger score first, equal scores by lower index; the first n_keep
 * (out_counts[2]) resp. n_prune (out_counts[1]) entries are the selected ones,
 * the entries behind them are the rest of the tensor in the same order.  The
 * tensor is updated exactly as by rigl_prune_regrow.  A test / inspection
 * entry point: one stable 33-bit radix sort of the whole tensor per list.
 * ORDER: the update (weights, mask, momentum) is enqueued FIRST, the two sorts
 * after it; if enqueueing a sort fails the call returns an error with the
 * tensor already updated and out_idx1 / out_idx2 undefined.                 */
size_t rigl_prune_regrow_selections_workspace_bytes(int64_t n);
int rigl_prune_regrow_selections(const RiglPruneRegrowLayer* layer /* host */,
                                 const RiglPruneRegrowParams* params,
                                 uint32_t* out_mask1_bits, uint32_t* out_mask2_bits,
                                 int32_t* out_idx1, int32_t* out_idx2,
                                 int32_t* out_counts /* device, 8, nullable */,
                                 void* workspace, size_t workspace_bytes,
                                 rigl_stream_t stream);

/* One-shot "keep the k best" mask (SNIP / DNW, sparse_optimizers.py:287-317,
 * :430-460): mask = the n_keep entries of score with the largest value, ties
 * by lower index.  Uses the same selection kernels as rigl_prune_regrow.    */
int rigl_topk_mask(const float* score, int64_t n, int64_t n_keep,
                   uint32_t* mask_bits, void* workspace,
                   size_t workspace_bytes, rigl_stream_t stream);
/* The same for many tensors in the same launches (DNW re-derives every mask
 * every step).  Workspace: rigl_prune_regrow_workspace_bytes of the sizes.   */
typedef struct RiglTopkLayer {
  const float* score;   /* fp32 [n] */
  int64_t n;
  int64_t n_keep;
  uint32_t* mask_bits;  /* out, ceil(n/32) words */
} RiglTopkLayer;
int rigl_topk_mask_batched(const RiglTopkLayer* layers /* host */,
                           int32_t n_layers, void* workspace,
                           size_t workspace_bytes, rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * K3: masked fused SGD / momentum update (+ bf16 shadow of mask*W).
 * Replaces: the TF ApplyMomentum / ApplyGradientDescent kernels behind
 *   tf.train.MomentumOptimizer(lr, momentum, use_nesterov=True)
 *   (rigl/imagenet_resnet/imagenet_train_eval.py:360-361,
 *    rigl/cifar_resnet/resnet_train_eval.py:202-203) together with the
 *   `mask *` of the masked-weight gradient and the l2_regularizer gradient:
 *     g  = (mask ? grad_scale*dense_grad : 0) + weight_decay * w
 *     a  = momentum*a + g
 *     w -= lr*g + lr*momentum*a   (nesterov)   |   w -= lr*a   (plain)
 *   every product / sum rounded to fp32 separately (no FMA contraction), so
 *   the result is bit-identical to the oracle.
 * mask_bits == NULL means "all ones" (dense tensors: BN, biases, dense stem).
 * momentum == NULL means plain gradient descent (w -= lr*g).
 * w_shadow (nullable) receives bf16(mask ? w_new : 0), same flat order.
 * ---------------------------------------------------------------------- */
int rigl_masked_sgd_momentum(int64_t n, float* w, float* momentum,
                             const float* dense_grad, const uint32_t* mask_bits,
                             float lr, float mu, float weight_decay,
                             float grad_scale, int32_t nesterov,
                             rigl_bf16* w_shadow, rigl_stream_t stream);

/* bf16 shadows of mask*W for the conv kernels:  hwio[i] = bf16(mask_i?w_i:0)
 * in flat HWIO order (dgrad operand) and ohwi = the [cout][kh*kw*cin]
 * transpose (fwd operand).  Either output may be NULL.  k = kh*kw*cin.      */
int rigl_pack_weights(const float* w, const uint32_t* mask_bits /* nullable */,
                      int32_t k, int32_t cout, rigl_bf16* hwio, rigl_bf16* ohwi,
                      rigl_stream_t stream);
/* The same for a whole model in one launch per 64 tensors (`layers` is a host
 * array; it is consumed before the call returns).                          */
typedef struct RiglPackLayer {
  const float* w;
  const uint32_t* mask_bits; /* nullable */
  rigl_bf16* hwio;           /* nullable */
  rigl_bf16* ohwi;           /* nullable */
  int32_t k, cout;
} RiglPackLayer;
int rigl_pack_weights_batched(const RiglPackLayer* layers, int32_t n_layers,
                              rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * K1: masked convolution as an implicit GEMM on MFMA (bf16 in, fp32
 * accumulate).  Replaces: layers.masked_conv2d / masked_fully_connected
 *   (rigl/imagenet_resnet/pruning_layers.py:139-157, :222-233), i.e.
 *   y = conv2d(x, mask*W), and their TF autodiff:
 *   dx = conv2d_backprop_input(dy, mask*W),
 *   dW_dense = conv2d_backprop_filter(x, dy)   (dense: RigL's grow score,
 *   rigl/sparse_optimizers_base.py:478-485).
 * The weight operand is the packed bf16 shadow of mask*W (rigl_pack_weights),
 * so the mask costs no bandwidth inside the conv.
 * Padding is explicit (pad_top/left; bottom/right implied by out size), which
 * covers TF 'SAME' (incl. its asymmetric stride-2 case), 'VALID' and the
 * reference's fixed_padding (resnet_model.py:85-106).  FC = 1x1 conv, H=W=1.
 * ---------------------------------------------------------------------- */
typedef struct RiglConvDesc {
  int32_t n, h, w, cin;      /* input  NHWC  */
  int32_t ho, wo, cout;      /* output NHWC  */
  int32_t kh, kw;
  int32_t stride_h, stride_w;
  int32_t pad_top, pad_left;
} RiglConvDesc;

size_t rigl_conv2d_workspace_bytes(const RiglConvDesc* d, int32_t which /*0 fwd,1 dgrad,2 wgrad*/);
int rigl_masked_conv2d_fwd(const RiglConvDesc* d, const rigl_bf16* x,
                           const rigl_bf16* w_ohwi, rigl_bf16* y,
                           void* workspace, size_t workspace_bytes,
                           rigl_stream_t stream);
int rigl_masked_conv2d_dgrad(const RiglConvDesc* d, const rigl_bf16* dy,
                             const rigl_bf16* w_hwio, rigl_bf16* dx,
                             void* workspace, size_t workspace_bytes,
                             rigl_stream_t stream);
/* Forward conv that also leaves the batch-norm statistics of its output
 * behind (the reference follows every conv with tf.layers.batch_normalization,
 * resnet_model.py:41-82, whose first pass re-reads the whole activation):
 * stats[p][0][c] = sum, stats[p][1][c] = sum of squares of the bf16-rounded
 * outputs of 128-row tile p, p < rigl_conv2d_stats_parts(d), computed in the
 * epilogue from the tile already in LDS (deterministic, no atomics).  The
 * parts are disjoint row sets that cover the output; which rows a part holds
 * is the kernel's choice: rows [128p, 128p + 128) for every layer but the
 * ImageNet stem (7x7/2, 3 -> 64 channels, knob "stem_direct"), whose kernel
 * works on 16 x 16 output-pixel tiles t = (image, tile row, tile column) and
 * leaves part 2t + h = pixel rows [8h, 8h + 8) of tile t.
 * stats == NULL: plain rigl_masked_conv2d_fwd.                              */
int32_t rigl_conv2d_stats_parts(const RiglConvDesc* d);
int rigl_masked_conv2d_fwd_stats(const RiglConvDesc* d, const rigl_bf16* x,
                                 const rigl_bf16* w_ohwi, rigl_bf16* y,
                                 float* stats, size_t stats_floats,
                                 void* workspace, size_t workspace_bytes,
                                 rigl_stream_t stream);
/* y = conv(relu(bn(x_pre)), mask*W) with the batch norm's APPLY pass done on
 * the conv's operand load (the reference runs batch_norm_relu between every
 * two convs, resnet_model.py:41-82,456-470: conv2 -> bn -> relu -> conv3):
 * x_pre = the pre-batch-norm tensor (the previous conv's output), scale_shift
 * = [2][cin] fp32, the per-channel scale = gamma * invstd and shift = beta -
 * mean * scale (rows 2-3 of rigl_bn_fwd_statistics' outputs, contiguous).
 * Every operand element becomes bf16(max(fma(x, scale, shift), 0)) -- the
 * rounding points of rigl_bn_fwd(relu = 1) -- in registers; a_out (bf16, the
 * shape of x_pre) receives that activated tensor as a side output (the
 * backward's weight-gradient operand and ReLU mask), so the separate apply
 * pass (read x_pre, write a_out) is not run.  y, a_out and the statistics
 * parts are bit-identical to rigl_bn_fwd + rigl_masked_conv2d_fwd_stats.
 * Only where rigl_conv2d_fwd_takes_bn_input(d) == 1 (1x1 / stride-1 layers on
 * the row-streaming body, knob "bn_on_load"); else RIGL_EUNSUPPORTED.  The
 * statistics parts are those of rigl_conv2d_stats_parts(d).                 */
int32_t rigl_conv2d_fwd_takes_bn_input(const RiglConvDesc* d);
int rigl_masked_conv2d_fwd_bnrelu(const RiglConvDesc* d, const rigl_bf16* x_pre,
                                  const float* scale_shift, rigl_bf16* a_out,
                                  const rigl_bf16* w_ohwi, rigl_bf16* y,
                                  float* stats /* nullable */, size_t stats_floats,
                                  void* workspace, size_t workspace_bytes,
                                  rigl_stream_t stream);
/* dx = conv2d_backprop_input(dy, mask*W) + addend: the gradient accumulation
 * TF's autodiff emits (AddN) where a tensor feeds two consumers -- a residual
 * block's input feeds conv1 and the shortcut (resnet_model.py:300-330, 374-420)
 * -- folded into the dgrad epilogue.  addend: bf16 NHWC like dx, nullable
 * (then identical to rigl_masked_conv2d_dgrad); may alias dx.
 * out = bf16(bf16(dgrad) + addend), bit-identical to dgrad followed by an add. */
int rigl_masked_conv2d_dgrad_acc(const RiglConvDesc* d, const rigl_bf16* dy,
                                 const rigl_bf16* w_hwio, const rigl_bf16* addend,
                                 rigl_bf16* dx, void* workspace,
                                 size_t workspace_bytes, rigl_stream_t stream);
/* dw: fp32 [kh][kw][cin][cout], DENSE (overwritten, not accumulated).       */
int rigl_masked_conv2d_wgrad(const RiglConvDesc* d, const rigl_bf16* x,
                             const rigl_bf16* dy, float* dw, void* workspace,
                             size_t workspace_bytes, rigl_stream_t stream);

/* dW and (dx != NULL) dX = dgrad (+ addend) of one conv in a single call, and
 * for ordinary layers a single launch: the dgrad and the split-K wgrad
 * workgroups share one grid (they are independent and each alone leaves much
 * of the chip idle), followed by the split-K reduce.  dX is bit-identical to
 * rigl_masked_conv2d_dgrad_acc; dW is the same sum split over fewer pixel ranges
 * than rigl_masked_conv2d_wgrad's plan (the weight-gradient workgroups leave
 * room for the dgrad tiles) -- deterministic, equal to it up to fp32 reassociation.
 * workspace: rigl_conv2d_workspace_bytes(d, 2).                              */
int rigl_masked_conv2d_bwd(const RiglConvDesc* d, const rigl_bf16* x,
                           const rigl_bf16* dy, const rigl_bf16* w_hwio,
                           const rigl_bf16* addend /* nullable */, float* dw,
                           rigl_bf16* dx /* nullable */, void* workspace,
                           size_t workspace_bytes, rigl_stream_t stream);

/* rigl_masked_conv2d_bwd whose addend is the gradient of a SUBSAMPLED view of
 * the conv's input: addend = bf16 [n][ceil(h / sub_h)][ceil(w / sub_w)][cin],
 * added to dX at the pixels with h % sub_h == 0 and w % sub_w == 0 (elsewhere
 * dX = dgrad).  The first block of a ResNet group reads its input twice -- conv1
 * and the strided 1x1 projection shortcut (resnet_model.py:456-501,
 * conv2d_fixed_padding with kernel 1: no padding, pixels (2i, 2j)) -- and the
 * projection's input gradient is zero off that grid; it is computed as the dgrad
 * of a stride-1 1x1 conv over the [n, ho, wo] grid (rigl_masked_conv2d_dgrad with
 * such a descriptor) and handed over compact: a quarter of the bytes written and
 * read, and none of the strided dgrad's zero rows.  dX is bit-identical to
 * rigl_masked_conv2d_bwd fed the same gradient scattered into a zero tensor.
 * Runs on the implicit-GEMM body (or, round 6, the channel-sliced single-pass
 * backward where that is the layer's kernel: the only bodies with this
 * addressing); sub_h = sub_w = 1 is rigl_masked_conv2d_bwd.                    */
/* The other half of that hand-over: the backward of a STRIDED 1x1 conv without
 * padding (ho = ceil(h / stride_h), wo likewise) with dX on the conv's own grid:
 * dx_grid = bf16 [n][ho][wo][cin], the gradient at the pixels the conv read (it is
 * zero at every other pixel, which is not written anywhere) = the dgrad of the
 * stride-1 1x1 conv over that grid; dW (dense, overwritten) from x as it lies.
 * One shared launch like rigl_masked_conv2d_bwd; workspace as for it.           */
int rigl_masked_conv2d_bwd_grid(const RiglConvDesc* d, const rigl_bf16* x,
                                const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                float* dw, rigl_bf16* dx_grid, void* workspace,
                                size_t workspace_bytes, rigl_stream_t stream);
int rigl_masked_conv2d_bwd_sub(const RiglConvDesc* d, const rigl_bf16* x,
                               const rigl_bf16* dy, const rigl_bf16* w_hwio,
                               const rigl_bf16* addend, int32_t sub_h, int32_t sub_w,
                               float* dw, rigl_bf16* dx, void* workspace,
                               size_t workspace_bytes, rigl_stream_t stream);
/* rigl_masked_conv2d_bwd whose addend arrives UNMASKED with a 1-bit-per-element
 * mask (addend_bits: [n*h*w][cin / 8] bytes, bit j of a byte = channel 8 b + j;
 * set = the addend counts there): dx = bf16(bf16(dgrad) + (bit ? addend : 0)).
 * Replaces the tf.where / relu-gradient op that masks the shortcut's gradient of
 * relu(bn3 + shortcut) (rigl/imagenet_resnet/resnet_model.py:497-501 through
 * autodiff): the batch norm's backward hands its output gradient over as it is,
 * with the ReLU bits its forward left, and never writes the masked copy.  Only for
 * the layers rigl_conv2d_bwd_takes_masked_addend says 1 for (the channel-sliced
 * single-pass backward, bwdslice.hpp); RIGL_EUNSUPPORTED otherwise.            */
int32_t rigl_conv2d_bwd_takes_masked_addend(const RiglConvDesc* d);
int rigl_masked_conv2d_bwd_masked(const RiglConvDesc* d, const rigl_bf16* x,
                                  const rigl_bf16* dy, const rigl_bf16* w_hwio,
                                  const rigl_bf16* addend, const uint8_t* addend_bits,
                                  float* dw, rigl_bf16* dx, void* workspace,
                                  size_t workspace_bytes, rigl_stream_t stream);

/* rigl_masked_conv2d_bwd with the BATCH-NORM BACKWARD REDUCTIONS of the
 * tensor dX is the gradient of riding in the dgrad epilogue.  In the reference every
 * conv input is y = relu?(batch_norm(x_bn) [+ shortcut]) (resnet_model.py:41-82,
 * 456-501), and autodiff's batch-norm gradient starts with two per-channel sums
 * over the whole tensor, sum(dz) and sum(dz * xhat), dz = relu-masked dL/dy -- a
 * pass that re-reads dL/dy and x_bn.  The dgrad tile being stored IS dL/dy (after
 * the fused `addend`), so the epilogue reads the matching x_bn tile and leaves
 *   partial[p][0][c] = sum dz,  partial[p][1][c] = sum dz * (x_bn - mean) * invstd
 * per row tile p < rigl_conv2d_dgrad_stats_parts(d) (fixed order, no atomics);
 * rigl_bn_bwd_stats consumes them instead of running its reduction pass.
 * x_bn / relu_bits have dX's shape [N,H,W,Cin]; params = the forward's saved
 * [4][Cin] (mean, invstd, scale, shift); relu_bits (1 bit per element, from
 * rigl_bn_fwd_stats) or NULL = the ReLU mask is recomputed from x_bn.
 * rigl_conv2d_dgrad_stats_parts returns 0 for layers whose dgrad kernel has no
 * such epilogue (cin or cout not a multiple of 8).                            */
typedef struct RiglBnReduceFuse {
  const rigl_bf16* x;
  const uint8_t* relu_bits;   /* nullable */
  const float* params;        /* [4][Cin] */
  int32_t relu;
  float* partial;             /* out: [parts][2][Cin] */
  size_t partial_floats;
} RiglBnReduceFuse;
int32_t rigl_conv2d_dgrad_stats_parts(const RiglConvDesc* d);
int rigl_masked_conv2d_bwd_bn(const RiglConvDesc* d, const rigl_bf16* x,
                              const rigl_bf16* dy, const rigl_bf16* w_hwio,
                              const rigl_bf16* addend, float* dw, rigl_bf16* dx,
                              void* workspace, size_t workspace_bytes,
                              const RiglBnReduceFuse* bn /* nullable */,
                              rigl_stream_t stream);

/* K1d: dense depthwise convolution (depth multiplier 1), NHWC bf16, fp32 HWIO
 * weights [kh][kw][c][1] read directly.  Replaces
 * contrib_layers.separable_conv2d(num_outputs=None)
 * (rigl/imagenet_resnet/mobilenetv1_model.py:81-92) and its autodiff; these
 * layers are NOT masked in the reference.  d->cin == d->cout,
 * channels % 8 == 0.  HBM-bound, no MFMA.  dw is overwritten (dense fp32).   */
size_t rigl_depthwise_conv2d_workspace_bytes(const RiglConvDesc* d);
int rigl_depthwise_conv2d_fwd(const RiglConvDesc* d, const rigl_bf16* x,
                              const float* w, rigl_bf16* y, rigl_stream_t stream);
/* The forward that also leaves the batch-norm statistics of its output (every
 * depthwise conv of mobilenetv1_model.py:188-198 is followed by a batch norm):
 * stats[p][0][c] = sum, stats[p][1][c] = sum of squares of the bf16-rounded
 * outputs of partial row p < rigl_depthwise_conv2d_stats_parts(d) (0: shape has
 * no such epilogue: 3x3, stride 1 / 2, (channels / 8) dividing 256 do).
 * Deterministic; consumed by rigl_bn_fwd_stats.  stats == NULL: plain forward. */
int32_t rigl_depthwise_conv2d_stats_parts(const RiglConvDesc* d);
int rigl_depthwise_conv2d_fwd_stats(const RiglConvDesc* d, const rigl_bf16* x,
                                    const float* w, rigl_bf16* y, float* stats,
                                    size_t stats_floats, rigl_stream_t stream);
int rigl_depthwise_conv2d_dgrad(const RiglConvDesc* d, const rigl_bf16* dy,
                                const float* w, rigl_bf16* dx,
                                rigl_stream_t stream);
int rigl_depthwise_conv2d_wgrad(const RiglConvDesc* d, const rigl_bf16* x,
                                const rigl_bf16* dy, float* dw, void* workspace,
                                size_t workspace_bytes, rigl_stream_t stream);

/* Direct (non-MFMA) kernels, same operand layouts, any channel count: taken
 * for shapes the MFMA path reports RIGL_EUNSUPPORTED for (cin/cout not a
 * multiple of 8 -- MNIST MLP 784-300-100-10, 10-class logits) and used as an
 * independent on-device cross-check.                                        */
int rigl_conv2d_fwd_ref(const RiglConvDesc* d, const rigl_bf16* x,
                        const rigl_bf16* w_ohwi, rigl_bf16* y,
                        rigl_stream_t stream);
int rigl_conv2d_dgrad_ref(const RiglConvDesc* d, const rigl_bf16* dy,
                          const rigl_bf16* w_hwio, rigl_bf16* dx,
                          rigl_stream_t stream);
int rigl_conv2d_wgrad_ref(const RiglConvDesc* d, const rigl_bf16* x,
                          const rigl_bf16* dy, float* dw, rigl_stream_t stream);

/* K1 in fp32 arithmetic -- the VALIDATION twin of the bf16 kernels, not the
 * measured path.  The reference trains in float32 unless --precision=bfloat16
 * (imagenet_train_eval.py:56-59, 553-554) and the path's float tolerance is
 * quoted against fp32; these three entry points compute conv(x, mask * W), dX
 * (+ addend) and the dense dW from fp32 NHWC activations and the fp32 master
 * weights (HWIO; mask_bits as in rigl_pack_weights, NULL = all ones, applied on
 * the fly) on v_mfma_f32_32x32x2_f32, any shape, deterministic (the weight
 * gradient's position slabs are summed in a fixed order; workspace =
 * rigl_conv2d_wgrad_f32_workspace_bytes(d), may be 0).  The host mirror takes
 * this path when the activations it is handed are fp32 (rigl_amd.workloads:
 * precision='float32', or knob "k1_fp32" = 1); tests/test_k1_fp32_gpu.py trains
 * WRN-22 and ResNet-50 on it against a float64 evaluation of the model.      */
int rigl_masked_conv2d_fwd_f32(const RiglConvDesc* d, const float* x,
                               const float* w_hwio, const uint32_t* mask_bits /* nullable */,
                               float* y, rigl_stream_t stream);
int rigl_masked_conv2d_dgrad_f32(const RiglConvDesc* d, const float* dy,
                                 const float* w_hwio, const uint32_t* mask_bits /* nullable */,
                                 const float* addend /* nullable, may alias dx */,
                                 float* dx, rigl_stream_t stream);
size_t rigl_conv2d_wgrad_f32_workspace_bytes(const RiglConvDesc* d);
int rigl_masked_conv2d_wgrad_f32(const RiglConvDesc* d, const float* x,
                                 const float* dy, float* dw, void* workspace,
                                 size_t workspace_bytes, rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * Glue between two masked convs: batch-norm with batch statistics (+ residual
 * add) (+ ReLU) on NHWC bf16 rows [m][c], fp32 parameters -- what the reference
 * runs as tf.layers.batch_normalization(fused=True) + tf.nn.relu (+ `inputs +
 * shortcut`), rigl/imagenet_resnet/resnet_model.py:41-82, 456-501.  Not a
 * graded hot-path row; fused here because it is HBM-bound and sits inside the
 * images/s number.  y = relu?(gamma*(x-mean)*invstd + beta (+ residual)).
 * Statistics are deterministic (fixed-order partial sums).  c % 8 == 0.
 * save_* ([c] fp32 each) carry mean, 1/sqrt(var+eps), gamma*invstd and
 * beta-mean*gamma*invstd to the backward call.  running_* may be NULL.
 * Backward: the ReLU mask of (bn + residual) comes from relu_bits (1 bit per
 * element, written by rigl_bn_fwd_stats: 1/16 of the bytes of y) or, failing
 * that, from y; with both NULL it is recomputed from x (no residual).  dresidual (nullable)
 * receives the relu-masked dy.  dgamma / dbeta are overwritten.
 * ---------------------------------------------------------------------- */
size_t rigl_bn_workspace_bytes(int64_t m, int32_t c);
int rigl_bn_fwd(int64_t m, int32_t c, const rigl_bf16* x,
                const rigl_bf16* residual /* nullable */, const float* gamma,
                const float* beta, float* running_mean, float* running_var,
                float momentum, float eps, int32_t relu, rigl_bf16* y,
                float* save_mean, float* save_invstd, float* save_scale,
                float* save_shift, void* workspace, size_t workspace_bytes,
                rigl_stream_t stream);
/* As rigl_bn_fwd, with the statistics pass replaced by the partial sums a
 * producer left behind: stats = [stats_parts][2][c] fp32 (sum x, sum x^2 over
 * disjoint row sets covering all m rows), e.g. from
 * rigl_masked_conv2d_fwd_stats.  stats == NULL behaves like rigl_bn_fwd
 * (workspace required); with stats the workspace may be NULL.               */
int rigl_bn_fwd_stats(int64_t m, int32_t c, const rigl_bf16* x,
                      const rigl_bf16* residual /* nullable */,
                      const float* gamma, const float* beta,
                      float* running_mean, float* running_var, float momentum,
                      float eps, int32_t relu, rigl_bf16* y, float* save_mean,
                      float* save_invstd, float* save_scale, float* save_shift,
                      const float* stats, int32_t stats_parts,
                      uint8_t* relu_bits /* nullable, out: ceil(m*c/8) bytes */,
                      void* workspace, size_t workspace_bytes, rigl_stream_t stream);
int rigl_bn_bwd(int64_t m, int32_t c, const rigl_bf16* x,
                const rigl_bf16* y /* nullable */,
                const uint8_t* relu_bits /* nullable */, const rigl_bf16* dy,
                const float* gamma, const float* save_mean,
                const float* save_invstd, const float* save_scale,
                const float* save_shift, int32_t relu, rigl_bf16* dx,
                rigl_bf16* dresidual /* nullable */, float* dgamma,
                float* dbeta, void* workspace, size_t workspace_bytes,
                rigl_stream_t stream);
/* rigl_bn_bwd whose reduction pass is replaced by the producer's partial sums:
 * stats = [stats_parts][2][c] (sum dz, sum dz * xhat) from the dgrad epilogue
 * that wrote dy (rigl_masked_conv2d_bwd_bn); stats == NULL: plain rigl_bn_bwd. */
int rigl_bn_bwd_stats(int64_t m, int32_t c, const rigl_bf16* x,
                      const rigl_bf16* y /* nullable */,
                      const uint8_t* relu_bits /* nullable */, const rigl_bf16* dy,
                      const float* gamma, const float* save_mean,
                      const float* save_invstd, const float* save_scale,
                      const float* save_shift, int32_t relu, rigl_bf16* dx,
                      rigl_bf16* dresidual /* nullable */, float* dgamma,
                      float* dbeta, const float* stats, int32_t stats_parts,
                      void* workspace, size_t workspace_bytes, rigl_stream_t stream);

/* Two batch norms meeting in one add: out = relu?(bn(x) + bn2(x2)) -- the first
 * block of every ResNet group, whose shortcut is projection conv + batch norm
 * (resnet_model.py:456-501: `shortcut = projection_shortcut(inputs)` ...
 * `tf.nn.relu(inputs + shortcut)`).  The normalised shortcut and (backward) the
 * relu-masked gradient are never written.  bn2's statistics come from
 * rigl_bn_fwd_statistics (scale2 / shift2 = its save_scale / save_shift); the
 * forward rounds the shortcut to bf16 where bn2 alone would have stored it and
 * the backward keeps rigl_bn_bwd's row partition, so y, relu_bits, dx, dx2 and
 * all four parameter gradients are bit-identical to rigl_bn_fwd(x2) ->
 * rigl_bn_fwd_stats(x, residual) and rigl_bn_bwd(dresidual) -> rigl_bn_bwd.
 * Backward: relu on, relu_bits required.                                      */
int rigl_bn_add_bn_fwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* x2,
                       const float* scale2, const float* shift2,
                       const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float momentum, float eps, int32_t relu,
                       rigl_bf16* y, float* save_mean, float* save_invstd,
                       float* save_scale, float* save_shift, const float* stats,
                       int32_t stats_parts, uint8_t* relu_bits /* nullable */,
                       void* workspace, size_t workspace_bytes, rigl_stream_t stream);
size_t rigl_bn_add_bn_bwd_workspace_bytes(int64_t m, int32_t c);
int rigl_bn_add_bn_bwd(int64_t m, int32_t c, const rigl_bf16* x, const rigl_bf16* x2,
                       const uint8_t* relu_bits, const rigl_bf16* dy,
                       const float* gamma, const float* save_mean, const float* save_invstd,
                       const float* gamma2, const float* save_mean2, const float* save_invstd2,
                       rigl_bf16* dx, rigl_bf16* dx2, float* dgamma, float* dbeta,
                       float* dgamma2, float* dbeta2, void* workspace,
                       size_t workspace_bytes, rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * Stateless random tensors with TensorFlow's bit layout:
 *   out[i] = rnd_i * scale + shift,  rnd = tf.random.stateless_{uniform,normal}
 *   (shape=[n], seed=[seed0, seed1], float32)  -- Philox-4x32-10 keyed by the
 *   scrambled seed pair, element i = lane i%4 of counter + i/4, Box-Muller for
 *   normals.  Replaces stateless_random_normal (drop noise, stddev = scale) and
 *   stateless_random_uniform (SET grow scores) of
 *   rigl/sparse_optimizers_base.py:402-418; seed0 = int32(offset +
 *   hash(name + 'drop'|'grow')), seed1 = int32(global_step).
 *   dist: 0 uniform [0,1), 1 standard normal.
 * ---------------------------------------------------------------------- */
int rigl_stateless_random(float* out, int64_t n, int32_t seed0, int32_t seed1,
                          int32_t dist, float scale, float shift,
                          rigl_stream_t stream);
/* The same for many tensors in one launch -- every layer's drop noise of one
 * mask update (the reference builds one stateless_random_normal op per layer,
 * rigl/sparse_optimizers_base.py:526-534, :411-416). */
typedef struct {
  float* out;
  int64_t n;
  int32_t seed0, seed1;
  int32_t dist;            /* 0 uniform [0,1), 1 standard normal */
  float scale, shift;
} RiglRandomItem;
int rigl_stateless_random_batched(const RiglRandomItem* items, int32_t n_items,
                                  rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * Glue: max pooling, NHWC bf16 (tf.layers.max_pooling2d(3, 2, 'SAME') after the
 * stem, rigl/imagenet_resnet/resnet_model.py:637-644).  The descriptor is a
 * RiglConvDesc with cin == cout (% 8 == 0); padding explicit, windows clipped
 * to the image.  argmax: 1 byte per output element (r*kw+s of the FIRST
 * maximum in row-major window order); the backward is a deterministic gather.
 * ---------------------------------------------------------------------- */
int rigl_maxpool_fwd(const RiglConvDesc* d, const rigl_bf16* x, rigl_bf16* y,
                     uint8_t* argmax, rigl_stream_t stream);
int rigl_maxpool_bwd(const RiglConvDesc* d, const rigl_bf16* dy,
                     const uint8_t* argmax, rigl_bf16* dx, rigl_stream_t stream);
/* The stem's tail in one piece: batch_norm_relu followed by that max pooling
 * (resnet_model.py:631-644) without the activated 112x112x64 tensor -- 205 MB at
 * batch 128 -- ever being written.
 *   rigl_bn_fwd_statistics: the statistics half of rigl_bn_fwd_stats (reduction
 *     pass over x unless `stats` partial sums are given, then mean / invstd /
 *     scale / shift and the moving averages); no apply pass.
 *   rigl_bn_relu_maxpool_fwd: y, argmax = maxpool(bf16(relu(x*scale + shift))),
 *     bit-identical to rigl_bn_fwd followed by rigl_maxpool_fwd.
 *   rigl_bn_relu_maxpool_bwd: dx, dgamma, dbeta of the pair from the pooled
 *     gradient dy: both batch-norm passes gather the pooling's input gradient
 *     (rounded to bf16 as rigl_maxpool_bwd stores it) from dy / argmax on the fly.
 *     dx equals rigl_maxpool_bwd + rigl_bn_bwd up to the fp32 summation order of
 *     the two reductions.  3x3 / stride 2, (channels / 4) dividing 64.          */
int rigl_bn_fwd_statistics(int64_t m, int32_t c, const rigl_bf16* x /* nullable with stats */,
                           const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum,
                           float eps, float* save_mean, float* save_invstd,
                           float* save_scale, float* save_shift,
                           const float* stats, int32_t stats_parts,
                           void* workspace, size_t workspace_bytes, rigl_stream_t stream);
int rigl_bn_relu_maxpool_fwd(const RiglConvDesc* d, const rigl_bf16* x,
                             const float* scale, const float* shift, rigl_bf16* y,
                             uint8_t* argmax, rigl_stream_t stream);
size_t rigl_bn_relu_maxpool_bwd_workspace_bytes(const RiglConvDesc* d);
int rigl_bn_relu_maxpool_bwd(const RiglConvDesc* d, const rigl_bf16* x,
                             const rigl_bf16* dy, const uint8_t* argmax,
                             const float* gamma, const float* save_mean,
                             const float* save_invstd, const float* save_scale,
                             const float* save_shift, rigl_bf16* dx, float* dgamma,
                             float* dbeta, void* workspace, size_t workspace_bytes,
                             rigl_stream_t stream);

/* ------------------------------------------------------------------------
 * Glue: classifier head.  Global average pool over `pixels` positions of an
 * NHWC bf16 map (average_pooling2d + reshape, resnet_model.py:701-712; fp32
 * accumulation, c even) and tf.losses.softmax_cross_entropy with label
 * smoothing (imagenet_train_eval.py:578-584) on bf16 logits: row_loss[r] =
 * -sum_k t_k log softmax(z)_k with t = onehot * (1 - eps) + eps / K, and --
 * when dlogits is given -- dlogits = bf16((softmax(z) - t) * grad_scale), the
 * gradient of grad_scale * sum_r row_loss[r] (grad_scale = 1 / rows for the mean).
 * ---------------------------------------------------------------------- */
int rigl_global_avgpool_fwd(int32_t n, int32_t pixels, int32_t c, const rigl_bf16* x,
                            rigl_bf16* y, rigl_stream_t stream);
int rigl_global_avgpool_bwd(int32_t n, int32_t pixels, int32_t c, const rigl_bf16* dy,
                            rigl_bf16* dx, rigl_stream_t stream);
int rigl_softmax_xent(int32_t rows, int32_t classes, const rigl_bf16* logits,
                      const int64_t* labels, float label_smoothing, float grad_scale,
                      float* row_loss, rigl_bf16* dlogits /* nullable */, rigl_stream_t stream);

/* Optional per-kernel timing (HIP events recorded on the launch stream around
 * every K1/K2/K3 launch while enabled).  rigl_prof_collect synchronises the
 * recorded events and returns accumulated milliseconds / launch counts per
 * kernel family: 0 conv_fwd, 1 conv_dgrad, 2 conv_wgrad, 3 prune_regrow,
 * 4 sgd_momentum, 5 pack_weights, 6 conv_bwd (the fused dgrad + wgrad launch
 * of rigl_masked_conv2d_bwd, with its split-K reduce), 7 depthwise (K1d, the
 * three rigl_depthwise_conv2d_* entry points).                              */
#define RIGL_PROF_KINDS 8
int rigl_prof_enable(int32_t on);
int rigl_prof_collect(double* ms_per_kind /*[8]*/, int64_t* launches /*[8]*/);
/* The same events one by one, in launch order (instead of rigl_prof_collect,
 * which consumes them too): kind as above; tag = (h, w, cin, cout, kh, stride_h)
 * of the conv descriptor the launch belongs to (zeros for K2 / K3 / pack); ms =
 * that dispatch's own duration.  Writes at most `cap` records, returns the
 * number of recorded launches in *n_launches (the rest is dropped).  For the
 * per-layer IN-STEP table of tools/instep_table.py.    */
typedef struct RiglProfLaunch {
  int32_t kind;
  int32_t tag[6];
  float ms;
} RiglProfLaunch;
int rigl_prof_collect_launches(RiglProfLaunch* out /* host */, int64_t cap, int64_t* n_launches);

/* Measurement aid (the MFMA peak "re-measured on the box"):
 * enqueues blocks x 4 waves, each issuing iters x 8 independent
 * v_mfma_f32_32x32x16_bf16 (32768 FLOP per wave-instruction) on register
 * operands; the caller brackets it with events.  `sink`: blocks*256 floats.  */
int rigl_probe_mfma_bf16(int32_t blocks, int32_t iters, float* sink, rigl_stream_t stream);

/* Development knobs: the run-time twin of the RIGL_* environment variables, so
 * that one process can A/B kernel selections (tools/pp_sweep.py,
 * tests/k1_check.py).  Process-wide state, NOT part of the drop-in surface: a
 * caller that never sets a knob gets the built-in selection rules; a knob never
 * set reads the environment variable RIGL_<KEY IN UPPER CASE> once.  Keys:
 *   "pp_fwd" / "pp_dgrad"   K1 tile of the 8-wave ping-pong body for the
 *                           forward / stand-alone dgrad GEMM: -1 built-in rule
 *                           (default), 0 never, 1 256x256, 2 128x256,
 *                           3 256x128, 4 512x128 (ignored where the shape does
 *                           not admit the tile);
 *   "pp_bwd"                the shared backward launch on the ping-pong bodies
 *                           (dgrad tiles + 256x256 weight-gradient tiles): -1
 *                           rule, 0 never, 1 / 2 dgrad on 256x256 / 128x256;
 *   "pp_wgrad"              1: stand-alone weight gradient on the ping-pong
 *                           body wherever legal (rule: off);
 *   "pp_slab_mb"            cap on a layer's split-K slab bytes (40);
 *   "pp_bwd_min_kt"         shortest dgrad reduction (K-tiles of 64) the
 *                           pp_bwd rule takes (16);
 *   "wgrad_il"              0: the splits of a weight gradient take contiguous
 *                           pixel ranges instead of interleaved K-tiles (1);
 *   "pp_ksplit"             0: no two-way K split of the few-tile forwards
 *                           (7x7 3x3 layers at batch 128) (1);
 *   "bn_il"                 0: a batch-norm reduction part is a contiguous range
 *                           of rows instead of every parts-th group (1);
 *   "bn_nt"                 non-temporal accesses of the batch-norm apply
 *                           passes: 0 none, 1 stores, 2 loads and stores (2;
 *                           -0.10 ms per ResNet-50 step); "bn_nt_mb" applies it
 *                           only to tensors of at least that many MB (0);
 *   "stem_wgrad"            0: only the stem's forward on stem.hpp (1);
 *   "stem_direct"           0: the ImageNet stem through the generic bodies
 *                           over a padded 4-channel copy instead of the
 *                           LDS-resident-patch kernels (1);
 *   "bwd1x1"                0: the single-pass backward of the 56x56-class 1x1
 *                           layers (64->256, 64->64; dY read once) off (1);
 *   "bwd1x1_il"             0: contiguous pixel ranges per workgroup instead of
 *                           interleaved K-tiles (1); "bn_fin1" 0: the one-level
 *                           forward batch-norm finalize for every size (1);
 *   "c3x3"                  0: the 3x3 / 64 -> 64 / stride-1 layers (24 <= w <=
 *                           62) through the generic bodies instead of the
 *                           LDS-resident-patch kernels of c3x3.hpp (1);
 *   "rowstream"             the row-streaming body of the 1x1 / stride-1 GEMMs
 *                           (rowstream.hpp: rows global -> registers, the
 *                           filter slice stationary in LDS): 0 never, 1 the
 *                           layers it measured faster on in the step (default:
 *                           forwards with reductions <= 256 channels, the
 *                           256 <- 64 dgrad), 2 every legal layer (reductions
 *                           of 64 .. 512 channels, both directions);
 *   "k1_fp32"               1: rigl_amd.workloads hands the models fp32
 *                           activations (the fp32 validation kernels of K1;
 *                           read by the host mirror, not by the library) (0).
 * rigl_conv2d_stats_parts(d) follows these selections: the c3x3 / rowstream kernels
 * leave ONE statistics part per persistent workgroup (of a column slice), not one per 128 rows.
 * Knobs whose A/B measurements said "no" in rounds 2-4 are gone with their
 * kernels (k1_nt_mb, pp_ph, bwd1x1_wgs, bwd1x1_256x64, stem_nt, RIGL_WGRAD_DEFER,
 * RIGL_WGRAD_STREAM, RIGL_CONV_STAGES, RIGL_CONV_W4_KT, RIGL_WGRAD_TR; round 4
 * also measured and did not keep bn_parts / bn_rpl and a side-stream reduce;
 * round 5 replaced x1x1.hpp and its knobs by rowstream.hpp).
 * Round 6: "bwdslice" (0 = off, 1 = default: the channel-sliced single-pass
 * backward of the 1x1 / stride-1 layers with cin % 128 == 0 and cout 128 / 256,
 * bwdslice.hpp) with "bwdslice512" (0 / 1 = default: its 64-channel-slice form
 * for cout 512, cin % 64 == 0); "bn_on_load" (0: rigl_conv2d_fwd_takes_bn_input says 0 for
 * every layer; 1 = default: the row-streaming forwards except the 64-channel /
 * 128-column variant take the transform -- whether a model USES the entry point
 * is the caller's choice: the host mirror does with RIGL_BN_ON_LOAD=1 only, it
 * measured 0.03 ms slower per ResNet-50 step).
 * No reference counterpart (the reference selects cuDNN/TPU algorithms inside
 * TensorFlow: rigl/imagenet_resnet/pruning_layers.py:139-157).              */
int rigl_tune_set(const char* key, int32_t value);
int32_t rigl_tune_get(const char* key, int32_t dflt);
/* Back to "never set" (the RIGL_<KEY> environment variable, else the built-in
 * default); rigl_tune_set(key, INT32_MIN) is the same call.                  */
int rigl_tune_unset(const char* key);
/* Counts the rigl_tune_set / rigl_tune_unset calls of this process: whoever caches
 * a plan-dependent answer of the library (rigl_conv2d_stats_parts,
 * rigl_conv2d_workspace_bytes ...) keys the cache on it -- a knob flipped through
 * ANY binding then invalidates it (the Python mirror does: ops._plan_cached). */
uint64_t rigl_tune_generation(void);
/* The address of that counter (valid for the life of the process; read it as a
 * volatile uint64): a binding that checks it on every call -- launch-bound models
 * make hundreds of cached-plan lookups per step -- reads memory instead of
 * crossing the FFI.                                                           */
const volatile uint64_t* rigl_tune_generation_addr(void);

#ifdef __cplusplus
}
#endif
#endif /* RIGL_HIP_H_ */
