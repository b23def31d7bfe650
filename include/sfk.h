/*
 * sfk.h -- C ABI of libsfk.so: the MI355X (gfx950) kernels behind the SlowFast training hot path.
 *
 * The reference (zc402/video-classification) has no FFI of its own: its hot path is
 *   train.py:216-252   Trainer.train_epoch  (forward, CrossEntropyLoss, backward, Adam)
 *   model/my_slowfast.py:44-126,260-344     model construction + lateral fusion forward
 * and every device kernel is launched implicitly by torch.nn modules.  Each entry point below replaces one
 * such implicit operator class; the comment above it names the reference construction site (file:line) and
 * the torch.nn semantics it must reproduce.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - plain C: pointers, sizes, POD structs.  No torch / C++ types cross this boundary.
 *   - every buffer (inputs, outputs, workspaces) is owned by the caller; nothing is allocated, freed or
 *     retained past the call.  All pointers are DEVICE pointers unless a comment says "host".
 *   - calls are asynchronous on the caller's hipStream_t (passed as void*), never synchronise, and are
 *     safe to capture into a hipGraph.  No global mutable state: the only process-wide datum is the tuning table
 *     below, written by sfk_init() BEFORE the first launch and read-only afterwards; no entry point reads the
 *     environment.
 *   - return value: SFK_OK or a negative sfk_status; no exceptions, no abort.
 *   - feature maps are CHANNELS-LAST in HBM: element (n,t,h,w,c) of an sfk_fmap lives at
 *         ptr[ (((n*T + t)*H + h)*W + w) * ld + c_off + c ]
 *     `ld` (elements between consecutive pixels) >= c_off + c lets a producer write straight into a slice of
 *     a wider buffer -- this is how torch.cat([x_slow, fuse], 1) (my_slowfast.py:343) is eliminated.
 *   - dtype: SFK_BF16 (storage bf16, fp32 accumulate; the benchmark precision) or SFK_F32 (storage fp32,
 *     exact-fp32 MFMA; the parity precision).  Statistics, master weights and gradients are always fp32.
 */
#ifndef SFK_H
#define SFK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped on EVERY change of a struct layout, a prototype or the meaning of an argument / tuning field.  include/sfk.abi holds
 * (version, hash of this header's declarations); tests/test_abi_cpu.py fails when the hash moves without the version
 * (tools/abi_lock.py refuses to re-lock the same version).  History: 10 = struct_size handshake in the descriptor structs; 11 = sfk_conv_wgrad_wants_workspace,
 * sfk_tuning.wgrad_target_256 / wgrad_min_stages_256 (retired in 15); 12 = sfk_bn_apply(out_sums), sfk_bn_tail_fwd / _bwd take the column sums
 * of `a` from it (no constant-1 channel group beside the activation any more); 13 = sfk_conv_pw_dual; 14 = sfk_tuning.igemm_p8 / wgrad_p8,
 * sfk_conv_igemm_family value 4; 15 = sfk_tuning.igemm_halo, family value 5; 16 = sfk_tuning.wgrad_band; 17 = sfk_tuning.stem_v3; 18 = sfk_bn_finalize_apply (19, 20: SFK_BN_SYNC_INTS counters),
 * sfk_bn_bwd_finalize_apply. */
#define SFK_ABI_VERSION 20
#define SFK_MAX_TAPS 16
#define SFK_BN_FOLD_ROWS 64 /* rows of the optional BatchNorm fold workspace */

typedef void* sfk_stream_t; /* hipStream_t */

typedef enum {
  SFK_OK = 0,
  SFK_ERR_INVALID = -1,     /* null pointer, non-positive extent, inconsistent descriptor   */
  SFK_ERR_UNSUPPORTED = -2, /* alignment / size outside what the kernels handle            */
  SFK_ERR_LAUNCH = -3       /* hipGetLastError() after the launch was not hipSuccess        */
} sfk_status;

typedef enum { SFK_F32 = 0, SFK_BF16 = 1 } sfk_dtype;

typedef struct {
  void* ptr;
  int32_t dtype;         /* sfk_dtype */
  int32_t n, t, h, w, c; /* logical extents */
  int32_t ld;            /* elements between consecutive pixels (>= c_off + c) */
  int32_t c_off;         /* first channel of this map inside the pixel record  */
} sfk_fmap;

/* One filter tap of an implicit-GEMM pass: the gathered pixel is  row*gs + (dt,dh,dw);  `widx` selects the
 * [cin] slice of the filter row  w[co][widx][:]. */
typedef struct {
  int8_t dt, dh, dw;
  uint8_t widx;
} sfk_tap;

/* ---------------------------------------------------------------------------------------------------------
 * sfk_conv_igemm -- Conv3d forward AND data-gradient as one implicit GEMM on MFMA.
 * Replaces: every nn.Conv3d the path constructs -- stems (my_slowfast.py:63-65), bottleneck conv_a/b/c and
 * branch1 shortcuts (pytorchvideo create_res_stage, called via my_slowfast.py:94-125), lateral fusion conv
 * (my_slowfast.py:195-202) -- and their autograd input gradients (train.py:230).
 *
 * Row space: rows m = (n, rt, rh, rw), n < x.n.  For each row and output channel co < cout:
 *     acc = sum_{tap < ntaps} sum_{ci < cin}  X[n, rt*gs[0]+dt, rh*gs[1]+dh, rw*gs[2]+dw, ci] * w[co][widx][ci]
 *   (pixels outside x's extents read as zero = the conv's zero padding), then
 *     Y[n, rt*os[0]+oo[0], rh*os[1]+oo[1], rw*os[2]+oo[2], co]  (=|+=)  acc      (+= when `accumulate`).
 *   forward conv, stride s, pad p:  gs = s, tap.d = k - p, os = 1, oo = 0, rows = output pixels.
 *   data gradient, stride 1:        x = dY, gs = 1, tap.d = p - k, w = filters with (co,ci) swapped.
 *   data gradient, stride > 1:      one pass per output-parity class: os = s, oo = class offset, taps = the taps
 *                                   of that class (see video-classification_amd/plan.py).
 * `stats` (optional): per row-tile partial sums of acc and acc^2 per channel, laid out [mtiles][cout][2],
 *   mtiles = sfk_conv_igemm_mtiles(desc).  They feed sfk_bn_finalize, so BatchNorm3d's batch statistics
 *   (my_slowfast.py:215-222; pytorchvideo norm_a/b/c) cost no extra pass over the conv output.
 * Requirements: x.dtype == y.dtype; cin % (16/sizeof(dtype)) == 0, likewise x.ld, x.c_off; cout % 4 == 0,
 *   y.ld % 4 == 0, y.c_off % 4 == 0; 16-byte aligned base pointers.
 */
/* Optional fusion of the BatchNorm-backward REDUCE into a data-gradient pass (autograd of conv -> BatchNorm3d -> ReLU
 * chains, train.py:230): when the pass's output Y is dA, the gradient w.r.t. the activation a = act(bn(y_bn) [+ shortcut]),
 * the epilogue stores  dz = result * mask  (mask as sfk_bn_bwd_reduce: (mask_src > 0) if mask_src.ptr, else
 * (y_bn*scale + shift > 0) if relu, else 1) instead of dA, and writes the partial rows
 *     partials[tile][cout][2] = (sum dz, sum dz * x_hat),  x_hat = (y_bn - mean) * invstd,  tile < sfk_conv_igemm_mtiles(d)
 * for sfk_bn_bwd_finalize -- sfk_bn_bwd_reduce's result without its three tensor reads and one write.
 * y_bn / mask_src share Y's pixel grid and channel count (own ld / c_off).  partials == NULL: off.
 * y_bn.ptr == NULL together with sfk_conv_desc.out_relu_bits: the mask is that bitmap and only  sum dz  is left
 * (partials[tile][co] = (sum dz, 0)) -- what the fused block tail (sfk_bn_tail_bwd) needs of the BatchNorm reduce.
 * Supported when sfk_conv_bnb_supported(d) != 0 (bf16, 8-channel groups 16-byte addressable, cout > 16). */
typedef struct {
  sfk_fmap y_bn;
  sfk_fmap mask_src; /* ptr == NULL: none */
  const float *mean, *invstd, *scale, *shift;
  int32_t relu;
  float* partials;
} sfk_bn_bwd_fuse;

/* Optional fused output transform of a pass whose rows are the pixels of y in order (os = 1, oo = 0, row extents = y
 * extents) -- the BatchNorm3d + shortcut + ReLU tail of a bottleneck block (pytorchvideo ResBlock: ReLU(x + norm_c(conv_c(.)))),
 * computed on the accumulators so that the conv output itself never exists in HBM:
 *     v = acc * scale[co] + shift[co]          (scale NULL = 1, shift NULL = 0; one of them non-NULL switches this on)
 *     v += old y                               (when `accumulate`)
 *     v += res * res_scale[co] + res_shift[co] (res.ptr non-NULL; res_scale NULL = plain residual)
 *     relu: bits = (v > 0) -> relu_bits (optional, layout of sfk_bn_apply), v = max(v, 0)
 *     y = v
 * res / relu / relu_bits need 16-byte channel groups (cout % 8 == 0 in bf16) and cout > 16; scale / shift work everywhere.
 * Not combinable with stats, bnb or out_relu_bits. */
typedef struct {
  const float *scale, *shift;
  sfk_fmap res; /* ptr == NULL: none */
  const float *res_scale, *res_shift;
  int32_t relu;
  int32_t reserved;
  uint8_t* relu_bits;
} sfk_conv_epilogue;

typedef struct {
  /* = sizeof(sfk_conv_desc) as the CALLER was compiled: a binding built against another layout of this struct is rejected
   * with SFK_ERR_INVALID by every entry point that takes it, instead of reading past (or short of) what it was given */
  uint32_t struct_size;
  uint32_t reserved0;
  sfk_fmap x, y;
  int32_t rt, rh, rw;
  int32_t gs[3], os[3], oo[3];
  int32_t ntaps;
  sfk_tap taps[SFK_MAX_TAPS];
  const void* w; /* [cout][wtaps][cin], dtype of x */
  int32_t wtaps, cin, cout;
  int32_t accumulate;
  float* stats;
  sfk_bn_bwd_fuse bnb;
  /* optional: Y is the gradient w.r.t. a ReLU output whose mask sfk_bn_apply left as a bitmap (relu_bits: byte
   * [pixel][co / V]); the pass then stores  result * mask  (after the += of `accumulate`), i.e. dz instead of dA, so the
   * BatchNorm backward of that unit neither re-applies the mask nor rewrites the tensor.  The pass must cover every
   * pixel of y in row order (os = 1, oo = 0, row extents = y extents); 8-byte aligned; supported when
   * sfk_conv_relu_out_supported(d). */
  const uint8_t* out_relu_bits;
  sfk_conv_epilogue ep;
} sfk_conv_desc;

int sfk_conv_igemm(const sfk_conv_desc* d, sfk_stream_t stream);
/* rows of d->stats / d->bnb.partials; <0 on error.  The kernel family, hence the row count, can depend on the epilogue
 * fields: for bnb / out_relu_bits ask with the descriptor as it will be launched (`stats` rows do not depend on them). */
int sfk_conv_igemm_mtiles(const sfk_conv_desc* d);
int sfk_conv_bnb_supported(const sfk_conv_desc* d);  /* 1 if d (ignoring d->bnb) can run with the bnb fusion, else 0 */
int sfk_conv_relu_out_supported(const sfk_conv_desc* d); /* 1 if d (ignoring out_relu_bits) can apply a bitmap, else 0 */
int sfk_conv_epilogue_supported(const sfk_conv_desc* d); /* 1 if d's ep (as filled in) can run, else 0 */
/* which kernel family sfk_conv_igemm runs for d (for tests / reports; never changes results beyond fp32 summation order):
 * 0 register-staged implicit GEMM, 1 LDS-DMA implicit GEMM, 3 the streaming pointwise kernel with the fused output transform
 * (conv_pw.hip: filter resident in LDS, no activation staging), 4 the deep-pipelined 256 x 256 tile (conv_igemm_p8.hip),
 * 5 the LDS-band 3 x 3 kernel (conv_halo.hip);
 * < 0: invalid descriptor */
int sfk_conv_igemm_family(const sfk_conv_desc* d);

/* ---------------------------------------------------------------------------------------------------------
 * sfk_conv_wgrad -- Conv3d filter gradient (autograd of the same nn.Conv3d modules, train.py:230).
 *   dw[co][widx][ci] += sum_rows dY[n,rt,rh,rw,co] * X[n, rt*gs[0]+dt, rh*gs[1]+dh, rw*gs[2]+dw, ci]
 * rows = the pixels of dy (dy.t/h/w are the row extents).  dw is fp32 and is ACCUMULATED into (zero it once per
 * step with sfk_fill_zero): through `workspace` when given, else with float atomics.  Same alignment rules as sfk_conv_igemm.
 */
typedef struct {
  uint32_t struct_size; /* = sizeof(sfk_wgrad_desc), as sfk_conv_desc.struct_size */
  uint32_t reserved0;
  sfk_fmap x, dy;
  int32_t gs[3];
  int32_t ntaps;
  sfk_tap taps[SFK_MAX_TAPS];
  float* dw; /* [cout][wtaps][cin] fp32 */
  int32_t wtaps, cin, cout;
  /* optional scratch of >= sfk_conv_wgrad_workspace_bytes(d) bytes (16-byte aligned): the pixel splits then store their
   * partial tiles with plain 16-byte stores and a second kernel adds them into dw in split order -- deterministic, and
   * it avoids re-adding every tile through fp32 atomics (~1.3 TB/s chip-wide).  NULL / too small: atomics. */
  float* workspace;
  int64_t workspace_bytes;
  /* optional FUSED DATA GRADIENT of the same dY (dg_w == NULL: off): dg_y[pixel][ci] = sum_co dY[pixel][co] * dg_w[ci][co],
   * dg_w a [cin][cout] matrix in the maps' dtype, dg_y a map of x's extents with cin channels (written, not accumulated).
   * For the fused block tail (sfk_bn_tail_*): R = dz^T a and the first data-gradient pass dz (A W)^T read dz ONCE -- the waves
   * of the filter-gradient tile whose columns lie past taps x cin compute it from the dY rows already staged in LDS.
   * Only where sfk_conv_wgrad_dg_supported(d) != 0 (bf16, pointwise, cout = 256, cin = 64). */
  const void* dg_w;
  sfk_fmap dg_y;
} sfk_wgrad_desc;

int sfk_conv_wgrad(const sfk_wgrad_desc* d, sfk_stream_t stream);
int64_t sfk_conv_wgrad_workspace_bytes(const sfk_wgrad_desc* d); /* host-side query; <0 on error */
int sfk_conv_wgrad_dg_supported(const sfk_wgrad_desc* d);        /* the fused data gradient (dg_w / dg_y) can run */
/* 1 when d would run one of the 256-column tiles of the MFMA-bound layers, which sum their pixel splits ONLY through `workspace`
 * (sfk_conv_wgrad_workspace_bytes(d) bytes; without it the call still works, on the 128-column tile with atomics) */
int sfk_conv_wgrad_wants_workspace(const sfk_wgrad_desc* d);

/* ---------------------------------------------------------------------------------------------------------
 * sfk_stem_conv_fwd / sfk_stem_conv_wgrad -- the stem Conv3d (kt,7,7) stride (1,2,2) padding (kt/2,3,3), bias=False
 * (my_slowfast.py:63-65 via create_res_basic_stem; canonical fast stem (5,7,7)), computed directly from the clip:
 * the input patch of a 16x16 output tile is staged once in LDS and every MFMA operand is built from it (an implicit
 * GEMM would re-read K*2 B = 1.5 KB per output pixel).  Also replaces the views of _prepare_slowfast_data
 * (train.py:136-140) and PackPathway's frame gather ((deprecated)/(torchvideo)train.py:60-71): the clip is read in
 * place through element strides, frame(t) = t_index ? t_index[t] : t (t_len logical frames).
 * Filter layout ("stem layout", compute dtype for fwd, fp32 for the gradient):
 *     w[co][ ((f*cin + ci)*7 + kh)*8 + kw ],  kw = 7 and rows beyond kt*cin*7 are ZERO padding;
 *     row length kp = sfk_stem_kp(cin, kt).  The gradient kernel never writes the padding.
 * y / dy: channels-last (n, t_len, ho, wo, cout), cout % 4 == 0, cout <= 64.
 * stats: optional [sfk_stem_conv_tiles(...)][cout][2] BatchNorm partial sums, as sfk_conv_igemm.
 */
typedef struct {
  const void* src;
  int32_t src_dtype; /* SFK_F32 or SFK_BF16 */
  int64_t sn, sc, st, sh, sw; /* element strides of (n, ci, t, h, w) */
  int32_t cin, t_in, h_in, w_in;
  const int32_t* t_index; /* device, t_len entries, or NULL (then t_len is ignored and t_in frames are used) */
  int32_t t_len;
  int32_t kt;
} sfk_stem_src;

int sfk_stem_kp(int32_t cin, int32_t kt);
int sfk_stem_conv_tiles(const sfk_stem_src* s, const sfk_fmap* y);
int sfk_stem_conv_fwd(const sfk_stem_src* s, const void* w, const sfk_fmap* y, float* stats, sfk_stream_t stream);
int sfk_stem_conv_wgrad(const sfk_stem_src* s, const sfk_fmap* dy, float* dw, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * BatchNorm3d (eps, momentum; affine; running stats) -- nn.BatchNorm3d after every conv
 * (my_slowfast.py:143-146,215-222; pytorchvideo stems / bottlenecks), ReLU (nn.ReLU), the residual
 * add of pytorchvideo ResBlock and their autograd.  All per-channel vectors are fp32.
 */

/* Training forward, step 1: reduce `nparts` partial (sum, sumsq) rows -> batch mean / biased var; writes
 * mean, invstd, the fused scale = gamma*invstd and shift = beta - mean*scale, and updates
 * running_mean/var (unbiased var, momentum) and num_batches_tracked exactly as nn.BatchNorm3d does.
 * workspace: NULL, or [SFK_BN_FOLD_ROWS][c][2] floats of scratch: with it, more than 4096 rows (the stems of the metric
 * geometry leave 50,176) are first folded down by a grid of workgroups; up to that a 256-thread block per channel pair
 * folds the rows in the finalize launch itself.  Either way the sums are accumulated in double in a fixed order
 * (deterministic).  c must be even (SFK_ERR_UNSUPPORTED otherwise; feature maps have c % 4 == 0); `partials` and `workspace` must
 * be 16-byte aligned (SFK_ERR_INVALID otherwise: the rows are read as float4), likewise in sfk_bn_bwd_finalize. */
int sfk_bn_finalize(const float* partials, int32_t nparts, int32_t c, int64_t count, const float* gamma,
                    const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                    int64_t* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                    float* workspace, sfk_stream_t stream);

/* Eval forward: scale/shift from the running statistics. */
int sfk_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, int32_t c, float* scale, float* shift,
                       sfk_stream_t stream);

/* Partial (sum, sumsq) of a feature map when no conv epilogue produced them: partials[nparts][c][2];
 * returns nparts through *nparts_out (host), at most `max_parts`. */
int sfk_bn_stats(const sfk_fmap* y, float* partials, int32_t max_parts, int32_t* nparts_out,
                 sfk_stream_t stream);

/* a = act( y*scale + shift + shortcut ),  shortcut = 0 | res | res*res_scale + res_shift ; act = ReLU if relu.
 * (stem / bottleneck norm+act, ResBlock "x + branch2(x)" then ReLU.)
 * relu_bits (optional, needs relu): the ReLU mask as one bit per element, byte [pixel][c / V] holding the V sign bits of
 * channel group g (V = 8 for bf16, 4 for f32; bit i = (a[pixel][g*V + i] > 0)) -- 1/16 of a bf16 map.  The backward of
 * a block output reads these instead of the activation (sfk_bn_bwd_reduce).
 * out_sums (optional; plain ReLU apply without shortcut only): partial column sums of the output AS STORED,
 * out_sums[nparts][c][2] = (sum a, 0), nparts <= max_parts returned through *nparts_out (host) -- the g = 1^T a of the fused
 * block tail (sfk_bn_tail_fwd) for free while `a` is written. */
int sfk_bn_apply(const sfk_fmap* y, const float* scale, const float* shift, const sfk_fmap* res,
                 const float* res_scale, const float* res_shift, int32_t relu, const sfk_fmap* out,
                 uint8_t* relu_bits, float* out_sums, int32_t max_parts, int32_t* nparts_out, sfk_stream_t stream);

/* sfk_bn_finalize + sfk_bn_apply in ONE launch: the apply's first workgroups fold the partial rows (the same deterministic block
 * sums: results bit-identical to the two calls), every workgroup waits for their counter before it reads scale / shift.  The
 * step's finalize launches are 8 us kernels on dependent chains behind a dispatch gap each; as a prologue the fold runs under
 * the dispatch ramp of the consumer's own grid.  Arguments as the two calls (c = y->c; out_sums is not available here);
 * sync: SFK_BN_SYNC_INTS int32 of the caller's, ZERO before the first call and left zero by every call (one set per BatchNorm that
 * may be in flight at the same time).  Meant for c <= 512: the channel pairs are claimed through one counter. */
#define SFK_BN_SYNC_INTS 2144
int sfk_bn_finalize_apply(const float* partials, int32_t nparts, int64_t count, const float* gamma, const float* beta,
                          float eps, float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                          float* mean, float* invstd, float* workspace, int32_t* sync, const sfk_fmap* y, float* scale,
                          float* shift, const sfk_fmap* res, const float* res_scale, const float* res_shift, int32_t relu,
                          const sfk_fmap* out, uint8_t* relu_bits, sfk_stream_t stream);

/* Backward of a = act(bn(y) [+ shortcut]) given dA:
 *   dz = dA * mask,   mask = relu_bits (as written by sfk_bn_apply) if given, else (mask_src > 0) if mask_src,
 *                     else (y*scale+shift > 0) if relu, else 1
 *   reduce: partials[nparts][c][2] = (sum dz, sum dz*xhat), xhat = (y-mean)*invstd; if dz_out: dz_out = dz
 *   finalize: dgamma (+)= sum dz*xhat, dbeta (+)= sum dz; coef[c][3] = (gamma*invstd, sum dz/count, sum dz*xhat/count)
 *   apply: dy = coef0 * (dz - coef1 - xhat*coef2)
 */
/* y == NULL (needs relu_bits): only the mask is applied and partials = (sum dz, 0) -- the fused block tail's variant */
int sfk_bn_bwd_reduce(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean,
                      const float* invstd, const float* scale, const float* shift, int32_t relu,
                      const sfk_fmap* dz_out, float* partials, int32_t max_parts, int32_t* nparts_out,
                      const uint8_t* relu_bits, sfk_stream_t stream);
int sfk_bn_bwd_finalize(const float* partials, int32_t nparts, int32_t c, int64_t count, const float* gamma,
                        const float* invstd, float* dgamma, float* dbeta, float* coef, float* workspace,
                        sfk_stream_t stream); /* workspace: as sfk_bn_finalize */
int sfk_bn_bwd_apply(const sfk_fmap* da, const sfk_fmap* y, const sfk_fmap* mask_src, const float* mean,
                     const float* invstd, const float* scale, const float* shift, int32_t relu,
                     const float* coef, const sfk_fmap* dy, sfk_stream_t stream);
/* sfk_bn_bwd_finalize + sfk_bn_bwd_apply in one launch (see sfk_bn_finalize_apply; results bit-identical to the two calls). */
int sfk_bn_bwd_finalize_apply(const float* partials, int32_t nparts, int64_t count, const float* gamma, float* dgamma,
                              float* dbeta, float* coef, float* workspace, int32_t* sync, const sfk_fmap* da, const sfk_fmap* y,
                              const sfk_fmap* mask_src, const float* mean, const float* invstd, const float* scale,
                              const float* shift, int32_t relu, const sfk_fmap* dy, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * The block tail  a -> conv_c (1x1x1, bias=False) -> norm_c (BatchNorm3d) [-> + shortcut -> ReLU]  WITHOUT the conv output
 * in HBM, forward or backward (pytorchvideo bottleneck tail, built via model/my_slowfast.py:94-125; its autograd,
 * train.py:230).  y = a W^T never has to exist because everything BatchNorm needs from it is a small matrix product:
 *   with the Gram matrix  G = a^T a  (c x c), the column sums g = 1^T a and the pixel count n
 *     sum_pix y[co]    = W[co] . g                      sum_pix y[co]^2 = W[co] G W[co]^T
 *   and, for the backward given dz (gradient at the BatchNorm output after the ReLU mask), R = dz^T a (cout x c), s = 1^T dz:
 *     sum dz*y[co] = W[co] . R[co];  dy = A dz + B y + C per channel (A = gamma*invstd, B = -A*c2*invstd,
 *     C = A*(c2*invstd*mean - c1), c1 = s/n, c2 = invstd*(sum dz*y - mean*s)/n), hence
 *     dW = diag(A) R + diag(B) (W G) + C (x) g,     da = dz (diag(A) W) + a (W^T diag(B) W) + (C W).
 * G is one ordinary filter-gradient call, g comes from the pass that wrote `a`, R is another filter-gradient call:
 *     gram = sfk_conv_wgrad(x = a, dy = a)       -> [c][c]: G   (one staged tile serves both operands)
 *     a_sums = sfk_bn_apply(..., out_sums)       -> partial column sums of a; n = the pixel count
 *     r    = sfk_conv_wgrad(x = a, dy = dz)      -> [cout][c]: R;  s = sum dz comes as partial rows from the kernel that wrote
 *            dz (its ReLU-mask pass)
 * (Rounds 1-2 kept `a` widened by a constant-1 channel group so that one filter-gradient call delivered G, g and n; on the
 * fast pathway's 8 .. 32-channel maps that doubled .. 1.25x-ed every pass over `a`, and 128 + 8 columns fell off the 128-wide
 * tile.)
 * sfk_bn_tail_fwd: folds a_sums into g [c] (caller's scratch, kept for the backward); batch statistics / running-stat update /
 *   scale, shift as sfk_bn_finalize, from G, g and count = n (the variance is W Gc W^T over the CENTRED matrix
 *   Gc = G/n - (g/n)(g/n)^T formed in double: no E[y^2] - E[y]^2 cancellation); also
 *   leaves t = W G ([cout][c] fp32) for the backward and, when wd != NULL, the filter of the backward's first
 *   data-gradient pass  wd [c][cout] (w_dtype) = (A W)^T, A = gamma * invstd  -> pass 1: da = dz . wd (sfk_conv_igemm, plain).
 *   A needs the forward statistics only, so pass 1 does not wait for R: it runs beside the R filter-gradient call.
 *   w = the conv's filter [cout][c] in compute precision (w_dtype).
 * sfk_bn_tail_bwd: dgamma += , dbeta += , dw += (fp32 [cout][c]), and the operands of the second data-gradient pass
 *     m [c][c] (w_dtype) = W^T diag(B) W   -> pass 2: da += a . m + bias   (sfk_conv_igemm, accumulate + ep.shift)
 *     bias [c] fp32 = C W
 */
int sfk_bn_tail_fwd(const float* gram, const float* a_sums, int32_t a_nparts, int64_t count, float* g, int32_t c,
                    const void* w, int32_t w_dtype, int32_t cout,
                    const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                    float* running_var, int64_t* num_batches_tracked, float* mean, float* invstd, float* scale,
                    float* shift, float* t, void* wd, sfk_stream_t stream);
int sfk_bn_tail_bwd(const float* r, const float* dz_partials, int32_t nparts, const float* g, int64_t count, const float* t,
                    int32_t c, const void* w, int32_t w_dtype, int32_t cout, const float* gamma,
                    const float* mean, const float* invstd, float* dgamma, float* dbeta, float* dw, void* m,
                    float* bias, float* coef, sfk_stream_t stream);
/* r [cout][c]; dz_partials [nparts][cout][2], component 0 = partial sums of dz as the kernel that WROTE dz left them
 * (sfk_bn_bwd_reduce with y == NULL, or the data-gradient pass with bnb.y_bn.ptr == NULL + out_relu_bits); coef: [cout][4]
 * fp32 scratch */

/* Both data-gradient passes of that tail in ONE pass over the pixels, for the narrow maps of the fast pathway (bf16):
 *     y[pix][co] = sum_k x1[pix][k] w1[co][k] + sum_k x2[pix][k] w2[co][k] + bias[co]
 * i.e. da = dz (A W)^T + a m + bias with x1 = dz, w1 = wd of sfk_bn_tail_fwd, x2 = a, w2 = m and bias of sfk_bn_tail_bwd:
 * da is written once instead of written, re-read and re-written (96 instead of 128 bytes per pixel on fast res2, and one
 * streaming pass instead of two MFMA tiles that are all epilogue at 8 .. 16 output channels).  fp32 accumulation, one
 * rounding; all three maps share the pixel grid, w1 [y.c][x1.c], w2 [y.c][x2.c] row-major in the maps' dtype, bias fp32 or NULL.
 * sfk_conv_pw_dual_supported: 1 for the channel counts the kernel is built for ((x1.c, x2.c, y.c) = (32, 8, 8), (64, 16, 16)),
 * bf16, 16-byte aligned pixel records; everything else keeps the two sfk_conv_igemm passes.  w1 / w2 must be 4-byte aligned
 * (SFK_ERR_INVALID otherwise: they are read as packed bf16 pairs). */
int sfk_conv_pw_dual_supported(const sfk_fmap* x1, const sfk_fmap* x2, const sfk_fmap* y);
int sfk_conv_pw_dual(const sfk_fmap* x1, const void* w1, const sfk_fmap* x2, const void* w2, const float* bias,
                     const sfk_fmap* y, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * MaxPool3d (1,k,k)/(1,s,s)/(0,p,p) of the stems (my_slowfast.py:66-68).  `argmax` (uint8 per output
 * element: kh*k + kw of the first maximum in torch's scan order) makes the backward tie-exact. */
int sfk_maxpool_fwd(const sfk_fmap* x, const sfk_fmap* y, uint8_t* argmax, int32_t k, int32_t s, int32_t p,
                    sfk_stream_t stream);
int sfk_maxpool_bwd(const sfk_fmap* dy, const uint8_t* argmax, const sfk_fmap* dx, int32_t k, int32_t s,
                    int32_t p, sfk_stream_t stream);

/* The stem's  BatchNorm3d -> ReLU -> MaxPool3d  (my_slowfast.py:63-68) without the activation map in HBM.
 * sfk_bn_maxpool_fwd:  out = maxpool(relu(y * scale + shift) rounded to the map's dtype), argmax as sfk_maxpool_fwd --
 *   bit-identical to sfk_bn_apply(relu) followed by sfk_maxpool_fwd.
 * The backward never stores the gradient of the activation either: with da = sfk_maxpool_bwd(d_out, argmax) rebuilt on
 * the fly and dz = da * [y * scale + shift > 0],
 *   sfk_bn_maxpool_bwd_reduce: partials[nparts][c][2] = partial (sum dz, sum dz * x_hat)   (-> sfk_bn_bwd_finalize)
 *   sfk_bn_maxpool_bwd_apply : dy = coef0 * (dz - coef1 - x_hat * coef2)                  (coef of sfk_bn_bwd_finalize)
 * -- the results of sfk_maxpool_bwd + sfk_bn_bwd_reduce + sfk_bn_bwd_apply.  Backward: window (3, 2, 1) only
 * (d_out extents = pooled extents of y), SFK_ERR_INVALID otherwise. */
int sfk_bn_maxpool_fwd(const sfk_fmap* y, const float* scale, const float* shift, const sfk_fmap* out, uint8_t* argmax,
                       int32_t k, int32_t s, int32_t p, sfk_stream_t stream);
int sfk_bn_maxpool_bwd_reduce(const sfk_fmap* d_out, const uint8_t* argmax, const sfk_fmap* y, const float* mean,
                              const float* invstd, const float* scale, const float* shift, float* partials,
                              int32_t max_parts, int32_t* nparts_out, sfk_stream_t stream);
int sfk_bn_maxpool_bwd_apply(const sfk_fmap* d_out, const uint8_t* argmax, const sfk_fmap* y, const float* mean,
                             const float* invstd, const float* scale, const float* shift, const float* coef,
                             const sfk_fmap* dy, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Head: AvgPool3d(kernel, stride 1) per pathway -> concat -> Dropout(p) -> Linear at every position ->
 * global mean (my_slowfast.py:75; pytorchvideo PoolConcatPathway + ResNetBasicHead).  By linearity
 *   feat[n, f_off + c] = (1/P) sum_{positions p} keep(n,c,p)/(1-rate) * mean_{window(p)} x[n, :, c]
 *   logits = feat @ W^T + b
 * keep() is a counter-based hash of (seed, n, feature index, p): stateless, so the backward regenerates it.
 * rate == 0 -> eval.  feat is fp32 [n][feat_ld].  seed is read from device memory (graph-replay safe). */
int sfk_head_pool_fwd(const sfk_fmap* x, int32_t kt, int32_t kh, int32_t kw, float rate,
                      const uint64_t* seed, float* feat, int32_t feat_ld, int32_t f_off, sfk_stream_t stream);
int sfk_head_pool_bwd(const float* dfeat, int32_t feat_ld, int32_t f_off, int32_t kt, int32_t kh, int32_t kw,
                      float rate, const uint64_t* seed, const sfk_fmap* dx, sfk_stream_t stream);
/* the keep mask itself, for tests: mask[n][c][P] uint8 */
int sfk_head_dropout_mask(int32_t n, int32_t c, int32_t f_off, int32_t positions, float rate,
                          const uint64_t* seed, uint8_t* mask, sfk_stream_t stream);

/* nn.Linear(2304, num_class) of the head, fp32.  logits[n][k] = feat[n] . w[k] + b[k] */
int sfk_fc_fwd(const float* feat, const float* w, const float* b, float* logits, int32_t n, int32_t f,
               int32_t k, sfk_stream_t stream);
/* dfeat = dlogits @ w ; dw += dlogits^T @ feat ; db += sum_n dlogits */
int sfk_fc_bwd(const float* dlogits, const float* feat, const float* w, float* dfeat, float* dw, float* db,
               int32_t n, int32_t f, int32_t k, sfk_stream_t stream);

/* nn.CrossEntropyLoss (mean) + argmax bookkeeping of train.py:228,239-242 in one kernel.
 * loss_sum[0] += mean CE of this batch; correct[0] += #(argmax == label); dlogits = (softmax - onehot)/n * gscale */
int sfk_softmax_ce(const float* logits, const int64_t* labels, int32_t n, int32_t k, float gscale,
                   float* dlogits, float* loss_out, float* loss_sum, int32_t* correct, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * optim.Adam(lr, betas, eps, weight_decay=0) (train.py:182,231) over one flat fp32 parameter arena.
 * step[0] is incremented on device first (bias correction uses it).  grad_scale multiplies g (1/world).
 * If shadow != NULL also writes the compute-precision copy of the updated parameters (bf16 or f32). */
int sfk_adam(float* p, const float* g, float* m, float* v, int64_t count, float lr, float beta1, float beta2,
             float eps, float grad_scale, int64_t* step, void* shadow, int32_t shadow_dtype,
             sfk_stream_t stream);

/* dst[ci][widx][co] = src[co][widx][ci]  (filters for the data-gradient pass), with dtype cast. */
int sfk_filter_transpose(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int32_t cout,
                         int32_t wtaps, int32_t cin, sfk_stream_t stream);
int sfk_cast(const void* src, int32_t src_dtype, void* dst, int32_t dst_dtype, int64_t count,
             sfk_stream_t stream);

/* The per-step refresh of every conv's compute-precision filters from the fp32 master arena (the weights Adam just
 * updated, train.py:231) in ONE launch: for each table entry, with i over [cout][wtaps][cin],
 *     s [off + i]                           = (dtype) master[off + i]                     (skipped when s  == NULL)
 *     st[off + (ci*wtaps + tap)*cout + co]  = (dtype) master[off + (co*wtaps + tap)*cin + ci]   (entries with
 *                                             transpose != 0; skipped when st == NULL)
 * `table` is DEVICE memory; entry e owns the wtaps * ceil(cout/32) * ceil(cin/32) blocks from first_block on (one
 * 32 x 32 (co, ci) tile of one tap each), ascending, and total_blocks is their sum. */
typedef struct {
  int64_t off;
  int32_t cout, wtaps, cin;
  int32_t first_block;
  int32_t transpose;
  int32_t reserved;
} sfk_filter_ent;
int sfk_filter_refresh(const float* master, void* s, void* st, int32_t dtype, const sfk_filter_ent* table,
                       int32_t n_layers, int32_t total_blocks, sfk_stream_t stream);
int sfk_fill_zero(void* p, size_t bytes, sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * Either side of the hot path.
 *
 * sfk_u8_normalize_crop -- the dataset's per-frame preprocessing on device, so clips cross PCIe as uint8:
 * transforms.ToTensor + Normalize(0.45, 0.225) (dataset/chalearn_dataset.py:41-46) and, for training,
 * RandomCrop(size, padding = size // 10) (:73-85), one crop offset per clip.
 *   src: uint8 [n][t][h][w][c] (the HWC `img_cat` frames of :113-114 stacked); lut: float[256] = the value
 *   ((u/255) - 0.45)/0.225 of every byte, computed by the caller in fp32 exactly as the reference transforms do;
 *   crop: device int32 [n][2] = (top, left) in [0, 2*pad], or NULL for no augmentation (test / valid clips);
 *   out[n][t][ch][y][x] = lut[src[n][t][y + top - pad][x + left - pad][ch]], 0 outside the frame (RandomCrop pads
 *   the normalised tensor with zeros); out dtype f32 (reference) or bf16.  The result is the (N,T,21,S,S) batch
 *   tensor train.py:125-145 slices, laid out as the stems read it. */
int sfk_u8_normalize_crop(const uint8_t* src, const float* lut, const int32_t* crop, int32_t pad, void* out,
                          int32_t out_dtype, int32_t n, int32_t t, int32_t c, int32_t h, int32_t w,
                          sfk_stream_t stream);

/* sfk_eval_aggregate -- Trainer.run_eval's score aggregation (train.py:337-362) on device: per clip softmax
 * (train.py:338; softmax = 0 keeps the scores as they are, as train_sparse.py:211-228 does), per video the mean over
 * its clips (rows seg_off[v] .. seg_off[v+1]-1), argmax (first maximum), compared with the label of its first clip.
 * ps_out (optional) [rows][c]; pred[v] = argmax or -1 for a video without clips; *correct += #(pred == label). */
int sfk_eval_aggregate(const float* logits, const int64_t* labels, const int32_t* seg_off, int32_t nvideos, int32_t c,
                       int32_t softmax, float* ps_out, int32_t* pred, int32_t* correct, sfk_stream_t stream);

/* SparseModel of the late-fusion ensemble (train_sparse.py:88-104): one nn.Linear(num_part, 1) per class,
 *   y[n][k] = b[k] + sum_p w[k][p] * x[n][p][k],   x = (N, P, C) stacked part scores, w = fcs.k.weight (1, P) rows.
 * bwd ACCUMULATES dw[k][p] += sum_n dy[n][k]*x[n][p][k], db[k] += sum_n dy[n][k] (CrossEntropyLoss / Adam of
 * train_sparse.py:157-158,171-176 are sfk_softmax_ce / sfk_adam). */
int sfk_sparse_fusion_fwd(const float* x, const float* w, const float* b, float* y, int32_t n, int32_t p, int32_t c,
                          sfk_stream_t stream);
int sfk_sparse_fusion_bwd(const float* x, const float* dy, float* dw, float* db, int32_t n, int32_t p, int32_t c,
                          sfk_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------
 * sfk_init -- the kernel-selection / tuning table ("an immutable kernel-selection table initialised
 * once").  Every field has a measured default (sfk_default_tuning); a caller that wants an A/B experiment fills the
 * struct and calls sfk_init ONCE per process before the first launch (calling it later is SFK_ERR_INVALID unless the
 * table is unchanged).  The fields only move work between equivalent kernels / grid shapes / cache hints: results keep
 * their meaning (summation order of split sums may change). */
typedef struct {
  uint32_t struct_size;       /* = sizeof(sfk_tuning) of the CALLER, set before sfk_default_tuning / sfk_get_tuning /
                                 sfk_init: a mismatch is SFK_ERR_INVALID and nothing is read or written                */
  int32_t igemm_short_k;      /* 5:    exact-count K loop for layers of <= this many K-steps                      */
  int32_t igemm_small_k;      /* 0:    128x128 instead of 256x128 tiles up to this K (cin * taps)                 */
  int32_t igemm_wide_store;   /* 1:    16-byte epilogue stores                                                    */
  int32_t wgrad_target_8w;    /* 192:  workgroup target (pixel splits x tiles) of the 8-wave filter-gradient tile  */
  int32_t wgrad_target_4w;    /* 256:  ... of the 4-wave tile                                                     */
  int32_t wgrad_use_workspace;/* 1:    use sfk_wgrad_desc.workspace when given                                    */
  int32_t wgrad_wide_co;      /* 7:    bit 0: one 256 x 64 tile for wide-output / narrow-input layers; bit 1: the LDS-DMA
                                         256 x 128 tile (half idle) when cout = 256 and taps x cin = 64; bit 2: Gram
                                         matrices (x and dy the same map) stage ONE tile for both operands              */
  int32_t bn_parts;           /* 1024: partial rows of the BatchNorm reductions (one resident generation)         */
  int32_t nt_apply_mb;        /* 0:    non-temporal loads+stores in sfk_bn_apply for maps >= this many MB (-1 off) */
  int32_t nt_reduce_mb;       /* 48:   non-temporal loads in sfk_bn_bwd_reduce                                    */
  int32_t nt_bwd_apply_mb;    /* 150:  non-temporal loads+stores in sfk_bn_bwd_apply                              */
  int32_t igemm_pw_stream;    /* 1:    streaming kernel for small-filter pointwise convs with a fused shortcut / ReLU     */
  int64_t pool_blocks;        /* 1<<20: grid cap of the pooling kernels (one pass per thread below it)            */
  int32_t igemm_tile256;      /* 3:    bit 0: 256 x 256 tile (one workgroup per CU) for MFMA-bound layers with 256 outputs;
                                         bit 1: 224 computed rows per tile where that fills the CUs better (M = 50,176)   */
  int32_t wgrad_target_gen;   /* 512:  ... of the register-staged filter-gradient kernels (0 = 1024); round 3: 768 -> 512,
                                         -0.15 ms per step on two boxes (fewer, longer workgroups on the side lanes)   */
  int32_t wgrad_target_256;   /* retired (ignored): the round-3 256-column ring tile; conv_wgrad_p8.hip serves those layers (wgrad_p8) */
  int32_t wgrad_min_stages_256; /* retired (ignored)                                                                          */
  int32_t igemm_p8;           /* 1:    bit 0: the deep-pipelined 256 x 256 conv tile (one workgroup per CU, 64-channel K-tiles, LDS-DMA
                                         in flight across the barriers, wave groups half a phase apart) for the MFMA-bound
                                         layers; bit 1: 224 computed rows per tile where that needs fewer row-generations   */
  int32_t wgrad_p8;           /* 16:   the same structure for the MFMA-bound filter gradients (pixel axis split over workgroups, partial
                                         tiles through the workspace): on for layers where a workgroup then still runs this many
                                         64-pixel K-tiles; 0 = off                                                         */
  int32_t igemm_halo;         /* 1:    the LDS-band kernel for the (1,3,3) stride-1 conv of slow res2 (64 -> 64, 56 x 56: conv_halo.hip)  */
  int32_t wgrad_band;         /* 3:    the LDS-band filter-gradient kernel (whole dW in the workgroups' accumulators, partials through the
                                         workspace: conv_wgrad_band.hip): bit 0: 64 -> 64 (1,3,3) over 56 x 56 frames (slow res2 conv_b);
                                         bit 1: 128 -> 128 over 28 x 28 (slow res3 conv_b)                                  */
  int32_t stem_v3;            /* 3:    bit 0: the input-frame-stationary forward of the canonical fast stem (filter in registers, one LDS
                                         pixel run per three MFMAs, half tiles with two workgroups per CU: stem_fwd_v4_kernel);
                                         bit 1: the same machinery for the canonical slow stem (kt = 1, 64 channels:
                                         stem_fwd_s4_kernel); bit 2 (EXPERIMENT): whole frames per workgroup in the fast stem's
                                         filter gradient                                                                   */
} sfk_tuning;
int sfk_default_tuning(sfk_tuning* out); /* out->struct_size must be set; fills every other field */
int sfk_init(const sfk_tuning* t);       /* NULL = defaults */
int sfk_get_tuning(sfk_tuning* out);     /* out->struct_size must be set */

int sfk_abi_version(void);
const char* sfk_status_string(int status);

#ifdef __cplusplus
}
#endif
#endif /* SFK_H */
