/* libwct_hip -- C ABI of the MI355X-native WCT stylisation path.
 *
 * Drop-in boundary for ONE path of MingSun-Tse/Collaborative-Distillation: the 5-level encode ->
 * whitening/colouring transform -> decode cascade of `PytorchWCT/WCT.py --mode 16x|original`.
 * The reference has no FFI of its own; its boundary is the Python call surface of WCT.py / util_wct.py.
 * Each entry point below names the reference interface it replaces (paths relative to the reference
 * repository).  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every function returns 0 on success or a negative WCT_ERR_* code; wct_last_error() has the text.
 *     Nothing ever calls exit() (the reference does on a bad mode, util_wct.py:57-59).
 *   - all tensor arguments are DEVICE pointers to fp32 unless the name says host / f64; images are planar
 *     3 x H x W (the reference's NCHW with N = 1); feature maps are NHWC (layout 0, native) or NCHW
 *     (layout 1, the reference's layout) as selected by `layout`.
 *   - a context is bound to one device and one HIP stream; calls are asynchronous on that stream.
 *     One context per GPU / rank; a context is not thread-safe.
 *     One exception: with feature maps wider than 128 channels (--mode original) the outcome of each matrix-function iteration
 *     is read back (a slow global-memory Jacobi is the net under a non-converged one).  The cascade-type calls (wct_stylize*,
 *     wct_style_transfer_level, wct_style_prepare*) enqueue everything, synchronise ONCE at their end, look at all outcomes and
 *     repeat the call the synchronous way if one failed; the split-level calls (wct_solve, wct_content_solve, wct_transform)
 *     synchronise once per solve.  Calls of that mode therefore block the host and cannot be captured into a HIP graph; the 16x
 *     path never synchronises.
 *   - the caller owns every buffer it passes; the context owns weights and an internal workspace that
 *     grows on first use of a size (no allocation on later calls of the same or smaller size).
 */
#ifndef WCT_HIP_H
#define WCT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WCT_OK 0
#define WCT_ERR_INVALID (-1) /* bad argument (shape, level, layout, NULL) */
#define WCT_ERR_HIP (-2)     /* a HIP runtime call failed */
#define WCT_ERR_NOMEM (-3)   /* device allocation failed */
#define WCT_ERR_STATE (-4)   /* module for this level not loaded */
#define WCT_ERR_RANGE (-5)   /* an activation left the f16x3 range (+-65504) and was clamped: results deviate from fp32 */

#define WCT_KIND_ENC 0
#define WCT_KIND_DEC 1
#define WCT_LAYOUT_NHWC 0
#define WCT_LAYOUT_NCHW 1

typedef struct wct_ctx wct_ctx;

/* One `relu(conv3x3(reflect_pad(x)))` stage, in execution order.
 * weight: HOST fp32 OIHW [cout][cin][3][3]; bias: HOST fp32 [cout] -- the tensors of the reference's
 * state_dict (model/model_cd.py:712-718).  pool_after: MaxPool2d(2,2) follows (encoder);
 * up_after: UpsamplingNearest2d(2) follows (decoder). */
typedef struct wct_layer {
  int cin, cout;
  int pool_after, up_after;
  const float* weight;
  const float* bias;
} wct_layer;

/* per-kernel-family timing collected when profiling is enabled (bench.py's roofline leg) */
typedef struct wct_prof_entry {
  char name[48];     /* e.g. "conv3x3_f16x3<co=64,dma>", "enc_head_fused<3-16-16,pool>" */
  double ms;         /* summed HIP-event time */
  double flops;      /* algorithmic FLOPs of those launches */
  double bytes;      /* algorithmic HBM bytes (input read once + output written once + weights) */
  long launches;
} wct_prof_entry;

int wct_version(void);

/* replaces `wct = WCT(args).cuda()` (WCT.py:97, util_wct.py:31-59): create, then load 10 modules */
int wct_create(int device, wct_ctx** out);
void wct_destroy(wct_ctx* ctx);
const char* wct_last_error(const wct_ctx* ctx);
int wct_set_stream(wct_ctx* ctx, void* hip_stream); /* hipStream_t; NULL = default stream */
/* waits for both of the context's streams; returns WCT_ERR_RANGE -- ONCE, clearing the flag -- when an activation was clamped
 * since the last report (below), so a later WCT_ERR_RANGE always means a later clamp */
int wct_sync(wct_ctx* ctx);

/* Range check of the f16x3 arithmetic (wct_set_conv_mode 1, the default).  The reference computes its convolutions in fp32
 * (model/model_cd.py:724-743: plain nn.Conv2d); the split-f16 kernels represent an activation as hi + lo in f16 and clamp
 * it to +-65504.  A clamp that actually changed a value is a deviation from the reference, so every clamp site raises a
 * per-context device counter (threads that saw |x| >= 65504; saturating, never wraps).  A NaN in external data (an image, an
 * fp32 feature map handed to wct_decode*) is clamped to a finite value by the same instruction and raises the counter too;
 * non-finite weights are rejected by wct_load_module.  Three ways to see it:
 *   wct_saturation_count  synchronises the context and returns the counter (count may be NULL), clearing it when reset != 0
 *   wct_sync              reports a non-zero counter as WCT_ERR_RANGE once and clears it
 *   wct_range_poll        NO synchronisation: every compute entry point ends with an asynchronous 4-byte copy of the counter
 *                         to pinned host memory on the caller's stream; this returns the last copy that has landed, i.e. the
 *                         state after some COMPLETED call (callers check it at the start of their next call)
 *   wct_range_flag_f64    writes the counter as one double to flag_dev on the caller's stream, so that a sharded run can fold it
 *                         into the all-reduce of the moments it already makes and every rank sees every rank's clamps
 * Conv mode 0 (exact fp32 MFMA) has no such limit. */
int wct_saturation_count(wct_ctx* ctx, int reset, unsigned long long* count);
int wct_range_poll(wct_ctx* ctx, unsigned long long* count);
int wct_range_flag_f64(wct_ctx* ctx, double* flag_dev);

/* Context-level experiment switches (tests, A/B measurements): key in {"fuse", "sp", "l1fuse", "u8fuse", "upconv", "fastfold",
 * "interleave", "foldgemm"} with value 0 / 1, "in3wide" (2 / 1 / 0), "mom32" (0 / 1 / 2: fp64 / fp32-block products of the moments), "nscoop" (0 multi-launch, 1 single launch, 2 single launch with an injected
 * placement fault), "side_priority" (-1 / 0 / 1).  They select between kernel formulations of the same operators (fused
 * full-resolution ends, SP16 intermediates, level 1 without relu1_1 in HBM, f16x3 or exact-fp32 first conv of the un-pruned
 * encoders); results agree to fp32 round-off or bitwise (tests/test_hip_parity.py).  "eig_skip" (N: after N solves the matrix
 * functions are SKIPPED and stale results reused) is a timing experiment that produces wrong pictures by design: it is refused
 * unless the environment variable WCT_DEBUG is set; so is "shard_emulate" (100 * ranks + rank: a context holding a ONE-rank communicator runs
 * wct_stylize_sharded with the geometry of one rank of a larger job, its peers being itself -- what one rank of the job executes, with
 * other numbers; 0: off).  Environment variables WCT_* are honoured only when WCT_DEBUG is set. */
int wct_debug_set(wct_ctx* ctx, const char* key, double value);
/* Health counters of the context (no reference counterpart): "nscoop_solves" = single-launch C = 128 matrix-function solves enqueued,
 * "nscoop_aborts" = those that aborted into the Jacobi net (watchdog / placement; synchronises the context), "nscoop_off" = bit mask
 * of lanes (1 main, 2 side) that went back to the multi-launch schedule because at least three and at least a quarter of their
 * single-launch solves aborted.  An aborted solve is repaired (same results) but costs ~7 ms: this is how that shows. */
int wct_debug_get(wct_ctx* ctx, const char* key, double* value);

/* replaces SmallEncoder{L}_16x_aux(path) / SmallDecoder{L}_16x(path) / Encoder{L} / Decoder{L} construction
 * (util_wct.py:36-55; model_cd.py:712-718).  conv0_w [3*3] / conv0_b [3] (HOST): the encoder's 1x1 colour
 * affine (model_cd.py:725), folded into the first conv; NULL for decoders. */
int wct_load_module(wct_ctx* ctx, int kind, int level, int n_layers, const wct_layer* layers,
                    const float* conv0_w, const float* conv0_b);

/* shape of relu{level}_1 for an H x W image (floor pooling): C, h, w */
int wct_feature_shape(const wct_ctx* ctx, int level, int H, int W, int* C, int* h, int* w);

/* replaces `encoder(img)`  (WCT.py:100-101 -> model_cd.py:724-743) */
int wct_encode(wct_ctx* ctx, int level, const float* img, int H, int W, float* feat, int layout);
/* replaces `decoder(csF)`  (WCT.py:105 -> model_cd.py:276-294); image is 3 x (h<<(L-1)) x (w<<(L-1)) */
int wct_decode(wct_ctx* ctx, int level, const float* feat, int h, int w, int layout, float* img);

/* raw fp64 moments over the window rows [0,h) x cols [x0,x1) of an NHWC feature map of width w:
 *   sum[C], sumsq[C*C] (device, f64).  Replaces torch.mean + mm(cF, cF.t()) (util_wct.py:68-70, 94-96);
 * raw sums (not centred) so that a content-sharded run can all-reduce them across GPUs.
 * Arithmetic: maps of >= 65 536 pixels (h * w of the MAP, whatever the window) take their products and the sums of 64-pixel blocks in
 * fp32 on the matrix cores, block totals in fp64 (raw sums within ~1e-8 relative of the all-fp64 form); smaller maps -- the deep,
 * worst-conditioned levels -- fp64 products throughout.  Consequences: windows of ONE map share one arithmetic but cut the 64-pixel
 * blocks differently, so window sums add up to the whole map's to ~1e-8 relative, not to fp64 round-off; and a strip (+ halo) of a
 * sharded frame is its own, smaller map and may take the fp64 form where the untiled frame takes fp32 blocks (sharded and untiled
 * (M, b) agree to ~1e-8 either way; tests/test_sharded_gpu.py).  wct_debug_set("mom32", 0) selects exact fp64 products everywhere:
 * windows then add up to 1e-13. */
int wct_moments(wct_ctx* ctx, const float* feat, int C, int h, int w, int x0, int x1, double* sum, double* sumsq);

/* (n, sum, sumsq) of content and style -> csF = M cF + b.  M [C*C] row-major, b [C], device f64.
 * Replaces svd / pow / diag / mm of util_wct.py:74-125 and the blend of :219.  info (HOST, may be NULL)
 * receives, after an internal stream sync, how {content, style} were solved: n < 100 = Newton-Schulz iterations
 * (19 for the deflated iteration of C > 128), 100 + n = the Jacobi fallback ran n sweeps. */
int wct_solve(wct_ctx* ctx, int C, double n_c, const double* sum_c, const double* sumsq_c, double n_s,
              const double* sum_s, const double* sumsq_s, double alpha, double* M, double* b, int* info);

/* out = M feat + b per pixel (un-fused form).  Replaces mm(step2,cF), mm(S,.), + s_mean (util_wct.py:120-126) */
int wct_apply(wct_ctx* ctx, const float* feat, int C, int h, int w, int layout, const double* M, const double* b,
              float* out);

/* replaces `wct.transform(cF, sF, csF, alpha)` (util_wct.py:210-223); out has the shape/layout of cF */
int wct_transform(wct_ctx* ctx, const float* cF, int C, int h, int w, const float* sF, int hs, int ws,
                  float alpha, int layout, float* out);

/* decoder with csF = M feat + b folded into its first convolution (feat is NHWC) */
int wct_decode_affine(wct_ctx* ctx, int level, const float* feat, int h, int w, const double* M, const double* b,
                      float* img);

/* replaces styleTransfer(encoder, decoder, contentImg, styleImg, csF) (WCT.py:98-106) for one level */
int wct_style_transfer_level(wct_ctx* ctx, int level, const float* content, int H, int W, const float* style,
                             int Hs, int Ws, float alpha, float* out, int* Ho, int* Wo);

/* Split form of one level for content-sharded (multi-GPU) runs -- the same work as wct_style_transfer_level, cut where
 * a sharded run exchanges data (wct_hip/sharded.py):
 *   wct_style_prepare   style side of all loaded levels (encode, moments, cov^1/2), on the context's side stream
 *   wct_content_encode  cF = encoder(content) kept inside the context + raw moments over feature columns [x0,x1)
 *                       (x1 < 0: to the end) into sum[C], sumsq[C*C] (device f64) -> caller all-reduces them
 *   wct_content_solve   global (n, sum, sumsq) -> M [C*C], b [C] (device f64)  -> caller may broadcast them
 *   wct_content_decode  decoder(M cF + b) with M, b folded into the first conv
 * Together they replace styleTransfer() (WCT.py:98-106). */
int wct_style_prepare(wct_ctx* ctx, const float* style, int Hs, int Ws);
int wct_content_encode(wct_ctx* ctx, int level, const float* content, int H, int W, int x0, int x1, double* sum,
                       double* sumsq, int* h, int* w);
int wct_content_solve(wct_ctx* ctx, int level, double n_c, const double* sum_c, const double* sumsq_c, float alpha,
                      double* M, double* b);
int wct_content_decode(wct_ctx* ctx, int level, const double* M, const double* b, float* out, int* Ho, int* Wo);

/* The multi-GPU variant of one level BEHIND the boundary ("multi-GPU variant takes an ncclComm_t"; the reference is
 * single-GPU, WCT.py:97,110 -- nothing to replace but styleTransfer() itself, WCT.py:98-106, run on a column strip).
 *   wct_comm_load        dlopen the RCCL the host process uses and resolve ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllReduce /
 *                        ncclBroadcast / ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd.  A non-NULL path is THE library to use: if it
 *                        cannot be opened the call fails (no second RCCL is mapped behind the caller's back); NULL searches "librccl.so.1",
 *                        "librccl.so", /opt/rocm/lib/librccl.so.  libwct_hip.so itself links against no communication library;
 *                        single-GPU callers never load one.  wct_comm_library returns the path that was loaded ("" before).
 *   wct_comm_unique_id   ncclGetUniqueId into 128 caller bytes (rank 0; the caller ships them to the other ranks, e.g. through the
 *                        rendezvous it already has)
 *   wct_comm_init        ncclCommInitRank on the context's device: the context then OWNS a communicator over `nranks` contexts
 *                        (one per GPU / process); collective over the job -- every rank calls it
 *   wct_comm_attach      use a communicator the caller owns (`nccl_comm` is an ncclComm_t); not destroyed by the library
 *   wct_comm_destroy     ncclCommDestroy of an owned communicator (also done by wct_destroy)
 *   wct_level_sharded    one level of the column-sharded cascade in ONE call, everything on the context's stream, no host
 *                        synchronisation: encoder of `content` (3 x H x W: this rank's strip + halo) + raw moments over the OWNED feature
 *                        columns [x0, x1) (x1 < 0: to the end) -> ncclAllReduce(SUM, fp64, C*C + C + 1 values: sums, second moments, the
 *                        f16x3 range flag of wct_range_flag_f64) -> matrix functions with n_total = feature pixels of the WHOLE image,
 *                        against the level's style statistics (wct_style_prepare* on this rank, or wct_style_import of another rank's)
 *                        -> (M, b) folded into the decoder's first conv -> decoder -> out (3 x Ho x Wo).  range_total (device, one
 *                        double, may be NULL) receives the node-wide clamp count.  Same arithmetic in the same order as
 *                        wct_content_encode / all-reduce / wct_content_solve / wct_content_decode: bit-identical to that path.
 *                        With feature maps wider than 128 channels (--mode original) each solve reads one flag back (host synchronisation),
 *                        as in the split-level entries. */
int wct_comm_load(const char* librccl_path);
const char* wct_comm_library(void);
int wct_comm_unique_id(unsigned char* id128);
int wct_comm_init(wct_ctx* ctx, int nranks, int rank, const unsigned char* id128);
int wct_comm_attach(wct_ctx* ctx, void* nccl_comm, int nranks, int rank);
int wct_comm_destroy(wct_ctx* ctx);
int wct_level_sharded(wct_ctx* ctx, int level, const float* content, int H, int W, int x0, int x1, double n_total, float alpha,
                      float* out, int* Ho, int* Wo, double* range_total);

/* The WHOLE column-sharded cascade behind the boundary: WCT.py:120-125 (levels 5 -> 1 of styleTransfer, WCT.py:98-106) run by `nranks`
 * contexts -- one per GPU / process -- on column strips of ONE content image.  The reference is single-GPU (WCT.py:97,110); what
 * makes the path shardable is that every operator of it is local except the content (and style) statistics (util_wct.py:68-70,94-96).
 * Everything is enqueued on the context's stream(s): strip geometry and per-level crops, the style side, encoders, owned-column
 * moments, the collectives, matrix functions, fold, decoders, the neighbour exchange.  No host synchronisation, no allocation after the
 * first call of a size (--mode 16x; the wide models' C > 128 solves read one flag back per solve, as the split-level entries do).
 *
 * Collectives.  The cascade talks to its peers through a wct_collectives table (below).  wct_comm_init / wct_comm_attach install the
 * RCCL one (ncclAllReduce / ncclBroadcast / ncclGroupStart + ncclSend + ncclRecv + ncclGroupEnd on the context's communicator and
 * stream); wct_comm_attach_collectives installs a caller-supplied transport (another communication library; the test suite's
 * in-process world that moves device buffers between rank threads).  All functions are asynchronous on `stream` and return 0 or an
 * error code of the transport; buffers are device memory.
 *
 * Geometry (wct_shard_geometry, a pure function): rank r owns content columns [own0, own1) (origins are multiples of 16, so every
 * pooling grid coincides with the untiled image's; the last strip takes the remainder) and must be GIVEN columns [in0, in1) = its strip
 * plus the level-5 margin of the halo mode towards the image interior:
 *   WCT_HALO_EXCHANGE   every level runs on own +- (160, 72, 24, 10, 2) columns (level 5..1: the composite encode->decode receptive
 *                       field) and, between levels, neighbours exchange the (72, 24, 10, 2) outermost OWNED columns of the image just
 *                       decoded -- two messages per level boundary, 3 x H x margin floats (<= 3.5 MB at H = 4096);
 *   WCT_HALO_RECOMPUTE  cumulative margins (272, 112, 40, 16, 6), no exchange;
 *   WCT_HALO_AUTO       exchange for strips narrower than 2560 columns (and at least 144), recompute otherwise.
 * With those margins a strip's owned columns are bit-identical to the untiled level given the same (M, b).
 *
 * Style side (style_mode; the style image is given WHOLE to every rank, as the reference gives it to every level, WCT.py:121-125):
 *   WCT_STYLE_STRIPS     the style image is cut into column strips like the content: rank r encodes its strip + the ENCODER's
 *                        receptive field (80, 32, 12, 4, 1 columns at level 5..1) at every level, raw moments over its owned feature
 *                        columns; level L's style sums travel in level L's all-reduce together with the content moments, and every
 *                        rank takes the level's matrix square root on its side stream right behind it.  Per-rank style work = 1/nranks (+ margins);
 *   WCT_STYLE_OWNER      level L's style side is computed whole by rank (5 - L) mod nranks and broadcast (C*C + C doubles per level);
 *   WCT_STYLE_REPLICATE  every rank computes all five levels (no communication; tiny styles);
 *   WCT_STYLE_AUTO       = owner.  Measured at 2, 4 and 8 ranks (profiles/r06_style_arrangement_per_level_join.txt): owner beats strips by 0.5-1.3 ms per
 *                        rank frame although its rank 0 carries 45.6 % of the style FLOPs -- each style-side matrix square root occupies an XCD the
 *                        content kernels then wait for, and strips / replicate run five of them on every rank, owner at most three.
 * WCT_SHARD_BROADCAST_MAP: rank 0 alone solves for the colouring map and broadcasts (M [C*C], b [C]) doubles per level; default: every
 * rank solves for itself (the all-reduce returns identical bits everywhere and the solver is deterministic).
 *
 * Per level: ONE all-reduce of [sum C | sumsq C*C | f16x3 range flag] doubles (+ the level's style sums in strips mode), the optional
 * broadcasts above, and in exchange mode one grouped send/recv pair per neighbour.
 *
 *   content_ext   columns [in0, in1) of the content, planar 3 x H x (in1 - in0)
 *   style         the whole style image, planar 3 x Hs x Ws
 *   out_owned     receives this rank's owned columns of the result, planar 3 x Ho x Wo (Ho = 16 floor(H / 16); Wo = own1 - own0,
 *                 less what floor pooling cut off the last strip); must hold 3 * H * (own1 - own0) floats
 *   range_total   device, one double, may be NULL: the node-wide count of f16x3 clamps as of the last all-reduce (wct_range_flag_f64)
 * Results are those of the split-level entries driven by wct_hip/sharded.py over the same transport, bit for bit. */
typedef struct wct_p2p {
  int peer;       /* rank */
  int is_send;    /* 1: send `bytes` from buf to peer; 0: receive `bytes` from peer into buf */
  void* buf;
  size_t bytes;
} wct_p2p;
typedef struct wct_collectives {
  void* user;
  int (*all_reduce_sum_f64)(void* user, double* buf, size_t count, void* hip_stream);       /* in place, every rank */
  int (*broadcast)(void* user, void* buf, size_t bytes, int root, void* hip_stream);        /* in place, every rank */
  int (*sendrecv)(void* user, const wct_p2p* ops, int n_ops, void* hip_stream);             /* ONE group: all ops posted together */
} wct_collectives;
#define WCT_HALO_AUTO 0
#define WCT_HALO_RECOMPUTE 1
#define WCT_HALO_EXCHANGE 2
#define WCT_STYLE_AUTO 0
#define WCT_STYLE_OWNER 1
#define WCT_STYLE_STRIPS 2
#define WCT_STYLE_REPLICATE 3
#define WCT_SHARD_BROADCAST_MAP 1   /* flags */
#define WCT_SHARD_FAST_FOLD 2       /* the single-GPU cascade's fold -- (W cov_s^1/2) cov_c^-1/2 straight into the decoder's first conv, no (M, b) on the
                                       critical path -- where the decoder allows it and every rank folds for itself; fp32 round-off from the (M, b) form,
                                       which stays the default because wct_hip/sharded.py's split-level orchestration is bit-identical to it */
int wct_comm_attach_collectives(wct_ctx* ctx, const wct_collectives* coll, int nranks, int rank);
/* nranks / rank of the communicator or transport the context holds (0 / 0: none) */
int wct_comm_info(const wct_ctx* ctx, int* nranks, int* rank);
/* Known data through all three functions of the context's transport between the job's ranks (all-reduce; broadcast from the last rank;
 * a ring send / recv; the cascade's grouped neighbour exchange), checked on the host: 0, or WCT_ERR_HIP naming what came back wrong.
 * Collective over the job (every rank calls it), synchronises the stream: a set-up time check (bench.py runs it before timing). */
int wct_comm_selftest(wct_ctx* ctx);
/* halo_mode may be WCT_HALO_AUTO; *halo_mode_resolved (may be NULL) receives what it resolves to for this width and rank count */
int wct_shard_geometry(int W_total, int nranks, int rank, int halo_mode, int* own0, int* own1, int* in0, int* in1, int* halo_mode_resolved);
int wct_stylize_sharded(wct_ctx* ctx, const float* content_ext, int H, int W_total, int in0, int in1, const float* style, int Hs, int Ws,
                        float alpha, int halo_mode, int style_mode, int flags, float* out_owned, int* Ho, int* Wo, double* range_total);
/* The two halves of WCT_STYLE_STRIPS for callers that run the collectives themselves (wct_hip/sharded.py over torch.distributed):
 *   wct_style_moments  encoder of `style_strip` (3 x Hs x Ws_strip: the rank's style columns + margins) + raw moments over feature columns
 *                      [x0, x1) (x1 < 0: to the end) into sum[C], sumsq[C*C] (device f64) on the context's SIDE stream; the caller's
 *                      stream is made to wait for them (so that a collective enqueued next sees them)
 *   wct_style_solve    global (n_s, sum, sumsq) of the style feature map -> cov_s^(1/2), mu_s inside the context (what
 *                      wct_style_prepare leaves for the level), on the side stream behind everything enqueued on the caller's */
int wct_style_moments(wct_ctx* ctx, int level, const float* style_strip, int Hs, int Ws_strip, int x0, int x1, double* sum, double* sumsq);
int wct_style_solve(wct_ctx* ctx, int level, double n_s, const double* sum_s, const double* sumsq_s);

/* replaces the cascade of WCT.py:120-125 (levels 5..1, num_run times).  out must hold 3*H*W floats. */
int wct_stylize(wct_ctx* ctx, const float* content, int H, int W, const float* style, int Hs, int Ws, float alpha,
                int num_run, float* out, int* Ho, int* Wo);

/* Style statistics cache.  wct_style_prepare leaves, per level, the style mean and cov^(1/2) inside the
 * context; they depend only on the style image, so
 *   wct_stylize_prepared  runs the content cascade against them (no style-side work): content x style batches pay
 *                         the style side once per style, not once per pair (data_loader.py:32-36 builds the product)
 *   wct_style_export / wct_style_import  move them (device f64: C*C matrix, then C means; wct_style_stats_count
 *                         doubles) so that the GPUs of a node compute each level ONCE and broadcast it
 *                         (wct_hip/sharded.py) instead of every rank repeating all five.
 * wct_style_prepare_levels: like wct_style_prepare for the levels in `level_mask` (bit L set = level L). */
int wct_style_prepare_levels(wct_ctx* ctx, const float* style, int Hs, int Ws, unsigned level_mask);
int wct_style_stats_count(const wct_ctx* ctx, int level, size_t* n_doubles);
int wct_style_export(wct_ctx* ctx, int level, double* stats);
int wct_style_import(wct_ctx* ctx, int level, const double* stats);
int wct_stylize_prepared(wct_ctx* ctx, const float* content, int H, int W, float alpha, int num_run, float* out, int* Ho,
                         int* Wo);

/* Image edge: the reference harness's ToTensor (PytorchWCT/data_loader.py:57-58: uint8 HWC -> fp32 CHW
 * / 255) and save_image (WCT.py:128; torchvision 0.2.1: mul(255).clamp(0,255).byte()) on the device.  Pointers are
 * device pointers, uint8 images are H x W x 3 interleaved and 4-byte aligned.  round_mode 0 = truncation (the
 * reference's pinned torchvision), 1 = +0.5 before truncation (torchvision >= 0.4). */
int wct_u8_to_planar(wct_ctx* ctx, const uint8_t* hwc, int H, int W, float* planar);
int wct_planar_to_u8(wct_ctx* ctx, const float* planar, int H, int W, uint8_t* hwc, int round_mode);
/* ToTensor -> wct_stylize -> save_image conversion in one call; out_hwc must hold H*W*3 bytes */
int wct_stylize_u8(wct_ctx* ctx, const uint8_t* content_hwc, int H, int W, const uint8_t* style_hwc, int Hs, int Ws,
                   float alpha, int num_run, uint8_t* out_hwc, int* Ho, int* Wo, int round_mode);

/* transforms.Resize(size) of the reference harness (PytorchWCT/data_loader.py:52-56: torchvision 0.2.1 functional.resize ->
 * PIL Image.resize((ow, oh), Image.BILINEAR)) on the device, bit-exact with Pillow's resampler (requirements.txt: Pillow==8.2.0;
 * libImaging/Resample.c: separable, horizontal pass first, antialiased triangle filter, 22-bit fixed-point weights).
 *   wct_resize_shape         torchvision's size rule: the smaller edge becomes `size` (0 or already equal: unchanged)
 *   wct_resize_u8            uint8 HWC -> uint8 HWC of oH x oW (any target size; equal sizes copy)
 *   wct_resize_u8_to_planar  the same followed by ToTensor (data_loader.py:57: planar fp32 / 255) without a uint8 round trip
 * Pointers are device pointers (any alignment: the kernels read single bytes); H and oH <= 65535 (the row index is a grid
 * dimension).  Weight tables per (input, output) size are built on the host on first use (O(W + H)) and cached in the context
 * (16 axes, least recently used evicted). */
int wct_resize_shape(int H, int W, int size, int* oH, int* oW);
int wct_resize_u8(wct_ctx* ctx, const uint8_t* src_hwc, int H, int W, uint8_t* dst_hwc, int oH, int oW);
int wct_resize_u8_to_planar(wct_ctx* ctx, const uint8_t* src_hwc, int H, int W, float* planar, int oH, int oW);

/* bytes of internal workspace a wct_stylize of this size will hold; wct_reserve allocates it up front */
size_t wct_workspace_bytes(const wct_ctx* ctx, int H, int W, int Hs, int Ws);
int wct_reserve(wct_ctx* ctx, int H, int W, int Hs, int Ws);

/* arithmetic of the 3x3 convolutions (all but the HBM-bound 3-channel first conv):
 *   1 (default)  split-f16 "f16x3" MFMA: x = hi + lo in f16, w.x ~ hi.hi + hi.lo + lo.hi, fp32 accumulate --
 *                fp32-class accuracy (6e-7 vs 8e-7 for exact fp32 on K = 1152 dot products) at 3/16 of the issue time;
 *                operands are ~22-bit (two f16 terms), the lo.lo product is dropped, accumulation is fp32, activations
 *                beyond +-65504 are clamped and flagged (wct_saturation_count)
 *   0            exact fp32 MFMA (v_mfma_f32_16x16x4_f32)
 * (env WCT_CONV_MODE=0|fp32 selects 0 at wct_create when WCT_DEBUG is set.) */
int wct_set_conv_mode(wct_ctx* ctx, int mode);

/* `--numpy` of the reference (WCT.py:34, util_wct.py:204-208): whiten_and_color_np adds the identity to the CONTENT
 * covariance before its SVD (util_wct.py:143) -- a different operator from the default path (max-abs 0.49 on a toy case),
 * otherwise the same steps.  0 (default): off. */
int wct_set_numpy_variant(wct_ctx* ctx, int on);

/* 1 (default): the style side of a level (encode, moments, eigen-decomposition -- independent of the content) runs on
 * a context-owned side stream and overlaps the content side; 0: everything runs in order on the caller's stream
 * (used when timing individual kernels). */
int wct_set_overlap(wct_ctx* ctx, int on);

/* profiling: when enabled every kernel launch is bracketed by HIP events on the context's stream */
int wct_profile_enable(wct_ctx* ctx, int on);
int wct_profile_reset(wct_ctx* ctx);
int wct_profile_read(wct_ctx* ctx, wct_prof_entry* entries, int max_entries, int* n_entries);

#ifdef __cplusplus
}
#endif
#endif /* WCT_HIP_H */
