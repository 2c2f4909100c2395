// The column-sharded cascade behind the C ABI (include/wct_hip.h wct_stylize_sharded, wct_shard_geometry, wct_style_moments,
// wct_style_solve, wct_comm_attach_collectives).  NOT a stand-alone header: wct_api.hip includes it once, after its internal helpers
// (encode_impl, moments_impl, eig_impl, ...) and the split-level entry points -- one translation unit, no second copy of wct_ctx.
//
// What runs here is WCT.py:120-125 (levels 5 -> 1 of styleTransfer, WCT.py:98-106) on ONE rank's column strip; the geometry is the one
// wct_hip/sharded.py documents (and still implements over torch.distributed, as the checker of this path: bit-identical results).

namespace shard {
namespace {

// composite encode -> decode receptive field per side at level L (image columns), its cumulative form, and the ENCODER's alone
// (style strips: 70 / 30 / 10 / 4 / 1 columns rounded up to the level's pooling stride)
constexpr int LEVEL_HALO[6] = {0, 2, 10, 24, 72, 160};
constexpr int CUM_HALO[6] = {0, 6, 16, 40, 112, 272};
constexpr int STYLE_HALO[6] = {0, 1, 4, 12, 32, 80};
constexpr int AUTO_EXCHANGE_BELOW = 2560;

// owned column range of every rank: origins are multiples of 16, the last strip takes the remainder (sharded.py strip_bounds)
bool strip_bounds(int W, int world, std::vector<int>& xs) {
  xs.resize((size_t)world + 1);
  for (int r = 0; r < world; ++r) xs[r] = (int)std::min<long>(W, ((long)r * W / world) / 16 * 16);
  xs[world] = W;
  if (world > 1)
    for (int r = 0; r < world; ++r)
      if (xs[r + 1] - xs[r] < 16) return false;
  return true;
}

int resolve_halo_mode(int halo_mode, int world, const std::vector<int>& xs) {
  int narrowest = xs[1] - xs[0];
  for (int r = 1; r < world; ++r) narrowest = std::min(narrowest, xs[r + 1] - xs[r]);
  if (halo_mode == WCT_HALO_AUTO)
    return (world > 1 && narrowest < AUTO_EXCHANGE_BELOW && narrowest >= 2 * LEVEL_HALO[4]) ? WCT_HALO_EXCHANGE : WCT_HALO_RECOMPUTE;
  if (halo_mode == WCT_HALO_EXCHANGE && world > 1 && narrowest < 2 * LEVEL_HALO[4]) return -1;   // a neighbour must own what it is asked for
  return halo_mode;
}

// AUTO = OWNER: measured at 2, 4 and 8 ranks (profiles/r06_style_arrangement_per_level_join.txt) dealing the levels out whole beats cutting the style into
// strips by 0.5-1.3 ms per rank frame although rank 0 then carries 45.6 % of the style FLOPs: every style-side matrix square root occupies an XCD that the
// content lane's one-workgroup-per-CU kernels then wait for, and the all-reduce arrangements (strips, replicate) run five of them on EVERY rank.
int resolve_style_mode(int style_mode, int world, int Ws) {
  (void)Ws;
  if (world == 1) return WCT_STYLE_REPLICATE;
  return style_mode == WCT_STYLE_AUTO ? WCT_STYLE_OWNER : style_mode;
}

inline int feat_channels(const wct_ctx* ctx, int level) { return ctx->mod[WCT_KIND_ENC][level].layers.back().d.cout; }

// ---- RCCL as a wct_collectives table (user = the context)
int rccl_all_reduce(void* user, double* buf, size_t count, void* stream) {
  wct_ctx* ctx = static_cast<wct_ctx*>(user);
  return g_rccl.AllReduce(buf, buf, count, NCCL_FLOAT64, NCCL_SUM, ctx->comm, static_cast<hipStream_t>(stream));
}
int rccl_broadcast(void* user, void* buf, size_t bytes, int root, void* stream) {
  wct_ctx* ctx = static_cast<wct_ctx*>(user);
  return g_rccl.Broadcast(buf, buf, bytes, NCCL_INT8, root, ctx->comm, static_cast<hipStream_t>(stream));
}
int rccl_sendrecv(void* user, const wct_p2p* ops, int n, void* stream) {
  wct_ctx* ctx = static_cast<wct_ctx*>(user);
  hipStream_t st = static_cast<hipStream_t>(stream);
  int rc = g_rccl.GroupStart();
  for (int i = 0; i < n && rc == 0; ++i)
    rc = ops[i].is_send ? g_rccl.Send(ops[i].buf, ops[i].bytes, NCCL_INT8, ops[i].peer, ctx->comm, st)
                        : g_rccl.Recv(ops[i].buf, ops[i].bytes, NCCL_INT8, ops[i].peer, ctx->comm, st);
  const int re = g_rccl.GroupEnd();
  return rc ? rc : re;
}
void install_rccl(wct_ctx* ctx) {
  ctx->coll.user = ctx;
  ctx->coll.all_reduce_sum_f64 = rccl_all_reduce;
  ctx->coll.broadcast = rccl_broadcast;
  ctx->coll.sendrecv = rccl_sendrecv;
  ctx->coll_rccl = true;
}
const char* coll_error(const wct_ctx* ctx, int rc) {
  return (ctx->coll_rccl && g_rccl.GetErrorString) ? g_rccl.GetErrorString(rc) : "transport error";
}

#define COLLCHK(ctx, what, expr)                                                                                  \
  do {                                                                                                            \
    const int r__ = (expr);                                                                                       \
    if (r__ != 0) return fail(ctx, WCT_ERR_HIP, "%s: %s (code %d)", what, shard::coll_error(ctx, r__), r__);      \
  } while (0)

// style side of one level on a strip of the style image: encoder + raw moments over feature columns [x0, x1) on the style lane
int style_moments_impl(wct_ctx* ctx, int level, const float* strip, int Hs, int Wstrip, int x0, int x1, double* sum, double* sumsq) {
  Module& me = ctx->mod[WCT_KIND_ENC][level];
  if (!me.loaded) return fail(ctx, WCT_ERR_STATE, "encoder %d not loaded", level);
  const int C = me.layers.back().d.cout;
  int hs, ws;
  level_dims(level, Hs, Wstrip, hs, ws);
  if (x1 < 0) x1 = ws;
  Lane& ln = ctx->overlap ? ctx->side : ctx->main;
  if (l1_fused(ctx, level)) return l1_moments_impl(ctx, ln, level, strip, Hs, Wstrip, x0, x1, sum, sumsq);
  if (int rc = ensure(ctx, ctx->featS, (size_t)hs * ws * C * sizeof(float))) return rc;
  float* fS = reinterpret_cast<float*>(ctx->featS.p);
  if (int rc = encode_impl(ctx, ln, level, strip, Hs, Wstrip, fS, nullptr, nullptr)) return rc;
  return moments_impl(ctx, ln, fS, C, hs, ws, x0, x1, sum, sumsq);
}

// global style moments of one level -> cov_s^(1/2), mu_s (+ the style-side part of the fast fold) on the style lane; leaves ev_style[level]
int style_solve_impl(wct_ctx* ctx, int level, double n_s, const double* sum, const double* sumsq) {
  const int C = feat_channels(ctx, level);
  Lane& ln = ctx->overlap ? ctx->side : ctx->main;
  SumsView sv;
  if (int rc = sums_view(ctx, ln, sv)) return rc;
  if (int rc = eig_impl(ctx, ln, C, n_s, sum, sumsq, 0, ctx->eigS[level], sv.info + 1)) return rc;
  if (int rc = style_fold(ctx, level, ln.stream)) return rc;
  HIPCHK(ctx, hipEventRecord(ctx->ev_style[level], ln.stream));
  return WCT_OK;
}

struct Job {
  int world, rank, halo_mode, style_mode;
  bool bmap, exchange;
  std::vector<int> xs;          // content strip origins (+ W)
  std::vector<int> sxs;         // style strip origins (+ Ws), strips mode
  const int* halo;              // LEVEL_HALO or CUM_HALO
};

int check_job(wct_ctx* ctx, Job& j, int W_total, int Ws, int halo_mode, int style_mode, int flags) {
  j.world = ctx->comm_ranks;
  j.rank = ctx->comm_rank;
  if (!shard::strip_bounds(W_total, j.world, j.xs)) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: image width %d too small for %d strips", W_total, j.world);
  if (halo_mode < WCT_HALO_AUTO || halo_mode > WCT_HALO_EXCHANGE) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: bad halo_mode %d", halo_mode);
  j.halo_mode = resolve_halo_mode(halo_mode, j.world, j.xs);
  if (j.halo_mode < 0) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: WCT_HALO_EXCHANGE needs strips of at least %d columns", 2 * LEVEL_HALO[4]);
  j.exchange = j.halo_mode == WCT_HALO_EXCHANGE && j.world > 1;
  j.halo = j.halo_mode == WCT_HALO_EXCHANGE ? LEVEL_HALO : CUM_HALO;
  if (style_mode < WCT_STYLE_AUTO || style_mode > WCT_STYLE_REPLICATE) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: bad style_mode %d", style_mode);
  j.style_mode = resolve_style_mode(style_mode, j.world, Ws);
  if (j.style_mode == WCT_STYLE_STRIPS && !shard::strip_bounds(Ws, j.world, j.sxs))
    return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: style width %d too small for %d strips (use WCT_STYLE_REPLICATE)", Ws, j.world);
  j.bmap = (flags & WCT_SHARD_BROADCAST_MAP) != 0 && j.world > 1;
  if (ctx->shard_emulate && j.bmap)
    return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: the one-rank emulation of a job (debug key shard_emulate) has no peer to receive (M, b) from");
  return WCT_OK;
}

}  // namespace
}  // namespace shard

extern "C" {

int wct_comm_attach_collectives(wct_ctx* ctx, const wct_collectives* coll, int nranks, int rank) {
  if (!ctx) return WCT_ERR_INVALID;
  WCT_GUARD(ctx);
  if (!coll || !coll->all_reduce_sum_f64 || !coll->broadcast || !coll->sendrecv || nranks < 1 || rank < 0 || rank >= nranks)
    return fail(ctx, WCT_ERR_INVALID, "comm_attach_collectives: bad arguments (nranks %d, rank %d, all three functions are needed)", nranks, rank);
  if (ctx->comm || ctx->coll_set) return fail(ctx, WCT_ERR_STATE, "comm_attach_collectives: the context already has a communicator or transport");
  ctx->coll = *coll;
  ctx->coll_set = true;
  ctx->coll_rccl = false;
  ctx->comm_ranks = nranks;
  ctx->comm_rank = rank;
  return WCT_OK;
}

int wct_comm_info(const wct_ctx* ctx, int* nranks, int* rank) {
  if (!ctx) return WCT_ERR_INVALID;
  const bool has = ctx->comm || ctx->coll_set;
  if (nranks) *nranks = has ? ctx->comm_ranks : 0;
  if (rank) *rank = has ? ctx->comm_rank : 0;
  return WCT_OK;
}

// Known data through every function of the context's transport, between the job's actual ranks; synchronises (a set-up time call).
int wct_comm_selftest(wct_ctx* ctx) {
  if (!ctx) return WCT_ERR_INVALID;
  WCT_GUARD(ctx);
  if (!ctx->coll_set) return fail(ctx, WCT_ERR_STATE, "comm_selftest: no communicator (wct_comm_init / wct_comm_attach / wct_comm_attach_collectives)");
  const int n = ctx->comm_ranks, r = ctx->comm_rank, N = 1024;
  hipStream_t st = ctx->main.stream;
  wct_collectives& co = ctx->coll;
  struct Quiet { wct_ctx* c; explicit Quiet(wct_ctx* c_) : c(c_) { c->quiet_readback = true; } ~Quiet() { c->quiet_readback = false; } } quiet(ctx);   // ONE readback, at the end
  if (int rc = ensure(ctx, ctx->shStats, (size_t)6 * N * sizeof(double))) return rc;
  double* d = reinterpret_cast<double*>(ctx->shStats.p);
  std::vector<double> h((size_t)6 * N), back((size_t)6 * N);
  auto pat = [](int rank, int i) { return 1000.0 * (rank + 1) + i; };
  // block 0: all-reduce; 1: broadcast from the last rank; 2: ring send (to r + 1); 3: ring recv (from r - 1); 4 | 5: the cascade's
  // neighbour pattern -- one group with a send to and a receive from each neighbour that exists (4: from the left, 5: from the right)
  for (int i = 0; i < N; ++i) {
    h[i] = pat(r, i);
    h[N + i] = r == n - 1 ? pat(n - 1, i) + 0.5 : -1.0;
    h[2 * N + i] = pat(r, i) + 0.25;
    h[3 * N + i] = h[4 * N + i] = h[5 * N + i] = -1.0;
  }
  HIPCHK(ctx, hipMemcpyAsync(d, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice, st));
  COLLCHK(ctx, "selftest all-reduce", co.all_reduce_sum_f64(co.user, d, N, st));
  COLLCHK(ctx, "selftest broadcast", co.broadcast(co.user, d + N, N * sizeof(double), n - 1, st));
  {
    wct_p2p ring[2] = {{(r + 1) % n, 1, d + 2 * N, N * sizeof(double)}, {(r + n - 1) % n, 0, d + 3 * N, N * sizeof(double)}};
    COLLCHK(ctx, "selftest ring send / recv", co.sendrecv(co.user, ring, 2, st));
  }
  if (n > 1) {
    wct_p2p ops[4];
    int k = 0;
    if (r > 0) ops[k++] = wct_p2p{r - 1, 1, d + 2 * N, (size_t)(N / 2) * sizeof(double)};
    if (r + 1 < n) ops[k++] = wct_p2p{r + 1, 1, d + 2 * N + N / 2, (size_t)(N / 2) * sizeof(double)};
    if (r > 0) ops[k++] = wct_p2p{r - 1, 0, d + 4 * N, (size_t)(N / 2) * sizeof(double)};
    if (r + 1 < n) ops[k++] = wct_p2p{r + 1, 0, d + 5 * N, (size_t)(N / 2) * sizeof(double)};
    COLLCHK(ctx, "selftest neighbour send / recv", co.sendrecv(co.user, ops, k, st));
  }
  HIPCHK(ctx, hipMemcpyAsync(back.data(), d, back.size() * sizeof(double), hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  const int prev = (r + n - 1) % n;
  for (int i = 0; i < N; ++i) {
    double want = 0.0;
    for (int q = 0; q < n; ++q) want += pat(q, i);
    if (back[i] != want) return fail(ctx, WCT_ERR_HIP, "comm_selftest: all-reduce element %d is %.17g, expected %.17g", i, back[i], want);
    if (back[N + i] != pat(n - 1, i) + 0.5) return fail(ctx, WCT_ERR_HIP, "comm_selftest: broadcast element %d is %.17g", i, back[N + i]);
    if (back[3 * N + i] != pat(prev, i) + 0.25) return fail(ctx, WCT_ERR_HIP, "comm_selftest: ring receive element %d is %.17g (from rank %d)", i, back[3 * N + i], prev);
    if (n > 1 && i < N / 2) {
      // the left neighbour sent me ITS right half-block, the right neighbour its left half-block
      if (r > 0 && back[4 * N + i] != pat(r - 1, N / 2 + i) + 0.25) return fail(ctx, WCT_ERR_HIP, "comm_selftest: block from the left neighbour, element %d is %.17g", i, back[4 * N + i]);
      if (r + 1 < n && back[5 * N + i] != pat(r + 1, i) + 0.25) return fail(ctx, WCT_ERR_HIP, "comm_selftest: block from the right neighbour, element %d is %.17g", i, back[5 * N + i]);
    }
  }
  return WCT_OK;
}

int wct_shard_geometry(int W_total, int nranks, int rank, int halo_mode, int* own0, int* own1, int* in0, int* in1, int* halo_mode_resolved) {
  if (W_total < 1 || nranks < 1 || rank < 0 || rank >= nranks || halo_mode < WCT_HALO_AUTO || halo_mode > WCT_HALO_EXCHANGE) return WCT_ERR_INVALID;
  std::vector<int> xs;
  if (!shard::strip_bounds(W_total, nranks, xs)) return WCT_ERR_INVALID;
  const int mode = shard::resolve_halo_mode(halo_mode, nranks, xs);
  if (mode < 0) return WCT_ERR_INVALID;
  const int h5 = (mode == WCT_HALO_EXCHANGE ? shard::LEVEL_HALO : shard::CUM_HALO)[5];
  if (own0) *own0 = xs[rank];
  if (own1) *own1 = xs[rank + 1];
  if (in0) *in0 = std::max(0, xs[rank] - h5);
  if (in1) *in1 = std::min(W_total, xs[rank + 1] + h5);
  if (halo_mode_resolved) *halo_mode_resolved = mode;
  return WCT_OK;
}

int wct_style_moments(wct_ctx* ctx, int level, const float* style_strip, int Hs, int Ws_strip, int x0, int x1, double* sum, double* sumsq) {
  if (!ctx) return WCT_ERR_INVALID;
  WCT_GUARD(ctx);
  if (!valid_level(level) || !style_strip || !sum || !sumsq || Hs < 1 || Ws_strip < 1) return fail(ctx, WCT_ERR_INVALID, "style_moments: bad arguments");
  if (int rc = fork_side(ctx)) return rc;
  if (int rc = shard::style_moments_impl(ctx, level, style_strip, Hs, Ws_strip, x0, x1, sum, sumsq)) return rc;
  if (ctx->overlap) {     // the caller's stream (the collective it enqueues next) sees the sums
    HIPCHK(ctx, hipEventRecord(ctx->ev_join, ctx->side.stream));
    HIPCHK(ctx, hipStreamWaitEvent(ctx->main.stream, ctx->ev_join, 0));
  }
  return range_readback(ctx);
}

int wct_style_solve(wct_ctx* ctx, int level, double n_s, const double* sum_s, const double* sumsq_s) {
  if (!ctx) return WCT_ERR_INVALID;
  WCT_GUARD(ctx);
  if (!valid_level(level) || !sum_s || !sumsq_s) return fail(ctx, WCT_ERR_INVALID, "style_solve: bad arguments");
  if (!ctx->mod[WCT_KIND_ENC][level].loaded) return fail(ctx, WCT_ERR_STATE, "encoder %d not loaded", level);
  if (int rc = fork_side(ctx)) return rc;     // behind the collective that produced the sums on the caller's stream
  return shard::style_solve_impl(ctx, level, n_s, sum_s, sumsq_s);
}

int wct_stylize_sharded(wct_ctx* ctx, const float* content_ext, int H, int W_total, int in0, int in1, const float* style, int Hs, int Ws,
                        float alpha, int halo_mode, int style_mode, int flags, float* out_owned, int* Ho_out, int* Wo_out, double* range_total) {
  if (!ctx) return WCT_ERR_INVALID;
  WCT_GUARD(ctx);
  if (!ctx->coll_set) return fail(ctx, WCT_ERR_STATE, "stylize_sharded: no communicator (wct_comm_init / wct_comm_attach / wct_comm_attach_collectives)");
  if (!content_ext || !style || !out_owned || H < 16 || W_total < 16 || Hs < 16 || Ws < 16) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: bad arguments");
  for (int level = 1; level <= 5; ++level)
    if (!ctx->mod[WCT_KIND_ENC][level].loaded || !ctx->mod[WCT_KIND_DEC][level].loaded) return fail(ctx, WCT_ERR_STATE, "stylize_sharded: level %d not loaded", level);
  shard::Job j;
  if (int rc = shard::check_job(ctx, j, W_total, Ws, halo_mode, style_mode, flags)) return rc;
  const int world = j.world, rank = j.rank;
  int own0 = j.xs[rank], own1 = j.xs[rank + 1];
  {
    const int e0 = std::max(0, own0 - j.halo[5]), e1 = std::min(W_total, own1 + j.halo[5]);
    if (in0 != e0 || in1 != e1) return fail(ctx, WCT_ERR_INVALID, "stylize_sharded: rank %d of %d must be given content columns [%d, %d), got [%d, %d) (wct_shard_geometry)", rank, world, e0, e1, in0, in1);
  }
  hipStream_t st = ctx->main.stream;
  wct_collectives& co = ctx->coll;

  // ---- buffers (grow on first use of a size)
  const size_t img_floats = (size_t)3 * H * (in1 - in0);
  if (int rc = ensure(ctx, ctx->shIn, img_floats * sizeof(float))) return rc;
  if (int rc = ensure(ctx, ctx->shOut, img_floats * sizeof(float))) return rc;
  if (j.exchange) {
    if (int rc = ensure(ctx, ctx->shNext, img_floats * sizeof(float))) return rc;
    if (int rc = ensure(ctx, ctx->shEdge, (size_t)4 * 3 * H * shard::LEVEL_HALO[4] * sizeof(float))) return rc;   // sendL | sendR | recvL | recvR
  }
  // one all-reduce buffer PER LEVEL: [sum_c C | sumsq_c C*C | range flag | sum_s C | sumsq_s C*C] -- the style half (strips mode) is written by
  // the style lane while the content lane is still levels ahead, so the levels' buffers must not alias
  size_t npk = 0, pk_off[6] = {0, 0, 0, 0, 0, 0}, ntot = 0;
  for (int level = 5; level >= 1; --level) {
    const size_t C = (size_t)shard::feat_channels(ctx, level);
    npk = std::max(npk, C * C + C + 1);
    pk_off[level] = ntot;
    ntot += 2 * (C * C + C) + 1;
  }
  if (int rc = ensure(ctx, ctx->packed, ntot * sizeof(double))) return rc;
  double* pk_base = reinterpret_cast<double*>(ctx->packed.p);
  if (j.bmap)
    if (int rc = ensure(ctx, ctx->shMb, npk * sizeof(double))) return rc;

  // ---- style side
  if (int rc = fork_side(ctx)) return rc;
  Lane& sl = ctx->overlap ? ctx->side : ctx->main;
  auto owner = [&](int level) { return (5 - level) % world; };
  // strips mode: level L's style sums ride in level L's all-reduce, and the style lane takes the level's matrix square root right behind it.
  // Enqueue order on the style lane: moments of levels 5 and 4 up front, then per content level L: [wait for all-reduce L] solve L, moments of
  // level L - 2 -- the lane never waits for the content lane with strip work still undone, and the content lane waits for ONE level's sums at a time
  // (the first form joined all five levels into level 5's all-reduce: the content lane then stood behind the whole style side; measured slower than
  // dealing the levels out whole at every rank count, profiles/r06_style_arrangement_all_at_level5.txt)
  auto style_strip_moments = [&](int level) -> int {
    const int s0 = j.sxs[rank], s1 = j.sxs[rank + 1];
    const int sh = level - 1, hl = shard::STYLE_HALO[level];
    const int lo = std::max(0, s0 - hl), hi = std::min(Ws, s1 + hl), wstrip = hi - lo;
    if (int rc = ensure(ctx, ctx->shStyle, (size_t)3 * Hs * (std::min(Ws, s1 + shard::STYLE_HALO[5]) - std::max(0, s0 - shard::STYLE_HALO[5])) * sizeof(float))) return rc;
    float* strip = reinterpret_cast<float*>(ctx->shStyle.p);
    HIPCHK(ctx, launch_copy_block(style + lo, Ws, strip, wstrip, (long)3 * Hs, wstrip, sl.stream));
    const int f0 = (s0 - lo) >> sh, f1 = s1 >= Ws ? -1 : (s1 - lo) >> sh;
    const size_t C = (size_t)shard::feat_channels(ctx, level);
    double* ps = pk_base + pk_off[level] + C * C + C + 1;
    if (int rc = shard::style_moments_impl(ctx, level, strip, Hs, wstrip, f0, f1, ps, ps + C)) return rc;
    HIPCHK(ctx, hipEventRecord(ctx->ev_smom[level], sl.stream));
    return WCT_OK;
  };
  if (j.style_mode == WCT_STYLE_STRIPS) {
    if (int rc = style_strip_moments(5)) return rc;
    if (int rc = style_strip_moments(4)) return rc;
  } else {
    for (int level = 5; level >= 1; --level)
      if (j.style_mode == WCT_STYLE_REPLICATE || owner(level) == rank)
        if (int rc = style_side(ctx, level, style, Hs, Ws)) return rc;
  }

  // ---- content cascade
  const float* cur = content_ext;
  int lo = in0, hi = in1, Hc = H, Wc = W_total;     // cur holds image columns [lo, hi) of the current level's Hc x Wc image
  for (int level = 5; level >= 1; --level) {
    const int sh = level - 1;
    const int C = shard::feat_channels(ctx, level);
    const size_t cc = (size_t)C * C;
    double* pk = pk_base + pk_off[level];
    // crop the running image to this level's extended strip
    const int nlo = std::max(0, own0 - j.halo[level]), nhi = std::min(Wc, own1 + j.halo[level]);
    if (nlo < lo || nhi > hi) return fail(ctx, WCT_ERR_STATE, "stylize_sharded: level %d needs columns [%d, %d), the running strip holds [%d, %d)", level, nlo, nhi, lo, hi);
    const float* img = cur;
    if (nlo != lo || nhi != hi) {
      float* dst = reinterpret_cast<float*>(ctx->shIn.p);
      HIPCHK(ctx, launch_copy_block(cur + (nlo - lo), hi - lo, dst, nhi - nlo, (long)3 * Hc, nhi - nlo, st));
      img = dst;
    }
    lo = nlo; hi = nhi;
    const int Win = hi - lo;
    const int f0 = (own0 - lo) >> sh, f1 = own1 >= Wc ? -1 : (own1 - lo) >> sh;     // owned feature columns (last strip: to the floored end)
    int h = 0, w = 0;
    if (int rc = wct_content_encode(ctx, level, img, Hc, Win, f0, f1, pk, pk + C, &h, &w)) return rc;
    HIPCHK(ctx, launch_counter_to_f64(ctx->sat_dev, pk + C + cc, st));
    const double n_total = (double)h * (double)(Wc >> sh);                          // feature pixels of the whole image
    if (j.style_mode == WCT_STYLE_STRIPS) {
      // ONE all-reduce per level: the content sums, the range flag, and this level's style sums (contiguous behind them)
      HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_smom[level], 0));
      if (world > 1) COLLCHK(ctx, "all-reduce (content + style moments)", co.all_reduce_sum_f64(co.user, pk, 2 * (cc + C) + 1, st));
      HIPCHK(ctx, hipEventRecord(ctx->ev_sar[level], st));
      HIPCHK(ctx, hipStreamWaitEvent(sl.stream, ctx->ev_sar[level], 0));
      int hs, ws;
      level_dims(level, Hs, Ws, hs, ws);
      if (int rc = shard::style_solve_impl(ctx, level, (double)hs * ws, pk + cc + C + 1, pk + cc + C + 1 + C)) return rc;
      if (level - 2 >= 1)
        if (int rc = style_strip_moments(level - 2)) return rc;
    } else if (world > 1) {
      COLLCHK(ctx, "all-reduce (content moments)", co.all_reduce_sum_f64(co.user, pk, cc + C + 1, st));
    }
    if (range_total && level == 1) HIPCHK(ctx, hipMemcpyAsync(range_total, pk + C + cc, sizeof(double), hipMemcpyDeviceToDevice, st));   // the counter is cumulative
    if (j.style_mode == WCT_STYLE_OWNER && world > 1 && (!j.bmap || owner(level) != 0)) {
      // the level's style statistics travel from their owner to every solver (all ranks, or rank 0 alone with a broadcast map)
      size_t ns = 0;
      (void)wct_style_stats_count(ctx, level, &ns);
      if (int rc = ensure(ctx, ctx->shStats, ns * sizeof(double))) return rc;
      double* stats = reinterpret_cast<double*>(ctx->shStats.p);
      if (rank == owner(level) || ctx->shard_emulate) {
        // (emulation: a level this rank does not own "arrives" as whatever an earlier wct_style_prepare left in the context -- the
        //  receiver's import + fold run, the link does not)
        if (ctx->shard_emulate && !ctx->eigS[level].p) return fail(ctx, WCT_ERR_STATE, "stylize_sharded (emulated): run wct_style_prepare once first");
        if (int rc = wct_style_export(ctx, level, stats)) return rc;
      }
      if (!ctx->shard_emulate || rank == owner(level))
        COLLCHK(ctx, "broadcast (style statistics)", co.broadcast(co.user, stats, ns * sizeof(double), ctx->shard_emulate ? 0 : owner(level), st));
      if (rank != owner(level) && (!j.bmap || rank == 0))
        if (int rc = wct_style_import(ctx, level, stats)) return rc;
    }
    float* out = reinterpret_cast<float*>(ctx->shOut.p);
    int Ho = 0, Wo = 0;
    // WCT_SHARD_FAST_FOLD: the single-GPU cascade's fold (content_side): (W cov_s^1/2) from the style lane times cov_c^-1/2 from the content solve straight
    // into the decoder's first conv -- no T, M, b and no assemble launch on the critical path; where the decoder allows it (cin <= 128, f16x3 mode) and
    // every rank folds for itself.  fp32 round-off from the (M, b) form below (tests/test_hip_parity.py test_fast_fold_matches_the_map_based_fold).
    const bool fast = (flags & WCT_SHARD_FAST_FOLD) && !j.bmap && ctx->conv_mode == 1 && ctx->fold_ready[level] && fast_fold_level(ctx, level);
    if (fast) {
      SumsView sv;
      if (int rc = sums_view(ctx, ctx->main, sv)) return rc;
      if (int rc = eig_impl(ctx, ctx->main, C, n_total, pk, pk + C, 1, ctx->eigC, sv.info)) return rc;
      HIPCHK(ctx, hipStreamWaitEvent(st, ctx->ev_style[level], 0));
      ConvDesc first;
      if (int rc = fold_fast_impl(ctx, level, alpha, first)) return rc;
      if (l1_fused(ctx, level)) {
        if (int rc = l1_decode_impl(ctx, level, reinterpret_cast<const float*>(ctx->l1img.p), ctx->cur_H, ctx->cur_W, first, out)) return rc;
      } else {
        if (int rc = decode_impl(ctx, level, reinterpret_cast<float*>(ctx->featC.p), ctx->cur_h, ctx->cur_w, &first, out)) return rc;
      }
      Ho = ctx->cur_h << sh;
      Wo = ctx->cur_w << sh;
    } else {
      double *M, *b;
      if (j.bmap) {
        M = reinterpret_cast<double*>(ctx->shMb.p);
        b = M + cc;
        if (rank == 0)
          if (int rc = wct_content_solve(ctx, level, n_total, pk, pk + C, alpha, M, b)) return rc;
        COLLCHK(ctx, "broadcast (M, b)", co.broadcast(co.user, M, (cc + C) * sizeof(double), 0, st));
      } else {
        if (int rc = mb_view(ctx, &M, &b)) return rc;
        if (int rc = wct_content_solve(ctx, level, n_total, pk, pk + C, alpha, M, b)) return rc;
      }
      if (int rc = wct_content_decode(ctx, level, M, b, out, &Ho, &Wo)) return rc;
    }
    // floor-mode pooling may have dropped trailing columns / rows of the full image
    Wc = (Wc >> sh) << sh;
    Hc = Ho;
    hi = lo + Wo;
    own1 = std::min(own1, Wc);
    cur = out;
    if (j.exchange && level > 1) {
      // the decoded strip is exact on the owned columns only: the next level's margin comes from the neighbours' outermost owned columns
      const int halo = j.halo[level - 1], my_w = own1 - own0;
      auto width_of = [&](int r) { return std::min(j.xs[r + 1], Wc) - j.xs[r]; };
      const int left_w = rank > 0 ? std::min(halo, width_of(rank - 1)) : 0;
      const int right_w = rank + 1 < world ? std::min(halo, width_of(rank + 1)) : 0;
      const int send_w = std::min(halo, my_w);
      const long rows = (long)3 * Hc;
      float* edge = reinterpret_cast<float*>(ctx->shEdge.p);
      const size_t slot = (size_t)3 * H * shard::LEVEL_HALO[4];
      float *sendL = edge, *sendR = edge + slot, *recvL = edge + 2 * slot, *recvR = edge + 3 * slot;
      const float* owned = out + (own0 - lo);
      wct_p2p ops[4];
      int n = 0;
      {
        // both edge blocks packed in one launch
        CopySegs g{};
        if (rank > 0) { g.src[0] = owned; g.src_pitch[0] = Wo; g.dst[0] = sendL; g.dst_pitch[0] = send_w; g.width[0] = send_w; }
        if (rank + 1 < world) { g.src[1] = owned + (my_w - send_w); g.src_pitch[1] = Wo; g.dst[1] = sendR; g.dst_pitch[1] = send_w; g.width[1] = send_w; }
        HIPCHK(ctx, launch_copy_blocks(g, rows, st));
      }
      if (rank > 0) ops[n++] = wct_p2p{rank - 1, 1, sendL, (size_t)rows * send_w * sizeof(float)};
      if (rank + 1 < world) ops[n++] = wct_p2p{rank + 1, 1, sendR, (size_t)rows * send_w * sizeof(float)};
      if (left_w) ops[n++] = wct_p2p{rank - 1, 0, recvL, (size_t)rows * left_w * sizeof(float)};
      if (right_w) ops[n++] = wct_p2p{rank + 1, 0, recvR, (size_t)rows * right_w * sizeof(float)};
      if (ctx->shard_emulate)
        for (int i = 0; i < n; ++i) ops[i].peer = 0;      // every peer is this rank: recv k <- send k (equal widths)
      if (n) COLLCHK(ctx, "neighbour exchange (send / recv)", co.sendrecv(co.user, ops, n, st));
      float* next = reinterpret_cast<float*>(ctx->shNext.p);
      const int Wn = left_w + my_w + right_w;
      {
        // received left margin | owned columns | received right margin: the next level's input, one launch
        CopySegs g{};
        g.src[0] = recvL; g.src_pitch[0] = left_w; g.dst[0] = next; g.dst_pitch[0] = Wn; g.width[0] = left_w;
        g.src[1] = owned; g.src_pitch[1] = Wo; g.dst[1] = next + left_w; g.dst_pitch[1] = Wn; g.width[1] = my_w;
        g.src[2] = recvR; g.src_pitch[2] = right_w; g.dst[2] = next + left_w + my_w; g.dst_pitch[2] = Wn; g.width[2] = right_w;
        HIPCHK(ctx, launch_copy_blocks(g, rows, st));
      }
      cur = next;
      lo = own0 - left_w;
      hi = lo + Wn;
    }
  }
  const int Wo = own1 - own0;
  HIPCHK(ctx, launch_copy_block(cur + (own0 - lo), hi - lo, out_owned, Wo, (long)3 * Hc, Wo, st));
  if (Ho_out) *Ho_out = Hc;
  if (Wo_out) *Wo_out = Wo;
  ctx->quiet_readback = false;
  return range_readback(ctx);
}

}  // extern "C"
