// solver_host.hip — the host-pointer drop-in entry points: c_lqp_solve / c_lcone_solve exactly as PMPC.jl/pmpcjl/module.cpp:9-23 declares them, and their
// extensions (row-major blocks, worst-k, smooth_cstr).  Split out of solver.hip (r05); shared declarations: solver_internal.h.
#include "solver_internal.h"

extern "C" {

// -------------------------------------------------------------------------------------------------
// host-pointer drop-in entry points
// -------------------------------------------------------------------------------------------------
static pmpc_ctx *g_ctx = nullptr;

static bool any_nan(const double *p, size_t n) {
  if (!p) return true;
  for (size_t k = 0; k < n; k++)
    if (p[k] != p[k]) return true;
  return false;
}

// Pageable host arrays -> HBM: hipMemcpyAsync from pageable memory stages through a single-threaded copy (~7 GB/s measured,
// 100 ms for config D's 700 MB).  Here worker threads copy 8 MB chunks into a pinned bounce buffer and hand each one to the
// copy engine as soon as it is staged (chunk order is irrelevant: the solve is enqueued behind all of them).
// The bounce buffer keeps one slot per chunk and outlives the call, and so do the device staging buffers: a chunk whose
// bytes equal what its slot holds from the previous call (memcmp: exact, no sampling) is neither copied nor sent again —
// inside an SCP loop that is Q, R, the references and the boxes, ~43 % of config D's 702 MB per call.
struct UploadItem { void *dst; const void *src; size_t bytes; };
static void upload_all(pmpc_ctx *c, const std::vector<UploadItem> &items) {
  constexpr size_t CH = 8u << 20;
  struct Chunk { char *dst; const char *src; size_t bytes, off; };
  std::vector<Chunk> chunks;
  size_t total = 0;
  for (const UploadItem &it : items)
    for (size_t o = 0; o < it.bytes; o += CH) {
      const size_t b = std::min(CH, it.bytes - o);
      chunks.push_back({(char *)it.dst + o, (const char *)it.src + o, b, total});
      total += (b + 255) & ~(size_t)255;
    }
  if (total > c->pinned_bytes) {
    c->staged.clear();
    if (c->pinned) HIP_WARN(hipHostFree(c->pinned));
    c->pinned = nullptr;
    c->pinned_bytes = 0;
    HIP_CHECK(hipHostMalloc(&c->pinned, total, hipHostMallocDefault));
    c->pinned_bytes = total;
  }
  const bool reuse_on = c->opt[OPT_HOST_REUSE] != 0.0;
  const std::vector<pmpc_ctx::StagedChunk> &prev = c->staged;
  unsigned nthreads = std::thread::hardware_concurrency();
  nthreads = std::max(1u, std::min(nthreads ? nthreads : 4u, 16u));
  if (chunks.size() < 4) nthreads = 1;
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};  // (an exception must not leave a worker thread)
  auto work = [&]() {
    (void)hipSetDevice(c->device);
    for (size_t k = next++; k < chunks.size(); k = next++) {
      const Chunk &ch = chunks[k];
      if (reuse_on && k < prev.size() && prev[k].dst == ch.dst && prev[k].bytes == ch.bytes && prev[k].off == ch.off &&
          memcmp((const char *)c->pinned + ch.off, ch.src, ch.bytes) == 0)
        continue;  // the device copy of the previous call is still current
      memcpy((char *)c->pinned + ch.off, ch.src, ch.bytes);
      if (hipMemcpyAsync(ch.dst, (char *)c->pinned + ch.off, ch.bytes, hipMemcpyHostToDevice, c->stream) != hipSuccess) failed = 1;
    }
  };
  std::vector<std::thread> pool;
  for (unsigned t = 1; t < nthreads; t++) pool.emplace_back(work);
  work();
  for (std::thread &t : pool) t.join();
  if (failed) {
    c->staged.clear();
    fprintf(stderr, "pmpc_hip: host -> device upload failed\n");
    throw PmpcHipError{-1, "hipMemcpyAsync (upload)", __FILE__, __LINE__};
  }
  c->staged.clear();
  for (const Chunk &ch : chunks) c->staged.push_back({ch.dst, ch.bytes, ch.off});
}

static void host_solve(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                       double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R,
                       double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu, double reg_x,
                       double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1, long long verbose, bool cone = false,
                       double smooth_alpha = std::numeric_limits<double>::quiet_NaN(), unsigned rowmajor = 0, long long cone_k = 0, int smooth_cstr = 0,
                       double smooth_beta = 1.0) {
  const size_t nx = xdim * N * M, nu = udim * N * M;
  const double nan = std::numeric_limits<double>::quiet_NaN();
  auto fail_out = [&]() {  // osqp_solver.jl:65-71 convention
    for (size_t k = 0; k < nx; k++) X_out[k] = nan;
    for (size_t k = 0; k < nu; k++) U_out[k] = nan;
  };
  if (!g_ctx && pmpc_create(&g_ctx, 0) != 0) {
    fprintf(stderr, "pmpc_hip: c_lqp_solve needs a HIP device; failing the solve (NaN outputs)\n");
    fail_out();
    return;
  }
  pmpc_ctx *c = g_ctx;
  try {
  HIP_CHECK(hipSetDevice(c->device));
  pmpc_problem p;
  memset(&p, 0, sizeof(p));
  p.xdim = xdim; p.udim = udim; p.N = N; p.M = M; p.Nc = Nc; p.reg_x = reg_x; p.reg_u = reg_u;
  // slew sentinels (tiny arrays) on the host: c_interface.jl:56-70
  bool slew_nonzero = false;
  if (!any_nan(slew_reg, M)) {
    for (size_t k = 0; k < M; k++) slew_nonzero |= (slew_reg[k] != 0.0);
    if (slew_nonzero) p.flags |= PMPC_HAS_SLEW;
  }
  if (!(any_nan(slew_reg0, M) || any_nan(slew_um1, udim * M))) p.flags |= PMPC_HAS_SLEW0;
  const void *src[19] = {x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, slew_reg, slew_reg0, slew_um1,
                         nullptr, nullptr};
  const size_t cnt[19] = {xdim * M, nx, nx * xdim, nx * udim, nx, nu, nx * xdim, nu * udim, nx, nu, nx, nx, nu, nu, M, M,
                          udim * M, nx, nu};
  bool used[19] = {true, true, true, true, true, true, true, true, true, true,
                   lx && ux, lx && ux, lu && uu, lu && uu,  // the box arrays are uploaded first and checked for NaN sentinels there
                   (bool)(p.flags & PMPC_HAS_SLEW), (bool)(p.flags & PMPC_HAS_SLEW0), (bool)(p.flags & PMPC_HAS_SLEW0),
                   true, true};
  std::vector<UploadItem> items;
  bool realloc_any = false;
  for (int k = 0; k < 19; k++) {
    if (!used[k]) continue;
    realloc_any |= c->stage[k].ensure(cnt[k] * sizeof(double));
    if (src[k]) items.push_back({c->stage[k].p, src[k], cnt[k] * sizeof(double)});
  }
  // a reallocated staging buffer holds nothing, even if the allocator hands the same address out again: the record of what
  // the previous call uploaded (upload_all's skip test) is void
  if (realloc_any) c->staged.clear();
  upload_all(c, items);
  // NaN sentinels of the boxes and exact symmetry of the cost blocks: checked on the device (one pass over what was uploaded)
  c->host_flags.ensure(4 * sizeof(int));
  HIP_CHECK(hipMemsetAsync(c->host_flags.p, 0, 4 * sizeof(int), c->stream));
  launch_host_checks(used[10] ? c->stage[10].d() : nullptr, used[11] ? c->stage[11].d() : nullptr, used[10] ? (long long)nx : 0,
                     used[12] ? c->stage[12].d() : nullptr, used[13] ? c->stage[13].d() : nullptr, used[12] ? (long long)nu : 0,
                     c->stage[6].d(), (long long)(nx * xdim), (int)xdim, c->stage[7].d(), (long long)(nu * udim), (int)udim,
                     (int *)c->host_flags.p, c->stream);
  int hf[4];
  HIP_CHECK(hipMemcpyAsync(hf, c->host_flags.p, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  if (used[10] && !hf[0]) p.flags |= PMPC_HAS_XBOUNDS;
  if (used[12] && !hf[1]) p.flags |= PMPC_HAS_UBOUNDS;
  if (!hf[2]) p.flags |= PMPC_SYMMETRIC_COST;
  used[10] = used[11] = (p.flags & PMPC_HAS_XBOUNDS) != 0;
  used[12] = used[13] = (p.flags & PMPC_HAS_UBOUNDS) != 0;
  // row-major blocks (numpy's (M, N, row, col) stacks handed over without the host-side transposition): transposed here.
  // Symmetric cost blocks are their own transpose.
  const void *blk[4] = {c->stage[2].p, c->stage[3].p, c->stage[6].p, c->stage[7].p};
  const int brow[4] = {(int)xdim, (int)xdim, (int)xdim, (int)udim}, bcol[4] = {(int)xdim, (int)udim, (int)xdim, (int)udim};
  for (int k = 0; k < 4; k++) {
    if (!(rowmajor >> k & 1u) || (k >= 2 && !hf[2])) continue;
    const size_t n = (size_t)brow[k] * bcol[k] * N * M;
    c->stage_t[k].ensure(n * sizeof(double));
    launch_block_transpose((const double *)blk[k], c->stage_t[k].d(), brow[k], bcol[k], (long long)n, c->stream);
    blk[k] = c->stage_t[k].p;
  }
  auto dp = [&](int k) { return used[k] ? (const double *)c->stage[k].p : (const double *)nullptr; };
  p.x0 = dp(0); p.f = dp(1); p.fx = dp(2); p.fu = dp(3); p.X_prev = dp(4); p.U_prev = dp(5); p.Q = dp(6); p.R = dp(7);
  p.fx = (const double *)blk[0]; p.fu = (const double *)blk[1]; p.Q = (const double *)blk[2]; p.R = (const double *)blk[3];
  p.X_ref = dp(8); p.U_ref = dp(9); p.lx = dp(10); p.ux = dp(11); p.lu = dp(12); p.uu = dp(13);
  p.slew_reg = dp(14); p.slew_reg0 = dp(15); p.slew_um1 = dp(16);
  p.X_out = c->stage[17].d(); p.U_out = c->stage[18].d();
  pmpc_info info;
  p.weights = nullptr;
  p.barrier_mu = 0.0;
  p.cone_k = cone_k;
  p.smooth_cstr = smooth_cstr; p.smooth_beta = smooth_beta;
  if (cone) pmpc_lcone_solve_device(c, &p, smooth_alpha, &info, (int)verbose);
  else pmpc_lqp_solve_device(c, &p, &info, (int)verbose);
  HIP_CHECK(hipMemcpyAsync(X_out, p.X_out, nx * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipMemcpyAsync(U_out, p.U_out, nu * sizeof(double), hipMemcpyDeviceToHost, c->stream));
  HIP_CHECK(hipStreamSynchronize(c->stream));
  if (verbose)
    printf("pmpc_hip: status %d, ipm iterations %d, structured solves %d, fast path %d\n", info.status, info.ipm_iters,
           info.structured_solves, info.fast_path);
  } catch (const PmpcHipError &) {  // failed HIP call / out of memory: the reference's failure convention, not an abort
    fail_after_error(c, nullptr, nullptr);
    fail_out();
  }
}

void c_lqp_solve(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc, double *x0,
                 double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R, double *X_ref,
                 double *U_ref, double *lx, double *ux, double *lu, double *uu, double reg_x, double reg_u,
                 double *slew_reg, double *slew_reg0, double *slew_um1, long long verbose) {
  host_solve(X_out, U_out, xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x,
             reg_u, slew_reg, slew_reg0, slew_um1, verbose);
}

void c_lcone_solve(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc, double *x0,
                   double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R, double *X_ref,
                   double *U_ref, double *lx, double *ux, double *lu, double *uu, double reg_x, double reg_u,
                   double *slew_reg, double *slew_reg0, double *slew_um1, long long verbose, double smooth_alpha,
                   char *solver) {
  // the epsilon-anchored epigraph objective of PMPC.jl/src/main.jl:204-238 (k = M through this ABI); `solver` only
  // selects the conic back end upstream (ecos / cosmo / mosek / gurobi, :320) — they share one optimum
  (void)solver;
  host_solve(X_out, U_out, xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x,
             reg_u, slew_reg, slew_reg0, slew_um1, verbose, true, smooth_alpha);
}

// Extensions of the two entry points above for callers that hold the Jacobian / cost stacks as row-major blocks (numpy's
// (M, N, row, col) arrays: the reference's Python side pays a host transposition of ~630 MB per call at M = 4096 to reach
// the column-major ABI layout, static_backend.py:83-101 through pybind11's f_style cast).  Bit k of `rowmajor` marks
// fx (0), fu (1), Q (2), R (3) as row-major; the transposition then happens in HBM after the upload.
void pmpc_lqp_solve_host(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc, double *x0,
                         double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R,
                         double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu, double reg_x,
                         double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1, long long verbose,
                         unsigned rowmajor) {
  host_solve(X_out, U_out, xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x,
             reg_u, slew_reg, slew_reg0, slew_um1, verbose, false, std::numeric_limits<double>::quiet_NaN(), rowmajor);
}

void pmpc_lcone_solve_host(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc,
                           double *x0, double *f, double *fx, double *fu, double *X_prev, double *U_prev, double *Q,
                           double *R, double *X_ref, double *U_ref, double *lx, double *ux, double *lu, double *uu,
                           double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                           long long verbose, double smooth_alpha, unsigned rowmajor, long long cone_k) {
  host_solve(X_out, U_out, xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x,
             reg_u, slew_reg, slew_reg0, slew_um1, verbose, true, smooth_alpha, rowmajor, cone_k);
}
// the same with the reference's `smooth_cstr` / `smooth_beta` settings (main.jl:247-279; pyjulia-only upstream): smooth_cstr 0 = "logbarrier",
// 1 = "squareplus"
void pmpc_lcone_solve_host_ex(double *X_out, double *U_out, size_t xdim, size_t udim, size_t N, size_t M, long long Nc, double *x0, double *f,
                              double *fx, double *fu, double *X_prev, double *U_prev, double *Q, double *R, double *X_ref, double *U_ref, double *lx,
                              double *ux, double *lu, double *uu, double reg_x, double reg_u, double *slew_reg, double *slew_reg0, double *slew_um1,
                              long long verbose, double smooth_alpha, unsigned rowmajor, long long cone_k, int smooth_cstr, double smooth_beta) {
  host_solve(X_out, U_out, xdim, udim, N, M, Nc, x0, f, fx, fu, X_prev, U_prev, Q, R, X_ref, U_ref, lx, ux, lu, uu, reg_x, reg_u, slew_reg, slew_reg0,
             slew_um1, verbose, true, smooth_alpha, rowmajor, cone_k, smooth_cstr, smooth_beta);
}

}  // extern "C"
