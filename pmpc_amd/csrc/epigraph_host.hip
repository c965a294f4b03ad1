// epigraph_host.hip — the reference's cone objective in the space of the SHARED controls (host code; no kernels).
//
// The reference's default solver path (PMPC.jl/src/main.jl:204-238) minimises
//     (1 + eps) sum_i y_i + (1 - eps) k t      s.t.   J_i(z) <= y_i + t,   y >= 0,   dynamics, boxes
// over every particle's trajectory at once.  With hard boxes a particle's optimal free variables GIVEN the shared controls u_c do not
// depend on how its cost is weighted, so with V_i(u_c) = min over particle i's own variables of J_i the problem is
//     min over (u_c, t, y >= 0) of  (1 + eps) sum y_i + (1 - eps) k t   s.t.   V_i(u_c) <= y_i + t
// — Nc * udim + 1 + M variables — and on the particles' current active sets V_i is the quadratic
// V_i(u_c + d) = J_i + g_i'd + 1/2 d'H_i d  that the factor sweeps leave per particle (condensed gradient / Hessian, LQArgs::gc_part,
// Hc_part).  Multipliers lam_i in [0, 1 + eps] of the M rows, sum lam_i = (1 - eps) k: lam_i = 1 + eps above the threshold cost t,
// 0 below, anything in between ON it — with ANY number of particles on it (ties are ordinary degenerate rows here; the rank-based
// weight assignment of the earlier rounds could place one or two).
//
// Solved by the proximal method of multipliers: for multipliers lam the augmented function
//     Phi(d, t) = (1 - eps) k t + sum_i max over mu in [0, 1 + eps] of  mu (V_i(d) - t) - rho/2 (mu - lam_i)^2
// is convex and piecewise quadratic in the Nc * udim + 1 unknowns; a semismooth Newton iteration with t eliminated exactly (its optimality
// condition sum_i mu_i = (1 - eps) k is a monotone scalar equation) minimises it, the maximisers mu replace lam, and the pair is a KKT
// point of the epigraph problem when they reproduce themselves.  No partition bookkeeping, identical particles (every cost on the
// threshold, multipliers not unique) included: the proximal term picks the multipliers closest to the previous ones.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>

namespace {

struct Epi {
  int M, nc, nf;            // particles, shared controls, free (not held) ones
  const double *J, *H, *g;  // V_i(d) = J_i + g_i'd + 1/2 d'H_i d;  H: nc x nc per particle, column-major, symmetric
  std::vector<int> fr;      // indices of the free components
  double K, cap, rho;
  std::vector<double> V, dV;  // values and gradients (M x nc) at the current d
  void eval(const std::vector<double> &d) {
    V.assign(M, 0.0);
    dV.assign((size_t)M * nc, 0.0);
    for (int i = 0; i < M; i++) {
      const double *Hi = H + (size_t)i * nc * nc, *gi = g + (size_t)i * nc;
      double v = J[i];
      for (int r = 0; r < nc; r++) {
        double hd = 0.0;
        for (int c = 0; c < nc; c++) hd += Hi[r + (size_t)nc * c] * d[c];
        dV[(size_t)i * nc + r] = gi[r] + hd;
        v += d[r] * (gi[r] + 0.5 * hd);
      }
      V[i] = v;
    }
  }
  static double clip(double x, double hi) { return x < 0.0 ? 0.0 : (x > hi ? hi : x); }
  // t with sum_i clip(lam_i + (V_i - t) / rho, 0, cap) = K (continuous, non-increasing in t): bisection to the cell, then the cell's
  // own linear equation
  double solve_t(const std::vector<double> &lam) const {
    double lo = 1e300, hi = -1e300;
    for (int i = 0; i < M; i++) {
      lo = std::min(lo, V[i] + rho * (lam[i] - cap));
      hi = std::max(hi, V[i] + rho * lam[i]);
    }
    lo -= 1.0; hi += 1.0;
    auto S = [&](double t) {
      double s = 0.0;
      for (int i = 0; i < M; i++) s += clip(lam[i] + (V[i] - t) / rho, cap);
      return s - K;
    };
    for (int it = 0; it < 200 && hi - lo > 1e-15 * std::max(1.0, std::fabs(lo) + std::fabs(hi)); it++) {
      const double mid = 0.5 * (lo + hi);
      if (S(mid) > 0.0) lo = mid; else hi = mid;
    }
    const double tm = 0.5 * (lo + hi);
    double num = 0.0, ncap = 0.0;
    int nu = 0;
    for (int i = 0; i < M; i++) {
      const double x = lam[i] + (V[i] - tm) / rho;
      if (x >= cap) ncap += 1.0;
      else if (x > 0.0) { num += lam[i] * rho + V[i]; nu++; }
    }
    if (nu == 0) return tm;
    const double t = (num + rho * (cap * ncap - K)) / nu;
    return (t >= lo - 1e-9 * std::max(1.0, std::fabs(tm)) && t <= hi + 1e-9 * std::max(1.0, std::fabs(tm))) ? t : tm;
  }
  double phi(const std::vector<double> &lam, double t) const {
    double f = K * t;
    for (int i = 0; i < M; i++) {
      const double v = V[i] - t, mu = clip(lam[i] + v / rho, cap);
      f += mu * v - 0.5 * rho * (mu - lam[i]) * (mu - lam[i]);
    }
    return f;
  }
};

// dense SPD solve (n <= a few hundred), in place: A x = b; returns false if a pivot is not positive
bool chol_solve_dense(std::vector<double> &A, std::vector<double> &b, int n) {
  for (int q = 0; q < n; q++) {
    double d = A[q + (size_t)n * q];
    for (int k = 0; k < q; k++) d -= A[q + (size_t)n * k] * A[q + (size_t)n * k];
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    A[q + (size_t)n * q] = d;
    for (int p = q + 1; p < n; p++) {
      double v = A[p + (size_t)n * q];
      for (int k = 0; k < q; k++) v -= A[p + (size_t)n * k] * A[q + (size_t)n * k];
      A[p + (size_t)n * q] = v / d;
    }
  }
  for (int p = 0; p < n; p++) {
    double v = b[p];
    for (int k = 0; k < p; k++) v -= A[p + (size_t)n * k] * b[k];
    b[p] = v / A[p + (size_t)n * p];
  }
  for (int p = n - 1; p >= 0; p--) {
    double v = b[p];
    for (int k = p + 1; k < n; k++) v -= A[k + (size_t)n * p] * b[k];
    b[p] = v / A[p + (size_t)n * p];
  }
  return true;
}

}  // namespace

// lam: in = current multipliers (any point of [0, cap]^M), out = multipliers of the epigraph problem on the given quadratics;
// delta (nc): step of the shared controls, t_out: threshold cost.  held[r] != 0: shared control r sits on its bound (no step).
// Returns 0 converged, 1 not converged.
extern "C" int pmpc_epigraph_solve_host(int M, int nc, const double *J, const double *H, const double *g, const unsigned char *held, double K, double cap,
                             double *lam_io, double *delta, double *t_out, int verbose) {
  Epi e;
  e.M = M; e.nc = nc; e.J = J; e.H = H; e.g = g; e.K = K; e.cap = cap;
  for (int r = 0; r < nc; r++)
    if (!held || !held[r]) e.fr.push_back(r);
  e.nf = (int)e.fr.size();
  double jlo = 1e300, jhi = -1e300;
  for (int i = 0; i < M; i++) { jlo = std::min(jlo, J[i]); jhi = std::max(jhi, J[i]); }
  const double scale = std::max(1.0, std::max(std::fabs(jlo), std::fabs(jhi)));
  e.rho = 1e-6 * scale;
  std::vector<double> lam(lam_io, lam_io + M), d(nc, 0.0), mu(M), grad(e.nf), step(e.nf), A((size_t)e.nf * e.nf), b(e.nf), dtry(nc);
  double t = 0.0;
  int status = 1, newton_total = 0;
  for (int outer = 0; outer < 40; outer++) {
    bool inner_ok = false;
    for (int it = 0; it < 60; it++) {
      e.eval(d);
      t = e.solve_t(lam);
      std::fill(grad.begin(), grad.end(), 0.0);
      std::fill(A.begin(), A.end(), 0.0);
      std::fill(b.begin(), b.end(), 0.0);
      double cnt = 0.0, gscale = 1.0;
      for (int i = 0; i < M; i++) {
        const double x = lam[i] + (e.V[i] - t) / e.rho, m = Epi::clip(x, cap);
        mu[i] = m;
        const double *dv = &e.dV[(size_t)i * nc], *Hi = H + (size_t)i * nc * nc;
        const bool unc = x > 0.0 && x < cap;
        double gmax = 0.0;
        for (int a_ = 0; a_ < e.nf; a_++) {
          const double ga = dv[e.fr[a_]];
          gmax = std::max(gmax, std::fabs(ga));
          grad[a_] += m * ga;
          if (unc) b[a_] += ga / e.rho;
          for (int c_ = 0; c_ < e.nf; c_++) {
            double v = m * Hi[e.fr[a_] + (size_t)nc * e.fr[c_]];
            if (unc) v += ga * dv[e.fr[c_]] / e.rho;
            A[a_ + (size_t)e.nf * c_] += v;
          }
        }
        gscale += m * gmax;
        if (unc) cnt += 1.0 / e.rho;
      }
      double gn = 0.0;
      for (int a_ = 0; a_ < e.nf; a_++) gn = std::max(gn, std::fabs(grad[a_]));
      if (gn <= 1e-13 * gscale) { inner_ok = true; break; }
      if (cnt > 0.0)
        for (int a_ = 0; a_ < e.nf; a_++)
          for (int c_ = 0; c_ < e.nf; c_++) A[a_ + (size_t)e.nf * c_] -= b[a_] * b[c_] / cnt;
      for (int a_ = 0; a_ < e.nf; a_++) step[a_] = -grad[a_];
      {
        std::vector<double> Ac = A;
        double reg = 0.0;
        while (!chol_solve_dense(Ac, step, e.nf)) {  // (round-off of the rank-one correction: lift the diagonal a little)
          reg = reg == 0.0 ? 1e-12 : 10.0 * reg;
          Ac = A;
          for (int a_ = 0; a_ < e.nf; a_++) { Ac[a_ + (size_t)e.nf * a_] *= 1.0 + reg; step[a_] = -grad[a_]; }
          if (reg > 1e-2) break;
        }
      }
      // backtracking on phi (piecewise quadratic, convex: the full step is exact once the clipping pattern has settled)
      const double f0 = e.phi(lam, t);
      double slope = 0.0;
      for (int a_ = 0; a_ < e.nf; a_++) slope += grad[a_] * step[a_];
      double al = 1.0;
      for (int ls = 0; ls < 40; ls++) {
        dtry = d;
        for (int a_ = 0; a_ < e.nf; a_++) dtry[e.fr[a_]] += al * step[a_];
        e.eval(dtry);
        const double tt = e.solve_t(lam);
        if (e.phi(lam, tt) <= f0 + 1e-4 * al * slope + 1e-14 * std::fabs(f0)) break;
        al *= 0.5;
      }
      d = dtry;
      newton_total++;
    }
    e.eval(d);
    t = e.solve_t(lam);
    double dl = 0.0;
    for (int i = 0; i < M; i++) {
      const double m = Epi::clip(lam[i] + (e.V[i] - t) / e.rho, cap);
      dl = std::max(dl, std::fabs(m - lam[i]));
      lam[i] = m;
    }
    if (verbose) printf("pmpc_hip: epigraph (host) outer %d: %d Newton steps so far, t %.12e, multiplier change %.3e%s\n", outer + 1, newton_total, t, dl, inner_ok ? "" : " (inner not converged)");
    if (inner_ok && dl <= 1e-10 * cap) { status = 0; break; }
  }
  // the multipliers sum to K up to round-off of the scalar equation: spread what is missing over the rows strictly inside (0, cap)
  double sum = 0.0;
  int nin = 0;
  for (int i = 0; i < M; i++) { sum += lam[i]; nin += (lam[i] > 0.0 && lam[i] < cap) ? 1 : 0; }
  if (nin > 0) {
    const double fix = (K - sum) / nin;
    for (int i = 0; i < M; i++)
      if (lam[i] > 0.0 && lam[i] < cap) lam[i] = Epi::clip(lam[i] + fix, cap);
  }
  for (int i = 0; i < M; i++) lam_io[i] = lam[i];
  for (int r = 0; r < nc; r++) delta[r] = d[r];
  *t_out = t;
  return status;
}
