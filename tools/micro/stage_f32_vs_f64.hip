// stage_f32_vs_f64.hip — VERDICT r04 item 9: decide fp32 ARITHMETIC for the Riccati sweeps with a micro-benchmark, not by default.
//
// The factor sweep of kernels_fast.hip (k_bwd_fast<12, 4, FACTOR, no bounds, DEEP>: the register-resident stage the active-set sweeps
// are built on — F'SF on 16x16x4 MFMA tiles, lane-uniform 4x4 Cholesky, in-lane substitution, S' = Hxx - Hxu K, gradient recursion,
// one coalesced record store per stage) as a template on the arithmetic type: double = what ships (v_mfma_f64_16x16x4_f64, 2 dpp /
// readlane moves per value), float = v_mfma_f32_16x16x4_f32, one 32-bit move per value, half the bytes.  Same lane layout, same loads
// and stores, same instruction structure; synthetic contractive stacks (x12, u4, N stages).  Prints the launch time of both at 512 and
// 4096 particles and the relative difference of the fp32 factor records from the fp64 ones (what a refinement round would have to remove).
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/stage_f32_vs_f64.hip -o tools/micro/stage_f32_vs_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>

#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { fprintf(stderr, "HIP error %s at line %d\n", hipGetErrorString(_e), __LINE__); exit(1); } } while (0)

typedef double v4d __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
template <class T> struct V4;
template <> struct V4<double> { typedef v4d type; };
template <> struct V4<float> { typedef v4f type; };

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
__device__ __forceinline__ v4f mfma(float a, float b, v4f c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ double readlane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ float readlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
template <int CTRL> __device__ __forceinline__ double dpp(double v) {
  return __hiloint2double(__builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true), __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ float dpp(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true)); }
__device__ __forceinline__ void swap32(double v, double &a, double &b) {
  auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap32(float v, float &a, float &b) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_int(v), __float_as_int(v), false, false);
  a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
__device__ __forceinline__ void swap16(double v, double &a, double &b) {
  auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]); b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16(float v, float &a, float &b) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_int(v), __float_as_int(v), false, false);
  a = __int_as_float(r[0]); b = __int_as_float(r[1]);
}
template <class T> __device__ __forceinline__ T grp_allsum(T v) { T a, b; swap32(v, a, b); v = a + b; swap16(v, a, b); return a + b; }
template <class T> __device__ __forceinline__ void grp_gather(T v, T (&out)[4]) { T p01, p23; swap32(v, p01, p23); swap16(p01, out[0], out[1]); swap16(p23, out[2], out[3]); }
__device__ __forceinline__ double rsqrt_t(double d) { const double r = __builtin_amdgcn_rsq(d); return fma(r, fma(-0.5 * d * r, r, 0.5), r); }
__device__ __forceinline__ float rsqrt_t(float d) { const float r = __builtin_amdgcn_rsqf(d); return fmaf(r, fmaf(-0.5f * d * r, r, 0.5f), r); }
template <class T> __device__ __forceinline__ T pick4(T v0, T v1, T v2, T v3, int k) { T o = v0; o = k == 1 ? v1 : o; o = k == 2 ? v2 : o; o = k == 3 ? v3 : o; return o; }

constexpr int XD = 12, UD = 4, KS = 3, XP = 12;

template <class T>
__global__ void __launch_bounds__(64) k_stage(const T *fx, const T *fu, const T *Q, const T *R, const T *xm, const T *xd, const T *um, const T *ud, T *K, T *kff, int N,
                                              T regx, T regu) {
  typedef typename V4<T>::type v4;
  const int lane = threadIdx.x, c = lane & 15, g = lane >> 4, i = blockIdx.x;
  const int oc = (c & 3) * KS + (c >> 2), cb = c - XP, row0 = KS * g;
  const bool cxv = c < XP, cu = cb >= 0 && cb < UD;
  const size_t pbase = (size_t)i * N;
  const T *pF = cxv ? fx + (pbase + N - 1) * (XD * XD) + XD * oc + row0 : fu + (pbase + N - 1) * (XD * UD) + XD * cb + row0;
  const int sF = cxv ? XD * XD : XD * UD;
  const T *pQ = Q + (pbase + N - 1) * (XD * XD) + (cxv ? XD * oc + row0 : 0);
  const T *pR = R + (pbase + N - 1) * (UD * UD) + (cu ? g + UD * cb : 0);
  const T *pgu = ud + (pbase + N - 1) * UD + (cu ? cb : 0);
  const T *pgx = xd + (pbase + N - 1) * XD + (cxv ? oc : 0);
  const T *pxm = xm + (pbase + N - 1) * XD + row0;
  const T *pum = um + (pbase + N - 1) * UD + g;
  T *pRec = K + (pbase + N - 1) * 64 + lane, *pk = kff + (pbase + N - 1) * UD;
  const bool diag_x = cxv && ((c & 3) == g), umask = cu && g == cb;
  const T mx = cxv ? T(1) : T(0), mu = cu ? T(1) : T(0);
  T S[KS], s_row[KS], s_col, Fn[KS], Qn[KS], xmn[KS], gxn, Rn, umn, gun;
  auto col_to_row = [&](T v, T *o) {
#pragma unroll
    for (int r = 0; r < KS; r++) o[r] = __shfl(v, (g + 4 * r) + 16 * g, 64);
  };
  {  // terminal
    T part = 0;
#pragma unroll
    for (int r = 0; r < KS; r++) { const T q = mx * pQ[r]; part += q * pxm[r]; S[r] = q + ((diag_x && (c >> 2) == r) ? regx : T(0)); }
    part = grp_allsum(part);
    s_col = cxv ? part + *pgx : T(0);
    col_to_row(s_col, s_row);
#pragma unroll
    for (int r = 0; r < KS; r++) Fn[r] = pF[r];
    gun = mu * *pgu; Rn = mu * *pR; umn = *pum;
    pQ -= XD * XD; pgx -= XD; pxm -= XD;
    gxn = mx * *pgx;
#pragma unroll
    for (int r = 0; r < KS; r++) { Qn[r] = mx * pQ[r]; xmn[r] = pxm[r]; }
  }
  for (int j = N - 1; j >= 0; j--) {
    T Fr[KS], Qc[KS], xm_row[KS];
#pragma unroll
    for (int r = 0; r < KS; r++) { Fr[r] = (j == 0 && cxv) ? T(0) : Fn[r]; Qc[r] = Qn[r]; xm_row[r] = xmn[r]; }
    const T Rc = Rn, um_g = umn, gu_c = gun, gx_c = gxn;
    if (j > 1) {
      pQ -= XD * XD; pgx -= XD; pxm -= XD;
      gxn = mx * *pgx;
#pragma unroll
      for (int r = 0; r < KS; r++) { Qn[r] = mx * pQ[r]; xmn[r] = pxm[r]; }
    }
    if (j > 0) {
      pF -= sF; pgu -= UD; pR -= UD * UD; pum -= UD;
#pragma unroll
      for (int r = 0; r < KS; r++) Fn[r] = pF[r];
      gun = mu * *pgu; Rn = mu * *pR; umn = *pum;
    }
    T hp = Rc * um_g;
#pragma unroll
    for (int r = 0; r < KS; r++) hp = fma(Fr[r], s_row[r], hp);
    const T h_col = grp_allsum(hp) + gu_c;
    T hu[UD];
#pragma unroll
    for (int b = 0; b < UD; b++) hu[b] = readlane(h_col, XP + b);
    v4 H = {0, 0, 0, 0};
    if (j > 0) {
#pragma unroll
      for (int r = 0; r < KS; r++) H[r] = Qc[r] + ((diag_x && (c >> 2) == r) ? regx : T(0));
    }
    H[KS] = Rc + (umask ? regu : T(0));
    v4 G = {0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < KS; r++) G = mfma(S[r], Fr[r], G);
#pragma unroll
    for (int r = 0; r < KS; r++) H = mfma(Fr[r], G[r], H);
    T Lc[UD][UD], Ld[UD], col[UD];
#pragma unroll
    for (int q = 0; q < UD; q++) {
#pragma unroll
      for (int pp = q; pp < UD; pp++) {
        T v = readlane(H[KS], (XP + q) + 16 * pp);
#pragma unroll
        for (int k = 0; k < q; k++) v -= Lc[pp][k] * Lc[q][k];
        if (pp == q) Ld[q] = rsqrt_t(v); else Lc[pp][q] = v * Ld[q];
      }
    }
    T rows4[4];
    grp_gather(H[KS], rows4);
#pragma unroll
    for (int k = 0; k < UD; k++) col[k] = cu ? (cb == k ? T(1) : T(0)) : rows4[k];
#pragma unroll
    for (int p = 0; p < UD; p++) { T v = col[p]; for (int k = 0; k < p; k++) v -= Lc[p][k] * col[k]; col[p] = v * Ld[p]; }
#pragma unroll
    for (int p = UD - 1; p >= 0; p--) { T v = col[p]; for (int k = p + 1; k < UD; k++) v -= Lc[k][p] * col[k]; col[p] = v * Ld[p]; }
    const T Kg = pick4(col[0], col[1], col[2], col[3], g);
    const T rec = (cxv || cu) ? Kg : T(0);
    v4 Sn = mfma(H[KS], cxv ? -Kg : T(0), H);
#pragma unroll
    for (int r = 0; r < KS; r++) S[r] = Sn[r];
    *pRec = rec; pRec -= 64;
    const T Kreg = cxv ? rec : T(0), hug = pick4(hu[0], hu[1], hu[2], hu[3], g);
    T kq = cu ? rec * pick4(hu[0], hu[1], hu[2], hu[3], cb) : T(0);
    kq += dpp<0xB1>(kq);
    kq += dpp<0x4E>(kq);
    if (c == XP) pk[g] = kq;
    pk -= UD;
    if (j == 0) break;
    T p2 = -Kreg * hug;
#pragma unroll
    for (int r = 0; r < KS; r++) p2 = fma(Qc[r], xm_row[r], p2);
    const T red2 = grp_allsum(p2);
    s_col = cxv ? h_col + red2 + gx_c : T(0);
    col_to_row(s_col, s_row);
  }
}

template <class T>
__global__ void k_fill(T *p, size_t n, unsigned seed, int kind, int blk) {  // kind 0: small noise; 1: fx = 0.9 I + noise (col-major 12x12); 2: SPD-ish diagonal + noise (blk x blk)
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)k * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    const double r = (h & 0xFFFFFF) / 16777216.0 - 0.5;
    double v = 0.2 * r;
    if (kind == 1) { const int e = (int)(k % 144); v = 0.1 * r + ((e % 13) == 0 ? 0.9 : 0.0); }
    if (kind == 2) { const int e = (int)(k % (blk * blk)); v = 0.02 * r + ((e % (blk + 1)) == 0 ? 1.0 + 0.5 * (r + 0.5) : 0.0); }
    p[k] = (T)v;
  }
}
__global__ void k_sym(double *p, size_t nblk, int blk) {  // symmetrise the cost blocks
  for (size_t b = blockIdx.x * (size_t)blockDim.x + threadIdx.x; b < nblk; b += (size_t)gridDim.x * blockDim.x)
    for (int r = 0; r < blk; r++) for (int q = r + 1; q < blk; q++) { const double v = 0.5 * (p[b * blk * blk + r + blk * q] + p[b * blk * blk + q + blk * r]); p[b * blk * blk + r + blk * q] = p[b * blk * blk + q + blk * r] = v; }
}
template <class T> __global__ void k_cast(const double *s, T *d, size_t n) { for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x) d[k] = (T)s[k]; }

template <class T>
double run(int M, int N, const double *const src[8], const size_t n[8], std::vector<double> *rec_out, double *ms_min) {
  T *a[8];
  for (int k = 0; k < 8; k++) { CK(hipMalloc(&a[k], n[k] * sizeof(T))); k_cast<T><<<1024, 256>>>(src[k], a[k], n[k]); }
  T *K, *kff;
  CK(hipMalloc(&K, (size_t)M * N * 64 * sizeof(T))); CK(hipMalloc(&kff, (size_t)M * N * UD * sizeof(T)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() { k_stage<T><<<M, 64>>>(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7], K, kff, N, (T)1.0, (T)0.1); };
  for (int w = 0; w < 3; w++) launch();
  CK(hipDeviceSynchronize());
  double sum = 0.0, mn = 1e30;
  const int reps = 20;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); sum += ms; mn = ms < mn ? ms : mn;
  }
  if (rec_out) {
    std::vector<T> h((size_t)M * N * 64);
    CK(hipMemcpy(h.data(), K, h.size() * sizeof(T), hipMemcpyDeviceToHost));
    rec_out->assign(h.begin(), h.end());
  }
  for (int k = 0; k < 8; k++) CK(hipFree(a[k]));
  CK(hipFree(K)); CK(hipFree(kff));
  *ms_min = mn;
  return sum / reps;
}

int main() {
  const int N = 50;
  for (int M : {512, 4096}) {
    const size_t rows = (size_t)M * N;
    const size_t n[8] = {rows * 144, rows * 48, rows * 144, rows * 16, rows * 12, rows * 12, rows * 4, rows * 4};
    const int kind[8] = {1, 0, 2, 2, 0, 0, 0, 0}, blk[8] = {12, 0, 12, 4, 0, 0, 0, 0};
    double *src[8];
    for (int k = 0; k < 8; k++) { CK(hipMalloc(&src[k], n[k] * sizeof(double))); k_fill<double><<<1024, 256>>>(src[k], n[k], 77u + k, kind[k], blk[k]); }
    k_sym<<<1024, 256>>>(src[2], rows, 12); k_sym<<<1024, 256>>>(src[3], rows, 4);
    CK(hipDeviceSynchronize());
    std::vector<double> r64, r32;
    double mn64, mn32;
    const double t64 = run<double>(M, N, src, n, &r64, &mn64), t32 = run<float>(M, N, src, n, &r32, &mn32);
    if (getenv("DEBUG")) {
      for (int j = N - 1; j >= N - 3; j--) { printf("stage %d fp64 rec:", j); for (int k = 0; k < 64; k += 5) printf(" %.3e", r64[(size_t)j * 64 + k]); printf("\n"); }
    }
    double num = 0.0, den = 0.0; bool finite = true;
    size_t bad64 = 0, bad32 = 0, first_bad = (size_t)-1;
    for (size_t k = 0; k < r64.size(); k++) {
      const bool f64 = std::isfinite(r64[k]), f32 = std::isfinite(r32[k]);
      bad64 += !f64; bad32 += !f32;
      if (!(f64 && f32)) { finite = false; if (first_bad == (size_t)-1) first_bad = k; continue; }
      const double d = r32[k] - r64[k]; num += d * d; den += r64[k] * r64[k];
    }
    if (!finite) printf("  non-finite record entries: fp64 %zu, fp32 %zu of %zu; first at particle %zu stage %zu slot %zu\n", bad64, bad32, r64.size(), first_bad / 64 / N, (first_bad / 64) % N, first_bad % 64);
    printf("M %5d N %d: factor sweep fp64 %.1f us (min %.1f), fp32 %.1f us (min %.1f): fp64 / fp32 = %.2fx; fp32 records vs fp64: rel 2-norm diff %.2e%s\n", M, N, 1e3 * t64,
           1e3 * mn64, 1e3 * t32, 1e3 * mn32, t64 / t32, std::sqrt(num / den), finite ? "" : " (NON-FINITE)");
    for (int k = 0; k < 8; k++) CK(hipFree(src[k]));
  }
  return 0;
}
