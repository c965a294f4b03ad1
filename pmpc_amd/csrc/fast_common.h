// fast_common.h — lane layout and cross-lane helpers shared by the register-resident MFMA kernels
// (kernels_fast.hip: sweeps of the interior-point / equality solves; kernels_as.hip: sweeps of the active-set rounds).
// See the header comment of kernels_fast.hip for the layouts.
#pragma once
#include "pmpc_dev.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ double readlane_d(double v, int l) {
  int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
  int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
  return __hiloint2double(hi, lo);
}
// a lane-uniform value into scalar registers (value of the first active lane)
__device__ __forceinline__ double rfl_d(double v) {
  int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
  // every lane has a valid source for these permutations: no "old" value needed (saves the init moves)
  int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xF, 0xF, true);
  int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum over the 16 lanes of a DPP row (all lanes get the total): xor1, xor2, half-mirror, mirror
__device__ __forceinline__ double row_allsum(double v) {
  v += dpp_d<0xB1>(v);
  v += dpp_d<0x4E>(v);
  v += dpp_d<0x141>(v);
  v += dpp_d<0x140>(v);
  return v;
}
// Cross-row exchanges on the VALU (gfx950 v_permlane32_swap / v_permlane16_swap; semantics checked on
// hardware, tools/micro/swap_test): with both operands equal to v,
//   permlane32_swap -> [0] = v of the lane in rows {0,1} at the same position, [1] = same for rows {2,3}
//   permlane16_swap -> [0] = v of the even row of this row pair,               [1] = v of the odd row
// Same-box A/B vs ds_bpermute shuffles: factor sweep -4.5 %, vector sweep -6 % at 256 particles (latency-bound).
__device__ __forceinline__ void swap32_d(double v, double &a, double &b) {
  auto lo = __builtin_amdgcn_permlane32_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane32_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
__device__ __forceinline__ void swap16_d(double v, double &a, double &b) {
  auto lo = __builtin_amdgcn_permlane16_swap(__double2loint(v), __double2loint(v), false, false);
  auto hi = __builtin_amdgcn_permlane16_swap(__double2hiint(v), __double2hiint(v), false, false);
  a = __hiloint2double(hi[0], lo[0]);
  b = __hiloint2double(hi[1], lo[1]);
}
// sum over the 4 k-groups (lanes c, c+16, c+32, c+48), all lanes get the total
__device__ __forceinline__ double grp_allsum(double v) {
  double a, b;
  swap32_d(v, a, b);
  v = a + b;
  swap16_d(v, a, b);
  return a + b;
}
// out[k] = v of lane (c, k): every lane gets the values its column holds in all four k-groups
__device__ __forceinline__ void grp_gather(double v, double (&out)[4]) {
  double p01, p23;
  swap32_d(v, p01, p23);
  swap16_d(p01, out[0], out[1]);
  swap16_d(p23, out[2], out[3]);
}
// 1/sqrt(d): v_rsq_f64 seed (measured max rel. error 5.1e-8 on gfx950, tools/micro/rsq_test.hip) + ONE Newton
// step -> 3.8e-15; a second step (3.4e-16) buys nothing for a Cholesky pivot and sits on the critical path
__device__ __forceinline__ double rsqrt_d(double d) {
  const double r = __builtin_amdgcn_rsq(d);
  const double e = fma(-0.5 * d * r, r, 0.5);  // 0.5 (1 - d r^2)
  return fma(r, e, r);
}
__device__ __forceinline__ const double *badd(const double *p, long long bytes) {
  return (const double *)((const char *)p + bytes);
}
// arr + off for a wave-uniform byte offset, pinned to scalar registers: loads through it take the `saddr + 32-bit lane offset`
// form.  (Left to itself the loop optimiser turns `uniform stage base + lane offset` into one 64-bit per-lane pointer per array,
// stepped with two vector adds per stage: more registers and more instructions than the kernels can afford.)
__device__ __forceinline__ const double *ubase(const double *arr, long long off) {
  const unsigned long long v = (unsigned long long)arr + (unsigned long long)off;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const double *)(((unsigned long long)hi << 32) | lo);
}
// loads / stores through pointers the compiler can no longer prove to be global (selected per lane between several arrays):
// without the address-space cast they become FLAT instructions, which also tie up the LDS counter the lane shuffles wait on
typedef const double __attribute__((address_space(1))) *gcptr_d;
typedef double __attribute__((address_space(1))) *gptr_d;
__device__ __forceinline__ double gld(const void *p) { return *(gcptr_d)(unsigned long long)p; }
__device__ __forceinline__ void gst(void *p, double v) { *(gptr_d)(unsigned long long)p = v; }
__device__ __forceinline__ double ldo(const double *base, unsigned boff) {  // uniform base + 32-bit byte offset
  return *(gcptr_d)((const __attribute__((address_space(1))) char *)(unsigned long long)base + boff);
}
__device__ __forceinline__ void gsto(const double *base, unsigned boff, double v) {  // store, same addressing
  *(gptr_d)((__attribute__((address_space(1))) char *)(unsigned long long)base + boff) = v;
}
__device__ __forceinline__ void gsto_i(const double *base, unsigned boff, int v) {
  *(int __attribute__((address_space(1))) *)((__attribute__((address_space(1))) char *)(unsigned long long)base + boff) = v;
}

// matrix stacks (fx, fu, Q, R) and the factor record in a storage type MT (double, or float: the fp32-storage mode — half the
// HBM bytes of the dominant arrays; every value is widened on load, the arithmetic stays fp64)
template <class MT>
__device__ __forceinline__ double gldm(const void *p) { return (double)*(const MT __attribute__((address_space(1))) *)(unsigned long long)p; }
template <class MT>
__device__ __forceinline__ double ldom(const void *base, unsigned boff) {  // uniform base + 32-bit byte offset
  return (double)*(const MT __attribute__((address_space(1))) *)((const __attribute__((address_space(1))) char *)(unsigned long long)base + boff);
}
template <class MT>
__device__ __forceinline__ void gstom(const void *base, unsigned boff, double v) {
  *(MT __attribute__((address_space(1))) *)((__attribute__((address_space(1))) char *)(unsigned long long)base + boff) = (MT)v;
}
__device__ __forceinline__ const void *ubase_v(const void *arr, long long off) { return (const void *)ubase((const double *)arr, off); }

// unconditional load from a per-lane VALID address, zeroed by a select (no exec-mask branch);
// `rv` guards padding rows (only when xdim is not a multiple of 4, where p[r] could leave the block)
template <bool PADX>
__device__ __forceinline__ double ldsel(const double *p, bool keep, bool rv) {
  if (PADX) return (keep && rv) ? *p : 0.0;
  const double t = *p;
  return keep ? t : 0.0;
}

template <int XD, int UD>
struct Lane {
  static constexpr int KS = (XD + 3) / 4, XP = 4 * KS;
  int c, g, oc, cb, row0;
  bool cxv, cu;
  __device__ explicit Lane(int lane) {
    c = lane & 15;
    g = lane >> 4;
    oc = (c & 3) * KS + (c >> 2);  // original state index of kernel column c
    cxv = c < XP && oc < XD;
    cb = c - XP;
    cu = cb >= 0 && cb < UD;
    row0 = KS * g;  // original index of kernel row g + 4r is row0 + r
  }
};

// s_row[r] = s_col of the lane that owns kernel column g + 4r (same k-group)
template <int KS>
__device__ __forceinline__ void col_to_row(double s_col, int g, double *s_row) {
#pragma unroll
  for (int r = 0; r < KS; r++) s_row[r] = __shfl(s_col, (g + 4 * r) + 16 * g, 64);
}

// v[k] for a per-lane index k < UD <= 4, by selects on scalars (an array indexed through a loop can end up as a dynamically
// indexed stack object — scratch memory — when the optimiser meets it before it has been split into registers)
__device__ __forceinline__ double pick4(double v0, double v1, double v2, double v3, int k) {
  double o = v0;
  o = (k == 1) ? v1 : o;
  o = (k == 2) ? v2 : o;
  o = (k == 3) ? v3 : o;
  return o;
}
template <int UD>
__device__ __forceinline__ double pick(const double (&v)[UD], int k) {
  static_assert(UD <= 4, "pick: at most 4 controls");
  return pick4(v[0], UD > 1 ? v[UD > 1 ? 1 : 0] : 0.0, UD > 2 ? v[UD > 2 ? 2 : 0] : 0.0, UD > 3 ? v[UD > 3 ? 3 : 0] : 0.0, k);
}

// y = (L L')^-1 y with L given as strict lower part + reciprocal diagonal
template <int UD>
__device__ __forceinline__ void chol_solve(const double (&Lc)[UD][UD], const double (&Ld)[UD], double (&y)[UD]) {
#pragma unroll
  for (int p = 0; p < UD; p++) {
    double v = y[p];
#pragma unroll
    for (int k = 0; k < p; k++) v -= Lc[p][k] * y[k];
    y[p] = v * Ld[p];
  }
#pragma unroll
  for (int p = UD - 1; p >= 0; p--) {
    double v = y[p];
#pragma unroll
    for (int k = p + 1; k < UD; k++) v -= Lc[k][p] * y[k];
    y[p] = v * Ld[p];
  }
}


}  // namespace

// (xdim, udim) pairs with compiled instances (PMPC_DIAG_DIMS_12_4: diagnostic / A-B builds of the quadrotor shape only — the full
// list takes minutes per file)
#ifdef PMPC_DIAG_DIMS_12_4
#define PMPC_FAST_DIMS(X) X(12, 4)
#else
#define PMPC_FAST_DIMS(X)                                                                                          \
  X(12, 4) X(12, 3) X(12, 2) X(10, 4) X(10, 2) X(9, 4) X(9, 3) X(8, 4) X(8, 2) X(7, 3) X(6, 4) X(6, 3) X(6, 2) X(5, 3) \
  X(5, 2) X(4, 4) X(4, 3) X(4, 2) X(4, 1) X(3, 3) X(3, 2) X(3, 1) X(2, 2) X(2, 1) X(1, 1)
#endif
