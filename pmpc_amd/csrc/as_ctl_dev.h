// as_ctl_dev.h — round control of the active-set iteration as a block-wide device function (1024 threads): particle sums of
// {released, activated, bad} + the solve's failure flag -> the decision the host used to take after reading them back.
// Called by k_as_ctl (kernels_as.hip: a launch of its own) and by the extra block of k_cons_small (kernels_generic.hip: the
// decision about round r rides in round r + 1's consensus-partials launch — one launch less per round).
#pragma once
#include "pmpc_dev.h"

namespace {

__device__ inline void as_ctl_block(AsCtl *ctl, const int *cnt_part, int M, const int *fail, int reduce, int decide, int last_of_batch,
                                                 AsCtl *mirror, unsigned long long *mirror_seq, unsigned long long seq, double *tail, const double *viol,
                                                 const int *open_part = nullptr) {
  __shared__ int sh[4][1024];
  __shared__ double shv[1024];
  if (ctl->done) {  // nothing ran in this round: republish (the host may be waiting on this sequence number)
    if (threadIdx.x == 0 && decide && last_of_batch && mirror) {
      *mirror = *ctl;
      __threadfence_system();
      *(volatile unsigned long long *)mirror_seq = seq;
    }
    return;
  }
  if (reduce) {
    int r = 0, d = 0, b = 0, op = 0;
    double vw = 0.0;
    for (int i = threadIdx.x; i < M; i += 1024) {
      r += cnt_part[3 * i]; d += cnt_part[3 * i + 1]; b |= cnt_part[3 * i + 2];
      if (viol) vw = fmax(vw, viol[i]);
      if (open_part) op += open_part[i];
    }
    sh[0][threadIdx.x] = r; sh[1][threadIdx.x] = d; sh[2][threadIdx.x] = b; sh[3][threadIdx.x] = op; shv[threadIdx.x] = vw;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) {
      if (threadIdx.x < o) {
        sh[0][threadIdx.x] += sh[0][threadIdx.x + o];
        sh[1][threadIdx.x] += sh[1][threadIdx.x + o];
        sh[2][threadIdx.x] |= sh[2][threadIdx.x + o];
        sh[3][threadIdx.x] += sh[3][threadIdx.x + o];
        shv[threadIdx.x] = fmax(shv[threadIdx.x], shv[threadIdx.x + o]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0 && viol && ctl->round < 16) ctl->worst[ctl->round] = shv[0];  // (local to this rank when sharded: a diagnostic)
    if (threadIdx.x == 0) {
      if (tail) { tail[0] = sh[0][0]; tail[1] = sh[1][0]; tail[2] = sh[2][0]; tail[3] = *fail; tail[4] = sh[3][0]; }
      else { ctl->cnt[0] = sh[0][0]; ctl->cnt[1] = sh[1][0]; ctl->cnt[2] = sh[2][0]; ctl->cnt[3] = *fail; ctl->open = sh[3][0]; }
    }
  }
  if (threadIdx.x == 0 && decide) {
    if (tail) {  // all-reduced sums (exact in fp64: small integers)
      for (int k = 0; k < 4; k++) ctl->cnt[k] = (int)(tail[k] < 2e9 ? tail[k] : 2e9);
      ctl->open = (int)(tail[4] < 2e9 ? tail[4] : 2e9);
    }
    const int rel = ctl->cnt[0], add = ctl->cnt[1], bad = ctl->cnt[2], fl = ctl->cnt[3];
    const int round = ctl->round;  // rounds completed before this one
    if (round < 16) { ctl->hist[round][0] = rel; ctl->hist[round][1] = add; }
    const int changes = rel + add;
    int done = 0, status = 1;
    if (bad || fl) { done = 1; status = 2; }
    else if (changes == 0 && ctl->open == 0) { done = 1; status = 0; }
    else if (changes == 0) {  // the set stands; the boundary cones' Newton iteration is still converging (quadratically): go on
      if (round + 1 >= ctl->max_rounds) done = 1;
    } else {
      if (changes * 2 > ctl->last_changes && ++ctl->stalls >= ctl->stall_limit) done = 1;  // not contracting: leave it to the interior-point iteration
      ctl->last_changes = changes;
      if (round + 1 >= ctl->max_rounds) done = 1;
    }
    ctl->round = round + 1;
    ctl->status = status;
    // anti-cycling on (nearly) degenerate boxes: the sign tolerance of the multipliers widens tenfold per round after the
    // fourth, up to 1e-8 of the dual scale
    const int e = ctl->round - 3 > 0 ? ctl->round - 3 : 0;
    double tl = 1e-11;
    for (int k = 0; k < e && tl < 1e-8; k++) tl *= 10.0;
    ctl->tol_l = ctl->dual_scale * (tl < 1e-8 ? tl : 1e-8);
    __threadfence();
    ctl->done = done;
    if ((done || last_of_batch) && mirror) {
      *mirror = *ctl;
      __threadfence_system();
      *(volatile unsigned long long *)mirror_seq = seq;
    }
  }
}

// launch order of a later round: the unsettled particles first, index order inside both groups (one 1024-thread block; k_as_perm, or
// behind as_ctl_block in the extra block of k_cons_small — then the round's factor sweep, which runs before that launch, still has
// the order of the round before: any permutation is a correct one, and the unsettled set changes little from round to round)
__device__ inline void as_perm_block(const int *settled, int M, int *perm) {
  __shared__ int cnt[1024];
  const int t = threadIdx.x, per = (M + 1023) / 1024, lo = t * per, hi = min(M, lo + per);
  int n = 0;
  for (int i = lo; i < hi; i++) n += settled[i] ? 0 : 1;
  cnt[t] = n;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {  // inclusive scan
    const int v = t >= o ? cnt[t - o] : 0;
    __syncthreads();
    cnt[t] += v;
    __syncthreads();
  }
  const int total = cnt[1023];
  int pu = cnt[t] - n, ps = total + lo - pu;  // first slots of this thread's unsettled / settled particles
  for (int i = lo; i < hi; i++) {
    if (settled[i]) perm[ps++] = i;
    else perm[pu++] = i;
  }
}

}  // namespace
