// Per-point / per-observation bodies of the reprojection ("bundle") kernels: linearisation of the
// PanoramaReprojResidual_1Angle blocks, elimination of the 3-D points (Schur complement onto the camera
// poses), back-substitution and cost.  Reference: AddCameraResidual (util/Optimization.cpp:172-222) inside
// CameraLidarOptimizer::Optimize (joint_optimization/CameraLidarOptimizer.cpp:431-432) solved by Ceres with
// SPARSE/DENSE_SCHUR (util/Optimization.cpp:608-634), i.e. the points are eliminated exactly like here.
//
// The bodies take an index and plain pointers; pvlm_ba.hip wraps each in a __global__ kernel (one thread per
// point / observation, fp64 atomics into the packed camera system).  tests/cpp/reproj_math_check.cpp compiles
// the same bodies for the host (serial loop, PVLM_ATOMIC_ADD = plain +=) to check the algebra against the
// oracle's Jacobians without a GPU — the product itself has no host path.
#pragma once
#include "pvlm_reproj.h"

#ifndef PVLM_ATOMIC_ADD
#define PVLM_ATOMIC_ADD(ptr, v) unsafeAtomicAdd((ptr), (v))
#endif
#ifndef PVLM_ATOMIC_MAXPOS  // max of non-negative doubles (IEEE bit patterns are monotone for x >= 0)
#define PVLM_ATOMIC_MAXPOS(ptr, v) atomicMax(reinterpret_cast<unsigned long long*>(ptr), (unsigned long long)__double_as_longlong(v))
#endif

#define PVLM_BA_POSE_TAB 21  // == PVLM_POSE_TAB

namespace pvlm_ba {

struct View {
  int n_points, n_cams, n_upairs;
  long long n_obs;
  const long long* pt_off;  // n_points + 1 : observations of point p are [pt_off[p], pt_off[p+1])
  const int* cam;           // n_obs : pose-table row of the observing camera
  const int* obs_pt;        // n_obs : point of each observation
  const double* s;          // n_obs x 3 unit bearings
  const double* X;          // n_points x 3 current points
  double* Xc;               // n_points x 3 candidate points (written by step_point)
  double* scale;            // n_points x 3 Jacobi scaling of the point columns (set when init_scale)
  double* Vinv;             // n_points x 6 inverse of the damped point block (0 = point frozen)
  double* gp;               // n_points x 3 gradient of the point block
  const int* adj_off;       // n_cams + 1 : CSR of co-visible cameras cj > ci
  const int* adj_cam;
  const int* adj_slot;      // index of the unordered pair (ci, cj) in the packed off-diagonal blocks
  const unsigned char* frozen;  // n_points or null: 1 = constant parameter block (SetParameterBlockConstant)
  double w;                 // residual weight (config.camera_weight)
  int loss;                 // 0 none, 1 Huber
  double a;
};

// packed camera system (doubles):
//   [Hdiag n_cams x 36 | Hoff n_upairs x 36 | g n_cams x 6 | cost | Udiag n_cams x 6 | gcam n_cams x 6 | gmax_points]
PVLM_HD inline long long packed_size(int n_cams, int n_upairs) { return (long long)n_cams * 54 + (long long)n_upairs * 36 + 2; }

struct Lin { double r, rho, rho1, Jc[6], Jp[3]; };

PVLM_HD inline void linearise(const View& v, const double* pose_tab, long long i, const double* X, Lin* o) {
  pvlm_reproj::eval_obs(pose_tab + (size_t)v.cam[i] * PVLM_BA_POSE_TAB, X, v.s + 3 * i, v.w, &o->r, o->Jc, o->Jp);
  pvlm_reproj::loss_eval(v.loss, v.a, o->r * o->r, &o->rho, &o->rho1);
}

// ---- pass A, one call per point: V = sum rho' Jp^T Jp, gp = sum rho' Jp r, damped inverse ---------------
PVLM_HD inline void point_pass(const View& v, const double* pose_tab, int p, int init_scale, double radius, double min_diag, double max_diag,
                               double* gmax) {
  const double* X = v.X + 3 * (size_t)p;
  double V[6] = {0, 0, 0, 0, 0, 0}, g[3] = {0, 0, 0};
  for (long long i = v.pt_off[p]; i < v.pt_off[p + 1]; ++i) {
    Lin l; linearise(v, pose_tab, i, X, &l);
    const double a0 = l.rho1 * l.Jp[0], a1 = l.rho1 * l.Jp[1], a2 = l.rho1 * l.Jp[2];
    V[0] += a0 * l.Jp[0]; V[1] += a0 * l.Jp[1]; V[2] += a0 * l.Jp[2];
    V[3] += a1 * l.Jp[1]; V[4] += a1 * l.Jp[2]; V[5] += a2 * l.Jp[2];
    g[0] += a0 * l.r; g[1] += a1 * l.r; g[2] += a2 * l.r;
  }
  double* sc = v.scale + 3 * (size_t)p;
  if (init_scale) { sc[0] = 1.0 / (1.0 + sqrt(V[0])); sc[1] = 1.0 / (1.0 + sqrt(V[3])); sc[2] = 1.0 / (1.0 + sqrt(V[5])); }
  double Vd[6], inv[6];
  pvlm_reproj::damp3(V, sc, radius, min_diag, max_diag, Vd);
  const bool frozen = v.frozen && v.frozen[p];
  if (frozen || !pvlm_reproj::spd3_inverse(Vd, inv)) { for (int k = 0; k < 6; ++k) inv[k] = 0.0; }   // Vinv = 0: not eliminated, does not move
  for (int k = 0; k < 6; ++k) v.Vinv[6 * (size_t)p + k] = inv[k];
  for (int k = 0; k < 3; ++k) v.gp[3 * (size_t)p + k] = g[k];
  const double m = fmax(fabs(g[0]), fmax(fabs(g[1]), fabs(g[2])));
  if (m > 0.0 && !frozen) PVLM_ATOMIC_MAXPOS(gmax, m);
}

PVLM_HD inline int find_slot(const View& v, int ci, int cj) {  // cj > ci; -1 if the pair is not in the structure
  int lo = v.adj_off[ci], hi = v.adj_off[ci + 1] - 1;
  while (lo <= hi) {
    const int mid = (lo + hi) >> 1;
    const int c = v.adj_cam[mid];
    if (c == cj) return v.adj_slot[mid];
    if (c < cj) lo = mid + 1; else hi = mid - 1;
  }
  return -1;
}

// ---- pass B, one call per observation i: its row of the reduced camera system ------------------------------
//   S(ci,ci) += rho' Jc_i^T Jc_i - T_i W_i^T ;  S(ci,cj) -= T_i W_j^T (j != i) ;  g(ci) += rho' Jc_i r_i - T_i gp
//   with W_i = rho' Jc_i^T Jp_i (6x3), T_i = W_i Vinv (6x3).  Returns rho_i / 2 (block cost).
PVLM_HD inline double obs_pass(const View& v, const double* pose_tab, long long i, double* packed) {
  const int p = v.obs_pt[i];
  const double* X = v.X + 3 * (size_t)p;
  const double* Vi = v.Vinv + 6 * (size_t)p;
  Lin li; linearise(v, pose_tab, i, X, &li);
  const int ci = v.cam[i];
  double* Hd = packed + (size_t)ci * 36;
  double* Ho = packed + (size_t)v.n_cams * 36;
  double* g = Ho + (size_t)v.n_upairs * 36 + (size_t)ci * 6;
  double* Ud = Ho + (size_t)v.n_upairs * 36 + (size_t)v.n_cams * 6 + 1 + (size_t)ci * 6;
  double* gc = Ud + (size_t)v.n_cams * 6;
  // y = Vinv Jp_i ;  T_i = rho' Jc_i y^T  (rank one)
  double y[3]; pvlm_reproj::sym3_mul(Vi, li.Jp, y);
  const double* gp = v.gp + 3 * (size_t)p;
  const double ygp = y[0] * gp[0] + y[1] * gp[1] + y[2] * gp[2];
  for (int k = 0; k < 6; ++k) {
    PVLM_ATOMIC_ADD(&g[k], li.rho1 * li.Jc[k] * (li.r - ygp));
    PVLM_ATOMIC_ADD(&Ud[k], li.rho1 * li.Jc[k] * li.Jc[k]);
    PVLM_ATOMIC_ADD(&gc[k], li.rho1 * li.Jc[k] * li.r);
  }
  for (long long j = v.pt_off[p]; j < v.pt_off[p + 1]; ++j) {
    const int cj = v.cam[j];
    if (cj < ci) continue;
    Lin lj;
    if (j == i) lj = li; else linearise(v, pose_tab, j, X, &lj);
    // -T_i W_j^T = -(rho'_i rho'_j (y . Jp_j)) Jc_i Jc_j^T ; the j == i term also carries +rho'_i Jc_i Jc_i^T
    double coef = -li.rho1 * lj.rho1 * (y[0] * lj.Jp[0] + y[1] * lj.Jp[1] + y[2] * lj.Jp[2]);
    if (j == i) coef += li.rho1;
    double* dst;
    if (cj == ci) dst = Hd;
    else { const int slot = find_slot(v, ci, cj); if (slot < 0) continue; dst = Ho + (size_t)slot * 36; }
    for (int a = 0; a < 6; ++a) {
      const double ca = coef * li.Jc[a];
      for (int b = 0; b < 6; ++b) PVLM_ATOMIC_ADD(&dst[a * 6 + b], ca * lj.Jc[b]);
    }
  }
  return 0.5 * li.rho;
}

// ---- pass B as a GATHER (round 2): one (i, j) couple of observations of the same point, cam[i] <= cam[j], contributes
//   coef Jc_i Jc_j^T to the block of the camera pair (cam[i], cam[j]) — coef as in obs_pass — and, when j == i, the
//   observation's share of g / Udiag / gcam / cost.  pvlm_ba.hip lists the couples of every block once (pvlm_ba_create) and
//   one wave sums a block's couples in a fixed order: no atomics, and the same bits on every run.
//   acc: 36 doubles (row-major 6 x 6, rows = camera of i), vec: g (6) | Udiag (6) | gcam (6) | cost (1), touched when j == i.
PVLM_HD inline void couple_pass(const View& v, const double* pose_tab, long long i, long long j, double* acc, double* vec) {
  const int p = v.obs_pt[i];
  const double* X = v.X + 3 * (size_t)p;
  const double* Vi = v.Vinv + 6 * (size_t)p;
  Lin li, lj;
  linearise(v, pose_tab, i, X, &li);
  if (j == i) lj = li; else linearise(v, pose_tab, j, X, &lj);
  double y[3]; pvlm_reproj::sym3_mul(Vi, li.Jp, y);
  double coef = -li.rho1 * lj.rho1 * (y[0] * lj.Jp[0] + y[1] * lj.Jp[1] + y[2] * lj.Jp[2]);
  if (j == i) coef += li.rho1;
  for (int a = 0; a < 6; ++a) {
    const double ca = coef * li.Jc[a];
    for (int b = 0; b < 6; ++b) acc[a * 6 + b] += ca * lj.Jc[b];
  }
  if (j == i) {
    const double* gp = v.gp + 3 * (size_t)p;
    const double ygp = y[0] * gp[0] + y[1] * gp[1] + y[2] * gp[2];
    for (int k = 0; k < 6; ++k) {
      vec[k] += li.rho1 * li.Jc[k] * (li.r - ygp);
      vec[6 + k] += li.rho1 * li.Jc[k] * li.Jc[k];
      vec[12 + k] += li.rho1 * li.Jc[k] * li.r;
    }
    vec[18] += 0.5 * li.rho;
  }
}

// ---- back-substitution, one call per point: dp = -Vinv (gp + sum_i rho' Jp_i (Jc_i . dc_i)); Xc = X + dp -----
// dcam: n_cams x 6 camera steps (0 for constant blocks).  out3 += [model decrease, |dp|^2, |X|^2] where the model
// decrease is -sum_i rho'_i (r_i d_i + d_i^2 / 2), d_i = Jc_i.dc_i + Jp_i.dp (Gauss-Newton model of these blocks).
PVLM_HD inline void step_point(const View& v, const double* pose_tab, int p, const double* dcam, double* out3_local) {
  const double* X = v.X + 3 * (size_t)p;
  const double* Vi = v.Vinv + 6 * (size_t)p;
  double b[3] = {v.gp[3 * (size_t)p], v.gp[3 * (size_t)p + 1], v.gp[3 * (size_t)p + 2]};
  for (long long i = v.pt_off[p]; i < v.pt_off[p + 1]; ++i) {
    Lin l; linearise(v, pose_tab, i, X, &l);
    const double* dc = dcam + 6 * (size_t)v.cam[i];
    double e = 0.0;
    for (int k = 0; k < 6; ++k) e += l.Jc[k] * dc[k];
    e *= l.rho1;
    b[0] += e * l.Jp[0]; b[1] += e * l.Jp[1]; b[2] += e * l.Jp[2];
  }
  double dp[3]; pvlm_reproj::sym3_mul(Vi, b, dp);
  dp[0] = -dp[0]; dp[1] = -dp[1]; dp[2] = -dp[2];
  double model = 0.0;
  for (long long i = v.pt_off[p]; i < v.pt_off[p + 1]; ++i) {
    Lin l; linearise(v, pose_tab, i, X, &l);
    const double* dc = dcam + 6 * (size_t)v.cam[i];
    double d = l.Jp[0] * dp[0] + l.Jp[1] * dp[1] + l.Jp[2] * dp[2];
    for (int k = 0; k < 6; ++k) d += l.Jc[k] * dc[k];
    model -= l.rho1 * (l.r * d + 0.5 * d * d);
  }
  for (int k = 0; k < 3; ++k) v.Xc[3 * (size_t)p + k] = X[k] + dp[k];
  out3_local[0] = model;
  out3_local[1] = dp[0] * dp[0] + dp[1] * dp[1] + dp[2] * dp[2];
  out3_local[2] = (v.frozen && v.frozen[p]) ? 0.0 : X[0] * X[0] + X[1] * X[1] + X[2] * X[2];   // |x|^2 of the FREE parameters
}

// ---- cost only, one call per observation (candidate = 1: at Xc) ---------------------------------------------
PVLM_HD inline double cost_obs(const View& v, const double* pose_tab, long long i, int candidate) {
  const double* X = (candidate ? v.Xc : v.X) + 3 * (size_t)v.obs_pt[i];
  double r, rho, rho1;
  pvlm_reproj::eval_obs(pose_tab + (size_t)v.cam[i] * PVLM_BA_POSE_TAB, X, v.s + 3 * i, v.w, &r, nullptr, nullptr);
  pvlm_reproj::loss_eval(v.loss, v.a, r * r, &rho, &rho1);
  return 0.5 * rho;
}

}  // namespace pvlm_ba
