// Host-compiled check of the device bodies in panovlm_amd/csrc/pvlm_ba_core.h / pvlm_reproj.h: the same
// per-point / per-observation functions the HIP kernels call, driven by serial loops, so that the closed-form
// Jacobian and the Schur algebra can be compared with the oracle on a machine without a GPU
// (tests/test_reproj_cpu.py).  TEST INFRASTRUCTURE ONLY — libpvlm.so has no host path.
#include <algorithm>
#include <cstring>

#define PVLM_HD
#define PVLM_ATOMIC_ADD(ptr, v) (*(ptr) += (v))
#define PVLM_ATOMIC_MAXPOS(ptr, v) (*(ptr) = std::max(*(ptr), (v)))
#include "../../panovlm_amd/csrc/pvlm_ba_core.h"

extern "C" {

struct chk_view {
  int n_points, n_cams, n_upairs; long long n_obs;
  const long long* pt_off; const int* cam; const int* obs_pt; const double* s; const double* X; double* Xc; double* scale; double* Vinv; double* gp;
  const int* adj_off; const int* adj_cam; const int* adj_slot; const unsigned char* frozen; double w; int loss; double a;
};

static pvlm_ba::View to_view(const chk_view* c) {
  pvlm_ba::View v;
  v.n_points = c->n_points; v.n_cams = c->n_cams; v.n_upairs = c->n_upairs; v.n_obs = c->n_obs; v.pt_off = c->pt_off; v.cam = c->cam;
  v.obs_pt = c->obs_pt; v.s = c->s; v.X = c->X; v.Xc = c->Xc; v.scale = c->scale; v.Vinv = c->Vinv; v.gp = c->gp; v.adj_off = c->adj_off;
  v.adj_cam = c->adj_cam; v.adj_slot = c->adj_slot; v.frozen = c->frozen; v.w = c->w; v.loss = c->loss; v.a = c->a;
  return v;
}

long long chk_packed_size(int n_cams, int n_upairs) { return pvlm_ba::packed_size(n_cams, n_upairs); }

void chk_eval(const chk_view* c, const double* pose_tab, double* r, double* J9) {
  const pvlm_ba::View v = to_view(c);
  for (long long i = 0; i < v.n_obs; ++i) {
    double Jc[6], Jp[3];
    pvlm_reproj::eval_obs(pose_tab + (size_t)v.cam[i] * PVLM_BA_POSE_TAB, v.X + 3 * (size_t)v.obs_pt[i], v.s + 3 * i, v.w, r + i, Jc, Jp);
    std::memcpy(J9 + 9 * i, Jc, 48); std::memcpy(J9 + 9 * i + 6, Jp, 24);
  }
}

void chk_reduce(const chk_view* c, const double* pose_tab, int init_scale, double radius, double min_diag, double max_diag, double* packed) {
  const pvlm_ba::View v = to_view(c);
  const long long psz = pvlm_ba::packed_size(v.n_cams, v.n_upairs);
  std::fill(packed, packed + psz, 0.0);
  for (int p = 0; p < v.n_points; ++p) pvlm_ba::point_pass(v, pose_tab, p, init_scale, radius, min_diag, max_diag, packed + psz - 1);
  double* cost = packed + (size_t)v.n_cams * 42 + (size_t)v.n_upairs * 36;
  for (long long i = 0; i < v.n_obs; ++i) *cost += pvlm_ba::obs_pass(v, pose_tab, i, packed);
}

void chk_step(const chk_view* c, const double* pose_tab, const double* dcam, double* out3) {
  const pvlm_ba::View v = to_view(c);
  out3[0] = out3[1] = out3[2] = 0.0;
  for (int p = 0; p < v.n_points; ++p) { double o[3]; pvlm_ba::step_point(v, pose_tab, p, dcam, o); for (int k = 0; k < 3; ++k) out3[k] += o[k]; }
}

double chk_cost(const chk_view* c, const double* pose_tab, int candidate) {
  const pvlm_ba::View v = to_view(c);
  double s = 0.0;
  for (long long i = 0; i < v.n_obs; ++i) s += pvlm_ba::cost_obs(v, pose_tab, i, candidate);
  return s;
}

}  // extern "C"
