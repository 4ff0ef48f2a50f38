// Device-side closed-form residual + gradient of the six hot functors of base/CostFunction.h.
//
// Every functor of the reference evaluates   P_r = R(aa_rw) R(-aa_nw) (P_n - t_nw) + t_rw
// (base/CostFunction.h:585-604 for the point functors, :369-403 / :466-486 for the two plane
// functors, whose via-world chain is the same map) and then a scalar residual r(P_r).  With
// g = dr/dP_r, m = P_r - t_rw and the left Jacobian J_l of SO(3) the AutoDiff row is
//   d r/d aa_rw = (m x g)^T J_l(aa_rw)        d r/d t_rw = g^T
//   d r/d aa_nw = (m x g)^T (-R_rn J_l(aa_nw))   d r/d t_nw = g^T (-R_rn)
// so each functor only has to produce r and the "wrench" v = [c = sum m_k x g_k ; g = sum g_k]
// (one term per transformed point).  Branches follow the VALUE of the residual exactly like
// ceres::Jet does (abs'(x) = x<0 ? -1 : +1, clamped acos returns a constant, early-outs return 0).
#pragma once
#include <hip/hip_runtime.h>

#include "pvlm_internal.h"

namespace pvlm_dev {

struct Wrench { double r; double c[3]; double g[3]; };

__device__ __forceinline__ double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = a[2] * b[0] - a[0] * b[2]; o[2] = a[0] * b[1] - a[1] * b[0];
}


// ---- fp64 helpers for the bandwidth-bound kernels ------------------------------------------------
// The compiler's IEEE-exact fp64 division / sqrt expand to 12-18 instructions each (scaling,
// fix-ups for denormals and specials).  The kernels only see well-scaled operands (metres,
// radians), so the hardware estimate + ONE Newton step is used instead.  Measured on MI355X
// (tools/micro/rcp_precision.hip): v_rcp_f64 4.6e-8, +1 step 2.2e-15; v_rsq_f64 5.2e-8, +1 step
// 4.2e-15 max relative error — nine orders of magnitude inside the 1e-6 parity tolerance.
__device__ __forceinline__ double fast_rcp(double x) {
  const double r = __builtin_amdgcn_rcp(x);
  return fma(r, fma(-x, r, 1.0), r);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  const double r = __builtin_amdgcn_rsq(x);
  return fma(r, fma(-x * r, 0.5 * r, 0.5), r);
}
// atan(t) for |t| <= tan(pi/8): degree-10 polynomial in t^2 (Chebyshev-node interpolant of
// atan(t)/t on [0, tan^2(pi/8)], max relative error 2.2e-16 measured against mpmath).
// A Horner step p * u + c costs ONE v_fma_f64 when the coefficient sits in a scalar register pair; written as fma(p, u, literal) the
// compiler keeps the ten literals in VGPRs and pays a 64-bit move per step to feed the two-address v_fmac_f64 (ISA of round 3:
// 20 of the ~330 VALU instructions of a two-row iteration) — hence the operand constraint below.  Same instruction, same bits.
__device__ __forceinline__ double fma_sconst(double p, double u, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(p), "v"(u), "s"(c));
  return d;
#else
  return fma(p, u, c);
#endif
}
__device__ __forceinline__ double atan_small(double t) {
  const double u = t * t;
  double p = 2.11272689568591313e-02;
  p = fma_sconst(p, u, -4.34739031566048761e-02);
  p = fma_sconst(p, u, 5.68812005923421543e-02);
  p = fma_sconst(p, u, -6.64019012732186553e-02);
  p = fma_sconst(p, u, 7.68994845277858746e-02);
  p = fma_sconst(p, u, -9.09077271667437237e-02);
  p = fma_sconst(p, u, 1.11111061650365134e-01);
  p = fma_sconst(p, u, -1.42857141805968924e-01);
  p = fma_sconst(p, u, 1.99999999988503901e-01);
  p = fma_sconst(p, u, -3.33333333333284132e-01);
  p = fma(p, u, 1.0);
  return t * p;
}
// atan2(y, x) for y >= 0.  When every lane of the wave has its angle below pi/8 (the normal ICP
// regime) one reciprocal and the polynomial suffice; otherwise octant folding as in FastAtan2
// (min/max) and the reduction atan(a) = pi/4 + atan((a-1)/(a+1)) are applied, branch-free.
__device__ __forceinline__ double atan2_pos(double y, double x) {
  if (__all(y <= 0.41421356237309503 * x)) return atan_small(y * fast_rcp(x));
  const double ax = fabs(x);
  const double mn = fmin(y, ax), mx = fmax(y, ax);
  const double a = mn * fast_rcp(mx);
  const bool big = a > 0.41421356237309503;
  const double red = (a - 1.0) * fast_rcp(a + 1.0);
  double r = atan_small(big ? red : a);
  r = big ? r + 0.78539816339744830962 : r;
  r = (y > ax) ? 1.57079632679489661923 - r : r;
  return (x < 0.0) ? 3.14159265358979323846 - r : r;
}

// VectorAngle3D (base/Geometry.hpp:450-466), un-normalised form, with gradients wrt v1, v2.
// Returns false when the clamped branches (constant result, zero gradient) were taken — decided on the cosine like
// upstream.  The VALUE is atan2(|v1 x v2|, v1.v2) and the gradients are (v1 x w)/(|v1|^2 |w|), -(v2 x w)/(|v2|^2 |w|)
// with w = v1 x v2: the same angle and derivatives as acos of the normalised dot product, without its loss of eps / r
// near r -> 0 (upstream's double evaluation is the less accurate side there; tests/test_eval_gpu.py compares both with
// an extended-precision evaluation of upstream's statements).
__device__ __forceinline__ double atan2_pos(double y, double x);
__device__ __forceinline__ bool angle_between(const double* v1, const double* v2, double& r, double* g1, double* g2) {
  const double d = dot3(v1, v2);
  const double q1 = dot3(v1, v1), q2 = dot3(v2, v2);
  const double n1 = sqrt(q1), n2 = sqrt(q2);
  const double c = d / (n1 * n2);
  if (c >= 1.0) { r = 0.0; return false; }
  if (c <= -1.0) { r = M_PI; return false; }
  double w[3], a1[3], a2[3];
  cross3(v1, v2, w);
  const double nw = sqrt(dot3(w, w));
  if (!(nw > 0.0)) { r = d < 0.0 ? M_PI : 0.0; return false; }
  r = atan2_pos(nw, d);
  cross3(v1, w, a1);
  cross3(v2, w, a2);
  const double k1 = 1.0 / (q1 * nw), k2 = -1.0 / (q2 * nw);
#pragma unroll
  for (int k = 0; k < 3; ++k) { g1[k] = k1 * a1[k]; g2[k] = k2 * a2[k]; }
  return true;
}

// Tail shared by Point2Plane_Angle / Point2Line_Angle (base/CostFunction.h:695-717, :899-920):
// r = angle(P, Pp) or, with normalize_distance, the angle seen from the centre placed 1 m
// before Pp on the ray O->Pp.  Outputs gradients wrt P (direct) and wrt Pp.
template <bool NORM>
__device__ __forceinline__ bool normalized_angle(const double* P, const double* Pp, double& r, double* gP, double* gPp) {
  if (NORM) {
    const double nu2 = dot3(Pp, Pp);
    const double nu = sqrt(nu2);
    const double ratio = (nu - 1.0) / nu;
    double v1[3], v2[3], g1[3], g2[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { const double ck = ratio * Pp[k]; v1[k] = Pp[k] - ck; v2[k] = P[k] - ck; }
    if (!angle_between(v1, v2, r, g1, g2)) return false;
    // c = ratio(Pp) * Pp, d ratio / d Pp = Pp / nu^3 ; cotangent of c is -(g1+g2)
    double gc[3] = {-(g1[0] + g2[0]), -(g1[1] + g2[1]), -(g1[2] + g2[2])};
    const double s = dot3(Pp, gc) / (nu2 * nu);
#pragma unroll
    for (int k = 0; k < 3; ++k) { gP[k] = g2[k]; gPp[k] = g1[k] + ratio * gc[k] + s * Pp[k]; }
    return true;
  }
  return angle_between(P, Pp, r, gP, gPp);
}

// ---- single-point functors: r and g = dr/dP_r -------------------------------------------------
// rec = SoA row (see pvlm_ctx.hip for the column layout)
template <int KIND, bool NORM>
__device__ __forceinline__ void residual_grad(const double* P, const double* rec, double w, double& r, double* g) {
  g[0] = g[1] = g[2] = 0.0;
  r = 0.0;
  if (KIND == PVLM_POINT2PLANE_METER) {
    // base/CostFunction.h:606-607, Geometry.hpp:275-283 (normalized=true)
    const double* n = rec + 3;
    const double sd = n[0] * P[0] + n[1] * P[1] + n[2] * P[2] + n[3];
    const double s = sd < 0.0 ? -w : w;
    r = s * sd;
    g[0] = s * n[0]; g[1] = s * n[1]; g[2] = s * n[2];
  } else if (KIND == PVLM_POINT2PLANE_ANGLE && NORM) {
    // base/CostFunction.h:679-715 with normalize_distance (the Room/Floor configuration), reduced
    // algebraically.  Pp = P - sd n is the foot of P on the plane (both branches of the reference's
    // sign test give this point), nu = |Pp|, the centre sits 1 m before Pp on the ray O->Pp, so
    // vec1 = Pp/nu =: e (unit) and vec2 = e + sd n.  With u = Pp + d n (the in-plane part of Pp,
    // |u| = q) and n.Pp = -d:   e.vec2 = 1 - sd d/nu =: x,   |e x vec2| = |sd| q/nu =: y,
    // r = angle(vec1, vec2) = atan2(y, x) — the same angle the reference obtains from acos of the
    // normalised dot product, without its cancellation near cos = 1.  r depends on P through
    // (sd, nu) only: grad sd = n, grad nu = u/nu.
    const double* n = rec + 3;
    const double dd = n[3];
    const double sd = n[0] * P[0] + n[1] * P[1] + n[2] * P[2] + dd;
    const double s = sd < 0.0 ? -1.0 : 1.0;
    const double dis = s * sd;
    const double u[3] = {P[0] - (sd - dd) * n[0], P[1] - (sd - dd) * n[1], P[2] - (sd - dd) * n[2]};
    const double q2 = dot3(u, u);
    const double nu2 = q2 + dd * dd;
    const double inv_nu = fast_rsqrt(nu2);
    const double invq = q2 > 0.0 ? fast_rsqrt(q2) : 0.0;
    const double q = q2 * invq;
    const double x = 1.0 - sd * dd * inv_nu, y = dis * q * inv_nu;
    const double invD = fast_rcp(x * x + y * y);
    const bool live = dis >= 1e-3;  // dis < 1e-3 -> residual 0, zero Jacobian (CostFunction.h:680-684)
    r = live ? atan2_pos(y, x) : 0.0;
    const double k0 = live ? inv_nu * invD : 0.0;
    const double fsd = (x * s * q + y * dd) * k0;
    const double cu = (x * dis * dd * dd * invq - y * sd * dd) * (inv_nu * inv_nu) * k0;
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = fsd * n[k] + cu * u[k];
  } else if (KIND == PVLM_POINT2PLANE_ANGLE) {
    // base/CostFunction.h:679-717 (weight is NOT applied by the reference)
    const double* n = rec + 3;
    const double sd = n[0] * P[0] + n[1] * P[1] + n[2] * P[2] + n[3];
    const double s = sd < 0.0 ? -1.0 : 1.0;
    const double dis = s * sd;
    if (dis < 1e-3) return;
    double Pp[3] = {P[0] - dis * n[0], P[1] - dis * n[1], P[2] - dis * n[2]};
    double sig = 1.0;
    if (fabs(n[0] * Pp[0] + n[1] * Pp[1] + n[2] * Pp[2] + n[3]) > 1e-4) {
      Pp[0] = P[0] + dis * n[0]; Pp[1] = P[1] + dis * n[1]; Pp[2] = P[2] + dis * n[2];
      sig = -1.0;
    }
    double gP[3], gPp[3];
    if (!normalized_angle<NORM>(P, Pp, r, gP, gPp)) return;
    // Pp = P - sig*s*(n.P+d) n  ->  dPp/dP = I - sig*s n n^T
    const double t = sig * s * dot3(n, gPp);
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = gP[k] + gPp[k] - t * n[k];
  } else if (KIND == PVLM_POINT2LINE_METER) {
    // base/CostFunction.h:813-815, Geometry.hpp:198-211
    const double* A = rec + 3; const double* d = rec + 6;
    const double dd = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
    const double k = (d[0] * (P[0] - A[0]) + d[1] * (P[1] - A[1]) + d[2] * (P[2] - A[2])) / dd;
    const double e[3] = {k * d[0] + A[0] - P[0], k * d[1] + A[1] - P[1], k * d[2] + A[2] - P[2]};
    const double dist = sqrt(dot3(e, e));
    r = w * dist;
    // d dist/dP = (e/dist)^T (d d^T/dd - I)
    const double q = dot3(d, e) / dd;
    const double wi = w / dist;
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) g[kk] = wi * (q * d[kk] - e[kk]);
  } else if (KIND == PVLM_POINT2LINE_ANGLE) {
    // base/CostFunction.h:881-920 (weight NOT applied)
    const double* A = rec + 3; const double* d = rec + 6;
    const double k = d[0] * (P[0] - A[0]) + d[1] * (P[1] - A[1]) + d[2] * (P[2] - A[2]);
    const double Pp[3] = {k * d[0] + A[0], k * d[1] + A[1], k * d[2] + A[2]};
    const double e[3] = {P[0] - Pp[0], P[1] - Pp[1], P[2] - Pp[2]};
    const double dis = sqrt(dot3(e, e));
    if (dis < 1e-3) return;
    double gP[3], gPp[3];
    if (!normalized_angle<NORM>(P, Pp, r, gP, gPp)) return;
    const double t = dot3(d, gPp);  // dPp/dP = d d^T
#pragma unroll
    for (int kk = 0; kk < 3; ++kk) g[kk] = gP[kk] + t * d[kk];
  }
}

// Full wrench for any kind.  T = pair table row (R_rn[9] | t_rn[3] | t_rw[3] | ...).
template <int KIND, bool NORM>
__device__ __forceinline__ void eval_wrench(const double* rec, const double* T, double w, Wrench& out) {
  const double* R = T; const double* trn = T + 9; const double* trw = T + 12;
  auto chain = [&](const double* p, double* o) {
    o[0] = R[0] * p[0] + R[1] * p[1] + R[2] * p[2] + trn[0];
    o[1] = R[3] * p[0] + R[4] * p[1] + R[5] * p[2] + trn[1];
    o[2] = R[6] * p[0] + R[7] * p[1] + R[8] * p[2] + trn[2];
  };
  if (KIND <= PVLM_POINT2LINE_ANGLE) {
    double P[3];
    chain(rec, P);
    residual_grad<KIND, NORM>(P, rec, w, out.r, out.g);
    const double m[3] = {P[0] - trw[0], P[1] - trw[1], P[2] - trw[2]};
    cross3(m, out.g, out.c);
  } else if (KIND == PVLM_PLANE2PLANE_GLOBAL) {
    // base/CostFunction.h:405-412, Geometry.hpp:471-485 ; rec = [n_img(3) a(3) b(3) w]
    const double* ni = rec; const double wk = rec[9];
    double a[3], b[3], nrm[3];
    chain(rec + 3, a);
    chain(rec + 6, b);
    cross3(a, b, nrm);
    out.r = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { out.c[k] = 0.0; out.g[k] = 0.0; }
    const double dp = dot3(ni, nrm);
    const double s = dp < 0.0 ? -1.0 : 1.0;
    const double q1 = dot3(ni, ni), q2 = dot3(nrm, nrm);
    const double c = s * dp / (sqrt(q1) * sqrt(q2));
    if (c >= 1.0) return;                      // PlaneAngle's clamp (Geometry.hpp:480-481), decided on the cosine like upstream
    // value and gradient in the cross-product form (see angle_between): theta = atan2(|ni x n|, |ni . n|),
    // d theta / d n = -s (n x w) / (|n|^2 |w|), w = ni x n
    double wv[3], nxw[3];
    cross3(ni, nrm, wv);
    const double nw = sqrt(dot3(wv, wv));
    if (!(nw > 0.0)) return;
    out.r = wk * atan2_pos(nw, s * dp);
    cross3(nrm, wv, nxw);
    const double kg = -wk * s / (q2 * nw);
    double gn[3], ga[3], gb[3], ca[3], cb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) gn[k] = kg * nxw[k];
    cross3(b, gn, ga);   // d(a x b) . gn  wrt a
    cross3(gn, a, gb);   // wrt b
    const double ma[3] = {a[0] - trw[0], a[1] - trw[1], a[2] - trw[2]};
    const double mb[3] = {b[0] - trw[0], b[1] - trw[1], b[2] - trw[2]};
    cross3(ma, ga, ca);
    cross3(mb, gb, cb);
#pragma unroll
    for (int k = 0; k < 3; ++k) { out.c[k] = ca[k] + cb[k]; out.g[k] = ga[k] + gb[k]; }
  } else {  // PVLM_PLANE_IOU — base/CostFunction.h:488-497 ; rec = [plane(4) mid_n(3) mid_r(3) angle w]
    const double* n = rec; const double* mr = rec + 7; const double ang = rec[10]; const double wk = rec[11];
    double mc[3];
    chain(rec + 4, mc);
    out.r = 0.0;
#pragma unroll
    for (int k = 0; k < 3; ++k) { out.c[k] = 0.0; out.g[k] = 0.0; }
    const double sd = n[0] * mc[0] + n[1] * mc[1] + n[2] * mc[2] + n[3];
    const double s = sd < 0.0 ? -1.0 : 1.0;
    const double dis = s * sd;
    double np[3] = {mc[0] - dis * n[0], mc[1] - dis * n[1], mc[2] - dis * n[2]};
    double sig = 1.0;
    if (fabs(n[0] * np[0] + n[1] * np[1] + n[2] * np[2] + n[3]) > 1e-4) {
      np[0] = mc[0] + dis * n[0]; np[1] = mc[1] + dis * n[1]; np[2] = mc[2] + dis * n[2];
      sig = -1.0;
    }
    double cur, g1[3], g2[3];
    const bool smooth = angle_between(np, mr, cur, g1, g2);
    if (cur < ang) return;
    out.r = wk * (cur - ang);
    if (!smooth) return;
    const double t = sig * s * dot3(n, g1);
    double g[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) g[k] = wk * (g1[k] - t * n[k]);
    const double m[3] = {mc[0] - trw[0], mc[1] - trw[1], mc[2] - trw[2]};
    cross3(m, g, out.c);
#pragma unroll
    for (int k = 0; k < 3; ++k) out.g[k] = g[k];
  }
}

}  // namespace pvlm_dev
