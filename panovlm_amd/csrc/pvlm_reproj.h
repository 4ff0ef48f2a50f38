// Closed-form residual + Jacobian of the panoramic reprojection block and the 3x3 point algebra of its
// Schur elimination.  Reference: PanoramaReprojResidual_1Angle, base/CostFunction.h:218-247 (AutoDiff over
// ceres::AngleAxisRotatePoint); added by AddCameraResidual, util/Optimization.cpp:172-222.
//
//   p = R(aa_cw) X + t_cw,  u = p/|p|,  r = w * angle(u, s)            (s = unit bearing of the keypoint)
//   dr/dp  = g = -w (s - (u.s) u) / (|p| sin(angle))
//   dr/dX  = g^T R            dr/dt_cw = g^T            dr/daa_cw = ((R X) x g)^T J_l(aa_cw)
//
// The angle is evaluated as atan2(|u x s|, u.s): the same value as the reference's acos(u.s) without its
// cancellation near 0.  At angle == 0 exactly the reference's derivative is 0 * inf; here g = 0.
// Plain precise math only (sqrt, /, atan2): the block count is 1e5..1e6, this is not the streaming kernel.
// The header is also compiled for the host by tests/cpp/reproj_math_check.cpp (PVLM_HD empty) to check
// the formulas against the CPU oracle without a GPU.
#pragma once
#include <cmath>

#ifndef PVLM_HD
#define PVLM_HD __host__ __device__
#endif

namespace pvlm_reproj {

// pose = row of the device pose table: [R row-major (9) | J_l row-major (9) | t (3)]
PVLM_HD inline void eval_obs(const double* pose, const double* X, const double* s, double w, double* r, double* Jc, double* Jp) {
  const double* R = pose; const double* Jl = pose + 9; const double* t = pose + 18;
  const double q0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double q1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double q2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double p0 = q0 + t[0], p1 = q1 + t[1], p2 = q2 + t[2];
  const double n = sqrt(p0 * p0 + p1 * p1 + p2 * p2);
  const double u0 = p0 / n, u1 = p1 / n, u2 = p2 / n;
  const double c = u0 * s[0] + u1 * s[1] + u2 * s[2];
  const double x0 = u1 * s[2] - u2 * s[1], x1 = u2 * s[0] - u0 * s[2], x2 = u0 * s[1] - u1 * s[0];
  const double sn = sqrt(x0 * x0 + x1 * x1 + x2 * x2);
  *r = w * atan2(sn, c);
  if (!Jc) return;
  double g0 = 0.0, g1 = 0.0, g2 = 0.0;
  if (sn > 0.0) {
    const double k = -w / (n * sn);
    g0 = k * (s[0] - c * u0); g1 = k * (s[1] - c * u1); g2 = k * (s[2] - c * u2);
  }
  const double m0 = q1 * g2 - q2 * g1, m1 = q2 * g0 - q0 * g2, m2 = q0 * g1 - q1 * g0;  // (R X) x g
  Jc[0] = m0 * Jl[0] + m1 * Jl[3] + m2 * Jl[6];
  Jc[1] = m0 * Jl[1] + m1 * Jl[4] + m2 * Jl[7];
  Jc[2] = m0 * Jl[2] + m1 * Jl[5] + m2 * Jl[8];
  Jc[3] = g0; Jc[4] = g1; Jc[5] = g2;
  Jp[0] = g0 * R[0] + g1 * R[3] + g2 * R[6];
  Jp[1] = g0 * R[1] + g1 * R[4] + g2 * R[7];
  Jp[2] = g0 * R[2] + g1 * R[5] + g2 * R[8];
}

// ceres::HuberLoss(a) on s = r^2: rho (block cost = rho/2) and rho' (Ceres' corrector with rho'' <= 0
// scales r and J by sqrt(rho')).  loss: 0 none, 1 Huber.
PVLM_HD inline void loss_eval(int loss, double a, double s, double* rho, double* rho1) {
  if (loss == 1 && s > a * a) {
    const double rr = sqrt(s);
    *rho = 2.0 * a * rr - a * a;
    *rho1 = a / rr;
  } else {
    *rho = s; *rho1 = 1.0;
  }
}

// Symmetric 3x3 stored as [xx xy xz yy yz zz].  Inverse through the Cholesky factor; false if not SPD.
PVLM_HD inline bool spd3_inverse(const double* V, double* inv) {
  const double l00s = V[0];
  if (!(l00s > 0.0)) return false;
  const double l00 = sqrt(l00s);
  const double l10 = V[1] / l00, l20 = V[2] / l00;
  const double l11s = V[3] - l10 * l10;
  if (!(l11s > 0.0)) return false;
  const double l11 = sqrt(l11s);
  const double l21 = (V[4] - l20 * l10) / l11;
  const double l22s = V[5] - l20 * l20 - l21 * l21;
  if (!(l22s > 0.0)) return false;
  const double l22 = sqrt(l22s);
  // M = L^-1 (lower)
  const double m00 = 1.0 / l00, m11 = 1.0 / l11, m22 = 1.0 / l22;
  const double m10 = -l10 * m00 * m11;
  const double m21 = -l21 * m11 * m22;
  const double m20 = -(l20 * m00 + l21 * m10) * m22;
  // inv = M^T M
  inv[0] = m00 * m00 + m10 * m10 + m20 * m20;
  inv[1] = m10 * m11 + m20 * m21;
  inv[2] = m20 * m22;
  inv[3] = m11 * m11 + m21 * m21;
  inv[4] = m21 * m22;
  inv[5] = m22 * m22;
  return true;
}

// Damped point block in the caller's (unscaled) coordinates:
//   Vd = V + diag(lambda_k / scale_k^2),  lambda_k = clamp(V_kk scale_k^2, min_diag, max_diag) / radius
// which is D^-1 (D V D + Lambda) D^-1 for the Jacobi scaling D = diag(scale) of the LM driver.
PVLM_HD inline void damp3(const double* V, const double* scale, double radius, double min_diag, double max_diag, double* Vd) {
  Vd[0] = V[0]; Vd[1] = V[1]; Vd[2] = V[2]; Vd[3] = V[3]; Vd[4] = V[4]; Vd[5] = V[5];
  const int dg[3] = {0, 3, 5};
  for (int k = 0; k < 3; ++k) {
    const double s2 = scale[k] * scale[k];
    double d = V[dg[k]] * s2;
    d = d < min_diag ? min_diag : (d > max_diag ? max_diag : d);
    Vd[dg[k]] += d / (radius * s2);
  }
}

PVLM_HD inline void sym3_mul(const double* A, const double* v, double* o) {
  o[0] = A[0] * v[0] + A[1] * v[1] + A[2] * v[2];
  o[1] = A[1] * v[0] + A[3] * v[1] + A[4] * v[2];
  o[2] = A[2] * v[0] + A[4] * v[1] + A[5] * v[2];
}

}  // namespace pvlm_reproj
