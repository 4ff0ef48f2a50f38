// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// Minimal forward-mode dual number restating the semantics of ceres::Jet<double,N>
// (Ceres 2.0.0 `include/ceres/jet.h`, un-vendored third-party dependency of the reference,
//  pinned by /root/reference/README.md:22-26). Comparison operators act on the scalar part
// only; abs(f) = (f.a < 0 ? -f : f); derivative formulas are the textbook ones.
// Parity status: "parity unpinned" by the reference (no tests/golden vectors upstream);
// cross-checked in this repo against torch.float64 autograd (tests/test_oracle_crosscheck.py).
#pragma once
#include <cmath>

namespace oracle {

template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT implicit like ceres
  Jet(double s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0.0; v[k] = 1.0; }
  Jet& operator+=(const Jet& o) { a += o.a; for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
  Jet& operator-=(const Jet& o) { a -= o.a; for (int i = 0; i < N; ++i) v[i] -= o.v[i]; return *this; }
  Jet& operator*=(const Jet& o) { *this = *this * o; return *this; }
};

template <int N> inline Jet<N> operator-(const Jet<N>& f) { Jet<N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) { Jet<N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  // ceres: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
  Jet<N> r; const double gi = 1.0 / g.a; const double q = f.a * gi; r.a = q;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return r;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) { Jet<N> r = f; r.a += s; return r; }
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) { Jet<N> r = f; r.a += s; return r; }
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) { Jet<N> r = f; r.a -= s; return r; }
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) { Jet<N> r = -f; r.a += s; return r; }
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) { Jet<N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) { return f * s; }
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) { const double si = 1.0 / s; return f * si; }
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) { Jet<N> r; const double m = -s / (g.a * g.a); r.a = s / g.a; for (int i = 0; i < N; ++i) r.v[i] = m * g.v[i]; return r; }

#define ORACLE_JET_CMP(op)                                                                  \
  template <int N> inline bool operator op(const Jet<N>& f, const Jet<N>& g) { return f.a op g.a; } \
  template <int N> inline bool operator op(const Jet<N>& f, double s) { return f.a op s; }         \
  template <int N> inline bool operator op(double s, const Jet<N>& f) { return s op f.a; }
ORACLE_JET_CMP(<) ORACLE_JET_CMP(<=) ORACLE_JET_CMP(>) ORACLE_JET_CMP(>=) ORACLE_JET_CMP(==) ORACLE_JET_CMP(!=)
#undef ORACLE_JET_CMP

template <int N> inline Jet<N> abs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
template <int N> inline Jet<N> sqrt(const Jet<N>& f) { Jet<N> r; r.a = std::sqrt(f.a); const double t = 1.0 / (2.0 * r.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
template <int N> inline Jet<N> cos(const Jet<N>& f) { Jet<N> r; r.a = std::cos(f.a); const double t = -std::sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
template <int N> inline Jet<N> sin(const Jet<N>& f) { Jet<N> r; r.a = std::sin(f.a); const double t = std::cos(f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
template <int N> inline Jet<N> acos(const Jet<N>& f) { Jet<N> r; r.a = std::acos(f.a); const double t = -1.0 / std::sqrt(1.0 - f.a * f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
template <int N> inline Jet<N> atan2(const Jet<N>& g, const Jet<N>& f) {
  // d/dx atan2(g,f) = (f dg - g df)/(f^2+g^2)
  Jet<N> r; r.a = std::atan2(g.a, f.a); const double t = 1.0 / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]);
  return r;
}

// scalar overloads so templated code can call oracle::sqrt etc. via ADL-free qualified names
inline double abs(double x) { return std::fabs(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double cos(double x) { return std::cos(x); }
inline double sin(double x) { return std::sin(x); }
inline double acos(double x) { return std::acos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }

}  // namespace oracle
