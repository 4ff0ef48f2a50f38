// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked or imported by the product path.
// Minimal forward-mode dual number restating the semantics of ceres::Jet<double,N>
// (Ceres 2.0.0 `include/ceres/jet.h`, un-vendored third-party dependency of the reference,
//  pinned by /root/reference/README.md:22-26). Comparison operators act on the scalar part
// only; abs(f) = (f.a < 0 ? -f : f); derivative formulas are the textbook ones.
// Parity status: "parity unpinned" by the reference (no tests/golden vectors upstream);
// cross-checked in this repo against torch.float64 autograd (tests/test_oracle_crosscheck.py).
// JetT<S, N> is the same dual number over the scalar type S: Jet<N> = JetT<double, N> is what Ceres evaluates;
// JetT<long double, N> (x87 extended precision, 64-bit significand) evaluates the SAME statements with 11 more bits and
// is the arbiter of the 1e-6 parity tests where the double evaluation of the reference's own formula (acos near 1) is
// the less accurate side (tests/test_eval_gpu.py).
#pragma once
#include <cmath>

namespace oracle {


template <typename S, int N>
struct JetT {
  S a;
  S v[N];
  JetT() : a(0) { for (int i = 0; i < N; ++i) v[i] = 0; }
  JetT(double s) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; }  // NOLINT implicit like ceres
  JetT(S s, int k) : a(s) { for (int i = 0; i < N; ++i) v[i] = 0; v[k] = 1; }
  JetT& operator+=(const JetT& o) { a += o.a; for (int i = 0; i < N; ++i) v[i] += o.v[i]; return *this; }
  JetT& operator-=(const JetT& o) { a -= o.a; for (int i = 0; i < N; ++i) v[i] -= o.v[i]; return *this; }
  JetT& operator*=(const JetT& o) { *this = *this * o; return *this; }
};
template <int N> using Jet = JetT<double, N>;

#define ORACLE_JT template <typename S, int N> inline
ORACLE_JT JetT<S, N> operator-(const JetT<S, N>& f) { JetT<S, N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r; }
ORACLE_JT JetT<S, N> operator+(const JetT<S, N>& f, const JetT<S, N>& g) { JetT<S, N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r; }
ORACLE_JT JetT<S, N> operator-(const JetT<S, N>& f, const JetT<S, N>& g) { JetT<S, N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r; }
ORACLE_JT JetT<S, N> operator*(const JetT<S, N>& f, const JetT<S, N>& g) { JetT<S, N> r; r.a = f.a * g.a; for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a; return r; }
ORACLE_JT JetT<S, N> operator/(const JetT<S, N>& f, const JetT<S, N>& g) {
  // ceres: g_a_inverse = 1/g.a; f_a_by_g_a = f.a * g_a_inverse; v = (f.v - f_a_by_g_a * g.v) * g_a_inverse
  JetT<S, N> r; const S gi = S(1.0) / g.a; const S q = f.a * gi; r.a = q;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - q * g.v[i]) * gi;
  return r;
}
ORACLE_JT JetT<S, N> operator+(const JetT<S, N>& f, double s) { JetT<S, N> r = f; r.a += s; return r; }
ORACLE_JT JetT<S, N> operator+(double s, const JetT<S, N>& f) { JetT<S, N> r = f; r.a += s; return r; }
ORACLE_JT JetT<S, N> operator-(const JetT<S, N>& f, double s) { JetT<S, N> r = f; r.a -= s; return r; }
ORACLE_JT JetT<S, N> operator-(double s, const JetT<S, N>& f) { JetT<S, N> r = -f; r.a += s; return r; }
ORACLE_JT JetT<S, N> operator*(const JetT<S, N>& f, double s) { JetT<S, N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r; }
ORACLE_JT JetT<S, N> operator*(double s, const JetT<S, N>& f) { return f * s; }
ORACLE_JT JetT<S, N> operator/(const JetT<S, N>& f, double s) { const S si = S(1.0) / S(s); JetT<S, N> r; r.a = f.a * si; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * si; return r; }
ORACLE_JT JetT<S, N> operator/(double s, const JetT<S, N>& g) { JetT<S, N> r; const S m = -S(s) / (g.a * g.a); r.a = S(s) / g.a; for (int i = 0; i < N; ++i) r.v[i] = m * g.v[i]; return r; }

#define ORACLE_JET_CMP(op)                                                                  \
  ORACLE_JT bool operator op(const JetT<S, N>& f, const JetT<S, N>& g) { return f.a op g.a; } \
  ORACLE_JT bool operator op(const JetT<S, N>& f, double s) { return f.a op s; }         \
  ORACLE_JT bool operator op(double s, const JetT<S, N>& f) { return s op f.a; }
ORACLE_JET_CMP(<) ORACLE_JET_CMP(<=) ORACLE_JET_CMP(>) ORACLE_JET_CMP(>=) ORACLE_JET_CMP(==) ORACLE_JET_CMP(!=)
#undef ORACLE_JET_CMP

ORACLE_JT JetT<S, N> abs(const JetT<S, N>& f) { return f.a < S(0.0) ? -f : f; }
ORACLE_JT JetT<S, N> sqrt(const JetT<S, N>& f) { JetT<S, N> r; r.a = std::sqrt(f.a); const S t = S(1.0) / (S(2.0) * r.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
ORACLE_JT JetT<S, N> cos(const JetT<S, N>& f) { JetT<S, N> r; r.a = std::cos(f.a); const S t = -std::sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
ORACLE_JT JetT<S, N> sin(const JetT<S, N>& f) { JetT<S, N> r; r.a = std::sin(f.a); const S t = std::cos(f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
ORACLE_JT JetT<S, N> acos(const JetT<S, N>& f) { JetT<S, N> r; r.a = std::acos(f.a); const S t = S(-1.0) / std::sqrt(S(1.0) - f.a * f.a); for (int i = 0; i < N; ++i) r.v[i] = t * f.v[i]; return r; }
ORACLE_JT JetT<S, N> atan2(const JetT<S, N>& g, const JetT<S, N>& f) {
  // d/dx atan2(g,f) = (f dg - g df)/(f^2+g^2)
  JetT<S, N> r; r.a = std::atan2(g.a, f.a); const S t = S(1.0) / (f.a * f.a + g.a * g.a);
  for (int i = 0; i < N; ++i) r.v[i] = t * (f.a * g.v[i] - g.a * f.v[i]);
  return r;
}
#undef ORACLE_JT

// scalar overloads so templated code can call oracle::sqrt etc. via ADL-free qualified names
inline double abs(double x) { return std::fabs(x); }
inline double sqrt(double x) { return std::sqrt(x); }
inline double cos(double x) { return std::cos(x); }
inline double sin(double x) { return std::sin(x); }
inline double acos(double x) { return std::acos(x); }
inline double atan2(double y, double x) { return std::atan2(y, x); }
inline long double abs(long double x) { return std::fabs(x); }
inline long double sqrt(long double x) { return std::sqrt(x); }
inline long double cos(long double x) { return std::cos(x); }
inline long double sin(long double x) { return std::sin(x); }
inline long double acos(long double x) { return std::acos(x); }
inline long double atan2(long double y, long double x) { return std::atan2(y, x); }

}  // namespace oracle
