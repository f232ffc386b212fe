// TEST INFRASTRUCTURE ONLY (oracle). Never linked into the product library.
//
// Second-order forward-mode AD ("jet": value, gradient, dense Hessian) that
// stands in for CasADi's symbolic `jacobian(jacobian(.))`
// (reference DGSQP/dynamics/dynamics_models.py:128-144).  Derivative
// conventions at kinks follow CasADi: if_else
// differentiates the selected branch, comparisons have zero derivative,
// fmax(0,z) has derivative (z>0), fmod(a,b) has d/da = 1.
#pragma once
#include <cmath>

#ifndef JET_MAXV
#define JET_MAXV 32
#endif

struct Jet {
  static thread_local int nv;  // active number of independent variables
  double v;
  double g[JET_MAXV];
  double h[JET_MAXV * JET_MAXV];  // full symmetric storage, row stride nv

  Jet() : v(0) { zero(); }
  Jet(double c) : v(c) { zero(); }
  void zero() {
    for (int i = 0; i < nv; i++) g[i] = 0;
    for (int i = 0; i < nv * nv; i++) h[i] = 0;
  }
  static Jet var(double val, int idx) {
    Jet r(val);
    r.g[idx] = 1.0;
    return r;
  }
  double& H(int i, int j) { return h[i * nv + j]; }
  double H(int i, int j) const { return h[i * nv + j]; }
};

// y = f(a) with f', f'' given
inline Jet jet_unary(const Jet& a, double f, double f1, double f2) {
  const int n = Jet::nv;
  Jet r;
  r.v = f;
  for (int i = 0; i < n; i++) r.g[i] = f1 * a.g[i];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) r.h[i * n + j] = f1 * a.h[i * n + j] + f2 * a.g[i] * a.g[j];
  return r;
}

inline Jet operator+(const Jet& a, const Jet& b) {
  const int n = Jet::nv;
  Jet r;
  r.v = a.v + b.v;
  for (int i = 0; i < n; i++) r.g[i] = a.g[i] + b.g[i];
  for (int i = 0; i < n * n; i++) r.h[i] = a.h[i] + b.h[i];
  return r;
}
inline Jet operator-(const Jet& a, const Jet& b) {
  const int n = Jet::nv;
  Jet r;
  r.v = a.v - b.v;
  for (int i = 0; i < n; i++) r.g[i] = a.g[i] - b.g[i];
  for (int i = 0; i < n * n; i++) r.h[i] = a.h[i] - b.h[i];
  return r;
}
inline Jet operator-(const Jet& a) {
  const int n = Jet::nv;
  Jet r;
  r.v = -a.v;
  for (int i = 0; i < n; i++) r.g[i] = -a.g[i];
  for (int i = 0; i < n * n; i++) r.h[i] = -a.h[i];
  return r;
}
inline Jet operator*(const Jet& a, const Jet& b) {
  const int n = Jet::nv;
  Jet r;
  r.v = a.v * b.v;
  for (int i = 0; i < n; i++) r.g[i] = a.v * b.g[i] + b.v * a.g[i];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      r.h[i * n + j] = a.v * b.h[i * n + j] + b.v * a.h[i * n + j] + a.g[i] * b.g[j] + a.g[j] * b.g[i];
  return r;
}
inline Jet jet_recip(const Jet& a) {
  double f = 1.0 / a.v;
  return jet_unary(a, f, -f * f, 2.0 * f * f * f);
}
inline Jet operator/(const Jet& a, const Jet& b) { return a * jet_recip(b); }
inline Jet operator+(const Jet& a, double c) { Jet r = a; r.v += c; return r; }
inline Jet operator+(double c, const Jet& a) { return a + c; }
inline Jet operator-(const Jet& a, double c) { return a + (-c); }
inline Jet operator-(double c, const Jet& a) { return (-a) + c; }
inline Jet operator*(const Jet& a, double c) {
  const int n = Jet::nv;
  Jet r;
  r.v = a.v * c;
  for (int i = 0; i < n; i++) r.g[i] = a.g[i] * c;
  for (int i = 0; i < n * n; i++) r.h[i] = a.h[i] * c;
  return r;
}
inline Jet operator*(double c, const Jet& a) { return a * c; }
inline Jet operator/(const Jet& a, double c) { return a * (1.0 / c); }
inline Jet operator/(double c, const Jet& a) { return jet_recip(a) * c; }

inline Jet sin(const Jet& a) { double s = std::sin(a.v), c = std::cos(a.v); return jet_unary(a, s, c, -s); }
inline Jet cos(const Jet& a) { double s = std::sin(a.v), c = std::cos(a.v); return jet_unary(a, c, -s, -c); }
inline Jet tan(const Jet& a) {
  double t = std::tan(a.v), sec2 = 1.0 + t * t;
  return jet_unary(a, t, sec2, 2.0 * t * sec2);
}
inline Jet atan(const Jet& a) {
  double d = 1.0 + a.v * a.v;
  return jet_unary(a, std::atan(a.v), 1.0 / d, -2.0 * a.v / (d * d));
}
inline Jet sqrt(const Jet& a) {
  double s = std::sqrt(a.v);
  return jet_unary(a, s, 0.5 / s, -0.25 / (s * a.v));
}
// |x|^p for x != 0 handled through the caller's ca_abs; here x>0 assumed
inline Jet powc(const Jet& a, double p) {
  double f = std::pow(a.v, p);
  return jet_unary(a, f, p * f / a.v, p * (p - 1.0) * f / (a.v * a.v));
}
// atan2(y, x): d/dy = x/(x^2+y^2), d/dx = -y/(x^2+y^2)
inline Jet atan2(const Jet& y, const Jet& x) {
  const int n = Jet::nv;
  const double r2 = x.v * x.v + y.v * y.v;
  const double fy = x.v / r2, fx = -y.v / r2;
  // second partials of atan2
  const double fyy = -2.0 * x.v * y.v / (r2 * r2);
  const double fxx = 2.0 * x.v * y.v / (r2 * r2);
  const double fxy = (y.v * y.v - x.v * x.v) / (r2 * r2);
  Jet r;
  r.v = std::atan2(y.v, x.v);
  for (int i = 0; i < n; i++) r.g[i] = fy * y.g[i] + fx * x.g[i];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      r.h[i * n + j] = fy * y.h[i * n + j] + fx * x.h[i * n + j] + fyy * y.g[i] * y.g[j] +
                       fxx * x.g[i] * x.g[j] + fxy * (y.g[i] * x.g[j] + x.g[i] * y.g[j]);
  return r;
}

// scalar helpers so that model code can be written once for double and Jet
inline double val(double a) { return a; }
inline double val(const Jet& a) { return a.v; }
inline double powc(double a, double p) { return std::pow(a, p); }
