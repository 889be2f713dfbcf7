// ORACLE (test infrastructure only -- never imported by the product path).
// Small fp64 vector / matrix helpers for the CPU restatement of the
// ThinShellLab engine.  Mirrors the Taichi vector semantics used in
// /root/reference/code/engine/*.py (ti.Vector / ti.Matrix, f64).
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

namespace tslo {

struct V3 {
  double v[3];
  V3() : v{0, 0, 0} {}
  V3(double a, double b, double c) : v{a, b, c} {}
  double& operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
};
inline V3 operator+(const V3& a, const V3& b) { return V3(a[0] + b[0], a[1] + b[1], a[2] + b[2]); }
inline V3 operator-(const V3& a, const V3& b) { return V3(a[0] - b[0], a[1] - b[1], a[2] - b[2]); }
inline V3 operator-(const V3& a) { return V3(-a[0], -a[1], -a[2]); }
inline V3 operator*(const V3& a, double s) { return V3(a[0] * s, a[1] * s, a[2] * s); }
inline V3 operator*(double s, const V3& a) { return V3(a[0] * s, a[1] * s, a[2] * s); }
inline V3 operator/(const V3& a, double s) { return V3(a[0] / s, a[1] / s, a[2] / s); }
inline V3& operator+=(V3& a, const V3& b) { a[0] += b[0]; a[1] += b[1]; a[2] += b[2]; return a; }
inline V3& operator-=(V3& a, const V3& b) { a[0] -= b[0]; a[1] -= b[1]; a[2] -= b[2]; return a; }
inline double dot(const V3& a, const V3& b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
inline V3 cross(const V3& a, const V3& b) {
  return V3(a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]);
}
inline double norm(const V3& a) { return std::sqrt(dot(a, a)); }
// ti ".normalized()" : v / v.norm()  (eps = 0)
inline V3 normalized(const V3& a) { return a / norm(a); }

struct M3 {
  double m[3][3];
  M3() { std::memset(m, 0, sizeof(m)); }
  double* operator[](int i) { return m[i]; }
  const double* operator[](int i) const { return m[i]; }
  static M3 identity() { M3 r; r[0][0] = r[1][1] = r[2][2] = 1.0; return r; }
};
inline M3 operator+(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i][j] + b[i][j]; return r; }
inline M3 operator-(const M3& a, const M3& b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i][j] - b[i][j]; return r; }
inline M3 operator*(const M3& a, double s) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i][j] * s; return r; }
inline M3 operator*(double s, const M3& a) { return a * s; }
inline M3 operator/(const M3& a, double s) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i][j] / s; return r; }
inline M3 operator*(const M3& a, const M3& b) {
  M3 r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) { double s = 0; for (int k = 0; k < 3; k++) s += a[i][k] * b[k][j]; r[i][j] = s; }
  return r;
}
inline V3 operator*(const M3& a, const V3& x) {
  return V3(a[0][0] * x[0] + a[0][1] * x[1] + a[0][2] * x[2], a[1][0] * x[0] + a[1][1] * x[1] + a[1][2] * x[2],
            a[2][0] * x[0] + a[2][1] * x[1] + a[2][2] * x[2]);
}
inline M3 transpose(const M3& a) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[j][i]; return r; }
inline M3 outer(const V3& a, const V3& b) { M3 r; for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r[i][j] = a[i] * b[j]; return r; }
inline double det(const M3& a) {
  return a[0][0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
         a[0][2] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]);
}
inline double trace(const M3& a) { return a[0][0] + a[1][1] + a[2][2]; }
// adjugate / det, the closed form Taichi emits for 3x3 Matrix.inverse()
inline M3 inverse(const M3& a) {
  double d = det(a);
  M3 r;
  r[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / d;
  r[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / d;
  r[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / d;
  r[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / d;
  r[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / d;
  r[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / d;
  r[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / d;
  r[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / d;
  r[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / d;
  return r;
}
// ti.Matrix.cols([c0, c1, c2])
inline M3 from_cols(const V3& c0, const V3& c1, const V3& c2) {
  M3 r;
  for (int i = 0; i < 3; i++) { r[i][0] = c0[i]; r[i][1] = c1[i]; r[i][2] = c2[i]; }
  return r;
}

struct I3 { int v[3]; int& operator[](int i) { return v[i]; } int operator[](int i) const { return v[i]; } };
struct I4 { int v[4]; int& operator[](int i) { return v[i]; } int operator[](int i) const { return v[i]; } };
struct D3 { double v[3]; double& operator[](int i) { return v[i]; } double operator[](int i) const { return v[i]; } };

}  // namespace tslo
