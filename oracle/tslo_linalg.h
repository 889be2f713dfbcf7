// ORACLE (test infrastructure only).
// Literal restatement of /root/reference/code/engine/linalg.py:
//   SPD_project_2d  (linalg.py:5-12)   -- ti.svd based PSD projection of a symmetric 2x2
//   SPD_Projector   (linalg.py:15-148) -- Householder tridiagonalisation + K shifted-QR sweeps
// plus a converged cyclic-Jacobi eigen-clamp used only to cross-check the literal one.
#pragma once
#include <algorithm>
#include <cmath>

namespace tslo {

// A is row-major n x n stored with leading dimension ld (ld >= n).
#define TSLO_A(i, j) A[(i) * ld + (j)]
#define TSLO_T(i, j) T[(i) * ld + (j)]
#define TSLO_Q(i, j) Q[(i) * ld + (j)]

// linalg.py:21-26
inline void spd_clear(double* T, double* Q, int ld, int n) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { TSLO_T(i, j) = 0; TSLO_Q(i, j) = (i == j) ? 1.0 : 0.0; }
}

// linalg.py:28-75
inline void spd_householder(double* A, double* T, double* Q, int ld, int n) {
  for (int i = 0; i < n - 2; i++) {
    double b = 0.0;
    for (int j = i + 1; j < n; j++) b += TSLO_A(j, i) * TSLO_A(j, i);
    b = std::sqrt(b);
    if (b < 1e-6) {
      TSLO_T(i, i) = -1;
      for (int j = i + 1; j < n; j++) TSLO_A(i, j) = 0;
    } else {
      TSLO_T(i, i) = 1;
      if (TSLO_A(i + 1, i) < 0) b *= -1;
      TSLO_T(i + 1, i) = TSLO_A(i + 1, i) + b;
      double c = TSLO_T(i + 1, i) * TSLO_T(i + 1, i);
      for (int j = i + 2; j < n; j++) {
        TSLO_T(j, i) = TSLO_A(j, i);
        c += TSLO_A(j, i) * TSLO_A(j, i);
      }
      c = std::sqrt(2 / c);
      for (int j = i + 1; j < n; j++) TSLO_T(j, i) *= c;
      for (int j = i + 1; j < n; j++) TSLO_T(i, j) = 0;
      for (int j = i + 1; j < n; j++) {
        for (int k = i + 1; k < j + 1; k++) TSLO_T(i, j) += TSLO_A(j, k) * TSLO_T(k, i);
        for (int k = j + 1; k < n; k++) TSLO_T(i, j) += TSLO_A(k, j) * TSLO_T(k, i);
      }
      double d = 0.0;
      for (int j = i + 1; j < n; j++) d += TSLO_T(i, j) * TSLO_T(j, i);
      d *= 0.5;
      for (int j = i + 1; j < n; j++) {
        TSLO_T(i, j) -= TSLO_T(j, i) * d;
        TSLO_A(i, j) = TSLO_A(j, i) = 0;
      }
      TSLO_A(i + 1, i) = TSLO_A(i, i + 1) = -b;
      for (int j = i + 1; j < n; j++)
        for (int k = i + 1; k < j + 1; k++)
          TSLO_A(j, k) -= TSLO_T(i, j) * TSLO_T(k, i) + TSLO_T(i, k) * TSLO_T(j, i);
      for (int k = 0; k < n; k++) {
        double s = 0.0;
        for (int j = i + 1; j < n; j++) s += TSLO_Q(k, j) * TSLO_T(j, i);
        for (int j = i + 1; j < n; j++) TSLO_Q(k, j) -= s * TSLO_T(j, i);
      }
    }
  }
  TSLO_A(n - 2, n - 1) = TSLO_A(n - 1, n - 2);
}

// linalg.py:77-129 ; returns the number of sweeps actually executed
inline int spd_qr(double* A, double* T, double* Q, int ld, int n, int K) {
  int sweeps = 0;
  for (int j = 0; j < K; j++) {
    int m = 0;
    for (int i = 0; i < n - 1; i++)
      if (std::fabs(TSLO_A(i + 1, i)) > 1e-5) m = i + 2;
    if (m == 0) break;
    sweeps++;
    double a = TSLO_A(m - 2, m - 2);
    double b = TSLO_A(m - 2, m - 1);
    double c = TSLO_A(m - 1, m - 1);
    double d = (a - c) / 2;
    double sd = d > 0 ? 1 : -1;
    double mu = c;
    if (std::fabs(b) > 1e-6) mu -= (sd * b * b) / (std::fabs(d) + std::sqrt(d * d + b * b));
    for (int i = 0; i < n; i++) TSLO_A(i, i) -= mu;
    for (int i = 0; i < m - 1; i++) {
      a = TSLO_A(i, i);
      b = TSLO_A(i, i + 1);
      double e = TSLO_A(i + 1, i);
      d = TSLO_A(i + 1, i + 1);
      double s = std::fabs(e) > 1e-5 ? std::fabs(e / std::sqrt(a * a + e * e)) : 0;
      if (a * e < 0) s *= -1;
      c = std::sqrt(std::max(1 - s * s, 0.0));
      TSLO_T(0, i) = s;
      TSLO_A(i, i) = a * c + e * s;
      TSLO_A(i, i + 1) = b * c + d * s;
      TSLO_A(i + 1, i + 1) = d * c - b * s;
      if (i < n - 2) TSLO_A(i + 1, i + 2) *= c;
    }
    for (int i = 0; i < m - 1; i++) {
      a = TSLO_A(i, i);
      b = TSLO_A(i, i + 1);
      d = TSLO_A(i + 1, i + 1);
      double s = TSLO_T(0, i);
      c = std::sqrt(std::max(1 - s * s, 0.0));
      TSLO_A(i, i) = a * c + b * s;
      TSLO_A(i + 1, i) = s * d;
      TSLO_A(i + 1, i + 1) = c * d;
      for (int r = 0; r < n; r++) {
        double qa = TSLO_Q(r, i), qb = TSLO_Q(r, i + 1);
        TSLO_Q(r, i) = qa * c + qb * s;
        TSLO_Q(r, i + 1) = -qa * s + qb * c;
      }
    }
    for (int i = 0; i < n - 1; i++) TSLO_A(i, i + 1) = TSLO_A(i + 1, i);
    for (int i = 0; i < n; i++) TSLO_A(i, i) += mu;
  }
  return sweeps;
}

inline void spd_project_jacobi(double* A, int ld, int n);
// 0: literal Householder + K QR sweeps (reference); 1: converged Jacobi eigen-clamp (cross-check mode for tests)
inline int& spd_mode() { static int m = 0; return m; }

// linalg.py:132-148.  A, T, Q: n x n with leading dimension ld.
inline int spd_project(double* A, double* T, double* Q, int ld, int n, int K) {
  if (spd_mode() == 1) { spd_project_jacobi(A, ld, n); return 0; }
  spd_clear(T, Q, ld, n);
  spd_householder(A, T, Q, ld, n);
  int sweeps = spd_qr(A, T, Q, ld, n, K);
  for (int i = 0; i < n; i++) TSLO_T(0, i) = TSLO_A(i, i);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) TSLO_A(i, j) = 0;
  for (int i = 0; i < n; i++) {
    double v = TSLO_T(0, i);
    if (v > 0) {
      for (int j = 0; j < n; j++) {
        double v2 = v * TSLO_Q(j, i);
        for (int k = 0; k < n; k++) TSLO_A(j, k) += v2 * TSLO_Q(k, i);
      }
    }
  }
  return sweeps;
}
#undef TSLO_A
#undef TSLO_T
#undef TSLO_Q

// Converged cyclic Jacobi eigen-clamp (NOT in the reference; cross-check only).
inline void spd_project_jacobi(double* A, int ld, int n) {
  double V[9 * 9], S[9 * 9];
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) { S[i * 9 + j] = 0.5 * (A[i * ld + j] + A[j * ld + i]); V[i * 9 + j] = (i == j); }
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; i++) { diag += S[i * 9 + i] * S[i * 9 + i]; for (int j = i + 1; j < n; j++) off += S[i * 9 + j] * S[i * 9 + j]; }
    if (off <= 1e-30 * (diag + off) || off == 0.0) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        double apq = S[p * 9 + q];
        if (apq == 0.0) continue;
        double theta = (S[q * 9 + q] - S[p * 9 + p]) / (2 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
        double c = 1 / std::sqrt(t * t + 1), s = t * c;
        for (int k = 0; k < n; k++) { double skp = S[k * 9 + p], skq = S[k * 9 + q]; S[k * 9 + p] = c * skp - s * skq; S[k * 9 + q] = s * skp + c * skq; }
        for (int k = 0; k < n; k++) { double spk = S[p * 9 + k], sqk = S[q * 9 + k]; S[p * 9 + k] = c * spk - s * sqk; S[q * 9 + k] = s * spk + c * sqk; }
        for (int k = 0; k < n; k++) { double vkp = V[k * 9 + p], vkq = V[k * 9 + q]; V[k * 9 + p] = c * vkp - s * vkq; V[k * 9 + q] = s * vkp + c * vkq; }
      }
  }
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) A[i * ld + j] = 0;
  for (int e = 0; e < n; e++) {
    double lam = S[e * 9 + e];
    if (lam > 0)
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) A[i * ld + j] += lam * V[i * 9 + e] * V[j * 9 + e];
  }
}

// linalg.py:5-12.  ti.svd(A) = U S V^T with S >= 0; a singular triple whose u_i . v_i < 0
// belongs to a negative eigenvalue of the (symmetric) input and is dropped.  For a symmetric
// 2x2 that is exactly sum_{lambda_i > 0} lambda_i q_i q_i^T, evaluated here in closed form.
inline void spd_project_2d(double h[2][2]) {
  double a = h[0][0], b = 0.5 * (h[0][1] + h[1][0]), d = h[1][1];
  double tr = a + d, df = a - d;
  double rad = std::sqrt(df * df * 0.25 + b * b);
  double l1 = tr * 0.5 + rad, l2 = tr * 0.5 - rad;
  // eigenvector of l1
  double q1x, q1y;
  if (rad == 0.0) { q1x = 1; q1y = 0; }
  else if (std::fabs(b) > 0 || df != 0) {
    // (A - l2 I) column gives eigenvector of l1
    double x0 = a - l2, y0 = b;
    double x1 = b, y1 = d - l2;
    if (x0 * x0 + y0 * y0 >= x1 * x1 + y1 * y1) { q1x = x0; q1y = y0; } else { q1x = x1; q1y = y1; }
    double nn = std::sqrt(q1x * q1x + q1y * q1y);
    q1x /= nn; q1y /= nn;
  } else { q1x = 1; q1y = 0; }
  double q2x = -q1y, q2y = q1x;
  double p1 = l1 > 0 ? l1 : 0, p2 = l2 > 0 ? l2 : 0;
  h[0][0] = p1 * q1x * q1x + p2 * q2x * q2x;
  h[0][1] = p1 * q1x * q1y + p2 * q2x * q2y;
  h[1][0] = h[0][1];
  h[1][1] = p1 * q1y * q1y + p2 * q2y * q2y;
}

}  // namespace tslo
