// ORACLE (test infrastructure only).
// System matrix accumulator standing in for /root/reference/code/engine/sparse_solver.py:10-105.
// The reference keeps a dense (n,n) f64 array with timestamped activation; the restatement keeps
// the same add(i,j,v) / clear_all() semantics on a real BSR(3x3) pattern (dense storage is
// 16*n^2 bytes and cannot hold the BASELINE sizes -- SURVEY.md section 0 fact 4).
#pragma once
#include <algorithm>
#include <cstdio>
#include <set>
#include <vector>

namespace tslo {

struct Bsr {
  int nb = 0;                 // block rows (= tot_NV)
  std::vector<int> row_ptr;   // nb+1
  std::vector<int> col;       // block column ids, sorted per row
  std::vector<double> vals;   // nnzb * 9, row-major 3x3 blocks
  long missing = 0;           // add() calls that hit no pattern entry (must stay 0)

  // pattern from vertex cliques (each clique = the vertices one element couples)
  void build(int nb_, const std::vector<std::vector<int>>& cliques) {
    nb = nb_;
    std::vector<std::vector<int>> adj(nb);
    for (int i = 0; i < nb; i++) adj[i].push_back(i);
    for (const auto& c : cliques)
      for (int a : c)
        for (int b : c) adj[a].push_back(b);
    row_ptr.assign(nb + 1, 0);
    col.clear();
    for (int i = 0; i < nb; i++) {
      auto& r = adj[i];
      std::sort(r.begin(), r.end());
      r.erase(std::unique(r.begin(), r.end()), r.end());
      row_ptr[i + 1] = row_ptr[i] + (int)r.size();
      col.insert(col.end(), r.begin(), r.end());
    }
    vals.assign((size_t)col.size() * 9, 0.0);
    missing = 0;
  }
  // sparse_solver.py:21-29
  void clear_all() { std::fill(vals.begin(), vals.end(), 0.0); }
  inline int find_block(int bi, int bj) const {
    const int* b = &col[row_ptr[bi]];
    const int* e = &col[row_ptr[bi + 1]];
    const int* p = std::lower_bound(b, e, bj);
    if (p == e || *p != bj) return -1;
    return (int)(p - &col[0]);
  }
  // sparse_solver.py:31-38  (scalar entry i,j of the n x n matrix)
  inline void add(int i, int j, double v) {
    int k = find_block(i / 3, j / 3);
    if (k < 0) {
#pragma omp atomic
      missing++;
      return;
    }
    double* p = &vals[(size_t)k * 9 + (i % 3) * 3 + (j % 3)];
#pragma omp atomic
    *p += v;
  }
  inline double get(int i, int j) const {
    int k = find_block(i / 3, j / 3);
    return k < 0 ? 0.0 : vals[(size_t)k * 9 + (i % 3) * 3 + (j % 3)];
  }
  // y = A x
  void matvec(const double* x, double* y) const {
#pragma omp parallel for schedule(static)
    for (int bi = 0; bi < nb; bi++) {
      double y0 = 0, y1 = 0, y2 = 0;
      for (int k = row_ptr[bi]; k < row_ptr[bi + 1]; k++) {
        const double* a = &vals[(size_t)k * 9];
        const double* xx = &x[col[k] * 3];
        y0 += a[0] * xx[0] + a[1] * xx[1] + a[2] * xx[2];
        y1 += a[3] * xx[0] + a[4] * xx[1] + a[5] * xx[2];
        y2 += a[6] * xx[0] + a[7] * xx[1] + a[8] * xx[2];
      }
      y[bi * 3 + 0] = y0; y[bi * 3 + 1] = y1; y[bi * 3 + 2] = y2;
    }
  }
  // y = 0.5 (A + A^T) x ; pattern is structurally symmetric
  void matvec_sym(const double* x, double* y, double* tmp) const {
    matvec(x, y);
    // A^T x by scatter (serial; only used by the solver's symmetric-part mode on small problems)
    for (int i = 0; i < nb * 3; i++) tmp[i] = 0;
    for (int bi = 0; bi < nb; bi++)
      for (int k = row_ptr[bi]; k < row_ptr[bi + 1]; k++) {
        const double* a = &vals[(size_t)k * 9];
        const double* xx = &x[bi * 3];
        double* t = &tmp[col[k] * 3];
        t[0] += a[0] * xx[0] + a[3] * xx[1] + a[6] * xx[2];
        t[1] += a[1] * xx[0] + a[4] * xx[1] + a[7] * xx[2];
        t[2] += a[2] * xx[0] + a[5] * xx[1] + a[8] * xx[2];
      }
    for (int i = 0; i < nb * 3; i++) y[i] = 0.5 * (y[i] + tmp[i]);
  }
};

}  // namespace tslo
