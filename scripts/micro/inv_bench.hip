// Micro-benchmark (not part of the product): time of the tile inversions inside a kernel, launch overhead excluded.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o inv_bench inv_bench.hip && ./inv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include "../../thinshelllab_amd/csrc/k_direct.hpp"

template <int V>
__global__ void __launch_bounds__(256) k_bench(const double* __restrict__ in, double* __restrict__ out, int reps, int* bad, long long* cyc) {
  __shared__ double T[DS_T][DS_T + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  long long t0 = 0, acc = 0;
  for (int r = 0; r < reps; r++) {
    for (int q = 0; q < 4; q++) T[ty + 8 * q][tx] = in[(ty + 8 * q) * DS_T + tx];
    __syncthreads();
    t0 = wall_clock64();
    if (V == 4) { __shared__ DsInvLds L4; ds_invert_tile_guarded(L4, &T[0][0], DS_T + 1, bad, 1, 0, 1e-8); } else ds_invert_tile(&T[0][0], DS_T + 1, bad, 1, 0, 1e-8);
    acc += wall_clock64() - t0;
  }
  for (int q = 0; q < 4; q++) out[(ty + 8 * q) * DS_T + tx] = T[ty + 8 * q][tx];
  if (threadIdx.x == 0) cyc[blockIdx.x] = acc;
}

static double check_ref(const std::vector<double>& h, const std::vector<double>& o) {   // max |o - inv(h)| / |inv(h)|_max per row-scaled entry, inverse by long double Gauss-Jordan with partial pivoting
  const int n = DS_T; std::vector<long double> a(n * 2 * n);
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) { a[i * 2 * n + j] = h[i * n + j]; a[i * 2 * n + n + j] = i == j; }
  for (int c = 0; c < n; c++) {
    int p = c; for (int i = c + 1; i < n; i++) if (fabsl(a[i * 2 * n + c]) > fabsl(a[p * 2 * n + c])) p = i;
    if (p != c) for (int j = 0; j < 2 * n; j++) std::swap(a[c * 2 * n + j], a[p * 2 * n + j]);
    const long double ip = 1.0L / a[c * 2 * n + c];
    for (int j = 0; j < 2 * n; j++) a[c * 2 * n + j] *= ip;
    for (int i = 0; i < n; i++) if (i != c) { const long double f = a[i * 2 * n + c]; if (f != 0) for (int j = 0; j < 2 * n; j++) a[i * 2 * n + j] -= f * a[c * 2 * n + j]; }
  }
  double worst = 0;   // relative to sqrt(|inv_ii| |inv_jj|): the natural scale of entry (i, j)
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) {
    const long double sc = sqrtl(fabsl(a[i * 2 * n + n + i]) * fabsl(a[j * 2 * n + n + j]));
    worst = fmax(worst, (double)(fabsl(o[i * n + j] - a[i * 2 * n + n + j]) / sc));
  }
  return worst;
}
static double check(const std::vector<double>& h, const std::vector<double>& o) {
  double err = 0;  // A * inv(A) = I
  for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) { double s = 0; for (int k = 0; k < DS_T; k++) s += h[i * DS_T + k] * o[k * DS_T + j]; err = fmax(err, fabs(s - (i == j))); }
  return err;
}

int main() {
  double *din, *dout; int* bad; long long* cyc;
  const size_t nn = DS_T * DS_T;
  hipMalloc(&din, nn * 8); hipMalloc(&dout, nn * 8); hipMalloc(&bad, (8 + 4 * DS_BADLOG) * 4); hipMalloc(&cyc, 8 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  // case 0: diagonally dominant; 1: SPD with condition 1e8; 2: indefinite with zero leading entries (needs the perturbation-free path); 3: one exactly singular 4 x 4 block
  for (int cs = 0; cs < 7; cs++) {
    std::vector<double> h(nn);
    for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) h[i * DS_T + j] = (i == j ? 40.0 : 0.0) + ((i * 37 + j * 11) % 17) * 0.1;
    if (cs == 1) {
      std::vector<double> q(nn);
      for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) q[i * DS_T + j] = sin(0.37 * (i + 1) * (j + 2)) + (i == j);
      for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) { double s = 0; for (int k = 0; k < DS_T; k++) s += q[k * DS_T + i] * pow(10.0, -8.0 * k / 31.0) * q[k * DS_T + j]; h[i * DS_T + j] = s * 1e6; }
    }
    if (cs == 2) for (int i = 0; i < DS_T; i += 2) { h[i * DS_T + i] = 0.0; h[(i + 1) * DS_T + i + 1] = 0.0; h[i * DS_T + i + 1] = 30.0; h[(i + 1) * DS_T + i] = 30.0; }
    if (cs == 3) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) h[i * DS_T + j] = 1.0;
    if (cs >= 4) {   // contact-like scales: a few diagonal entries of 1e13 among entries of 0.3 .. 1e7, couplings of relative size 0.3 (cs 4), 0.9 (cs 5); cs 6: 3 x 3 vertex blocks of rank-1 contact terms k n n^T
      unsigned z = 12345u + cs; auto rnd = [&]() { z = z * 1664525u + 1013904223u; return (z >> 8) * (1.0 / 16777216.0) - 0.5; };
      std::vector<double> dsc(DS_T);
      for (int i = 0; i < DS_T; i++) dsc[i] = (i % 7 == 3) ? 1e13 : ((i % 5 == 1) ? 1e7 : 0.3 + (i % 3));
      const double cpl = cs == 4 ? 0.3 : 0.9;
      for (int i = 0; i < DS_T; i++) for (int j = 0; j <= i; j++) { const double v = i == j ? dsc[i] : cpl * rnd() * sqrt(dsc[i] * dsc[j]) / 4.0; h[i * DS_T + j] = v; h[j * DS_T + i] = v; }
      if (cs == 6) {
        for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) h[i * DS_T + j] = (i == j ? 0.3 : 0.0) + ((i / 3 == j / 3 || abs(i / 3 - j / 3) == 1) ? 1e5 * rnd() + (i == j ? 3e5 : 0) : 0.0);
        for (int v = 0; v < 10; v += 3) { double n[3] = {0.6, 0.0, 0.8}; for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) h[(3 * v + a) * DS_T + 3 * v + b] += 1e13 * n[a] * n[b]; }
      }
    }
    hipMemcpy(din, h.data(), nn * 8, hipMemcpyHostToDevice);
    for (int V : {4, 6}) {
      hipMemset(bad, 0, 32);
      for (int blocks : {1, 256}) {
        for (int reps : {1, 101}) {
          auto L = [&]() { if (V == 1) hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else if (V == 2) hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else if (V == 4) hipLaunchKernelGGL(k_bench<4>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else if (V == 6) hipLaunchKernelGGL(k_bench<6>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else hipLaunchKernelGGL(k_bench<5>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); };
          L(); hipDeviceSynchronize();
          hipEventRecord(e0); L(); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
          if (cs == 0) printf("form %d blocks %4d reps %3d: kernel %.2f us (%.2f us per inversion), wall_clock64 per inversion %.0f ticks (100 MHz)\n", V, blocks, reps, ms * 1e3, ms * 1e3 / reps, (double)c / reps);
        }
      }
      std::vector<double> o(nn); int hb[8];
      hipMemcpy(o.data(), dout, nn * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
      printf("case %d form %d: max |A inv(A) - I| = %.2e, scaled error against a long-double inverse %.2e, perturbed pivots counted %d, tiles redone %d\n", cs, V, check(h, o), check_ref(h, o), hb[1], hb[6]);
    }
  }
  return 0;
}
