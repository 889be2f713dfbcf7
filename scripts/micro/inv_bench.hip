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
    if (V == 1) ds_invert_tile_wg(&T[0][0], DS_T + 1, bad, 1, 0, 1e-8); else if (V == 2) ds_invert_tile_wg2(&T[0][0], DS_T + 1, bad, 1, 0, 1e-8); else ds_invert_tile_wg4(&T[0][0], DS_T + 1, bad, 1, 0, 1e-8);
    acc += wall_clock64() - t0;
  }
  for (int q = 0; q < 4; q++) out[(ty + 8 * q) * DS_T + tx] = T[ty + 8 * q][tx];
  if (threadIdx.x == 0) cyc[blockIdx.x] = acc;
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
  for (int cs = 0; cs < 4; cs++) {
    std::vector<double> h(nn);
    for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) h[i * DS_T + j] = (i == j ? 40.0 : 0.0) + ((i * 37 + j * 11) % 17) * 0.1;
    if (cs == 1) {
      std::vector<double> q(nn);
      for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) q[i * DS_T + j] = sin(0.37 * (i + 1) * (j + 2)) + (i == j);
      for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) { double s = 0; for (int k = 0; k < DS_T; k++) s += q[k * DS_T + i] * pow(10.0, -8.0 * k / 31.0) * q[k * DS_T + j]; h[i * DS_T + j] = s * 1e6; }
    }
    if (cs == 2) for (int i = 0; i < DS_T; i += 2) { h[i * DS_T + i] = 0.0; h[(i + 1) * DS_T + i + 1] = 0.0; h[i * DS_T + i + 1] = 30.0; h[(i + 1) * DS_T + i] = 30.0; }
    if (cs == 3) for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) h[i * DS_T + j] = 1.0;
    hipMemcpy(din, h.data(), nn * 8, hipMemcpyHostToDevice);
    for (int V = 1; V <= 3; V++) {
      hipMemset(bad, 0, 32);
      for (int blocks : {1, 256}) {
        for (int reps : {1, 101}) {
          auto L = [&]() { if (V == 1) hipLaunchKernelGGL(k_bench<1>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else if (V == 2) hipLaunchKernelGGL(k_bench<2>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); else hipLaunchKernelGGL(k_bench<3>, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc); };
          L(); hipDeviceSynchronize();
          hipEventRecord(e0); L(); hipEventRecord(e1); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
          if (cs == 0) printf("form %d blocks %4d reps %3d: kernel %.2f us (%.2f us per inversion), wall_clock64 per inversion %.0f ticks (100 MHz)\n", V == 3 ? 4 : V, blocks, reps, ms * 1e3, ms * 1e3 / reps, (double)c / reps);
        }
      }
      std::vector<double> o(nn); int hb[8];
      hipMemcpy(o.data(), dout, nn * 8, hipMemcpyDeviceToHost); hipMemcpy(hb, bad, 32, hipMemcpyDeviceToHost);
      printf("case %d form %d: max |A inv(A) - I| = %.2e, perturbed pivots counted %d\n", cs, V == 3 ? 4 : V, check(h, o), hb[1]);
    }
  }
  return 0;
}
