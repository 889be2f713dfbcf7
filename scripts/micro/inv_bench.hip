// Micro-benchmark (not part of the product): time of ds_invert_tile_wg inside a kernel, launch overhead excluded.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../../thinshelllab_amd/csrc/k_direct.hpp"

__global__ void __launch_bounds__(256) k_bench(const double* __restrict__ in, double* __restrict__ out, int reps, int* bad, long long* cyc) {
  __shared__ double T[DS_T][DS_T + 1];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  long long t0 = 0, acc = 0;
  for (int r = 0; r < reps; r++) {
    for (int q = 0; q < 4; q++) T[ty + 8 * q][tx] = in[(ty + 8 * q) * DS_T + tx];
    __syncthreads();
    t0 = clock64();
    ds_invert_tile_wg(T, bad, 1, 0, 1e-8);
    acc += clock64() - t0;
  }
  for (int q = 0; q < 4; q++) out[(ty + 8 * q) * DS_T + tx] = T[ty + 8 * q][tx];
  if (threadIdx.x == 0) cyc[blockIdx.x] = acc;
}

int main() {
  std::vector<double> h(DS_T * DS_T);
  for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) h[i * DS_T + j] = (i == j ? 40.0 : 0.0) + ((i * 37 + j * 11) % 17) * 0.1;
  double *din, *dout; int* bad; long long* cyc;
  hipMalloc(&din, h.size() * 8); hipMalloc(&dout, h.size() * 8); hipMalloc(&bad, 16); hipMalloc(&cyc, 8 * 1024);
  hipMemcpy(din, h.data(), h.size() * 8, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int blocks : {1, 256, 1024}) {
    for (int reps : {1, 101}) {
      hipLaunchKernelGGL(k_bench, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_bench, dim3(blocks), dim3(256), 0, 0, din, dout, reps, bad, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
      printf("blocks %4d reps %3d: kernel %.2f us, clock64 per inversion %.0f ticks\n", blocks, reps, ms * 1e3, (double)c / reps);
    }
  }
  std::vector<double> o(h.size());
  hipMemcpy(o.data(), dout, o.size() * 8, hipMemcpyDeviceToHost);
  double err = 0;  // A * inv(A) = I
  for (int i = 0; i < DS_T; i++) for (int j = 0; j < DS_T; j++) { double s = 0; for (int k = 0; k < DS_T; k++) s += h[i * DS_T + k] * o[k * DS_T + j]; err = fmax(err, fabs(s - (i == j))); }
  printf("max |A inv(A) - I| = %.2e\n", err);
  return 0;
}
