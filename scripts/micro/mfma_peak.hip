// Micro-benchmark (not part of the product): sustained rate of v_mfma_f64_16x16x4_f64 from registers alone (no LDS, no memory), to see
// what the matrix cores deliver under load against the 78.6 TFLOP/s of the data sheet.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_peak mfma_peak.hip && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k_mfma(double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; i++) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 12345.678) out[0] = s;
}
int main() {
  double* out; hipMalloc(&out, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs_per_cu : {1, 2, 4}) {
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma<4>, dim3(grid), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 4 /*acc*/ * 2048.0;
    printf("%d workgroups per CU (4 waves each, 4 accumulators per wave): %.2f ms, %.1f TFLOP/s\n", wgs_per_cu, ms, flops / ms * 1e-9);
  }
  return 0;
}
