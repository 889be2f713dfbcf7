// Micro-benchmark (not part of the product): the K loop of a GEMM tile -- LDS operand reads + v_mfma_f64_16x16x4_f64, nothing else -- for
// two register tilings per wave: 2 x 2 MFMA tiles (4 accumulators, 4 operand reads per 4 products; k_ds_gemm today, 4 workgroups per CU)
// and 4 x 4 (16 accumulators, 8 operand reads per 16 products, 2 workgroups per CU).  `hoist` = 1: operand addresses constant over the loop (the
// compiler moves the reads out: matrix cores alone, distinct operand registers).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_lds mfma_lds.hip && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int R, int C>   // R x C MFMA tiles per wave
__global__ void __launch_bounds__(256) k_loop(double* out, int iters) {
  __shared__ double As[128 * 33], Bs[32 * 129];
  for (int i = threadIdx.x; i < 128 * 33; i += 256) As[i] = 1.0 + i * 1e-9;
  for (int i = threadIdx.x; i < 32 * 129; i += 256) Bs[i] = 1.0 - i * 1e-9;
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, wi = w >> 1, wj = w & 1, lr = lane & 15, lk = lane >> 4;
  d4 acc[R][C];
  for (int a = 0; a < R; a++) for (int b = 0; b < C; b++) acc[a][b] = d4{0.0, 0.0, 0.0, 0.0};
  for (int it = 0; it < iters; it++) {
    const int sh = (it * 33) & 1023;   // the operand addresses change with the iteration: the reads stay inside the loop
#pragma unroll
    for (int kk = 0; kk < 8; kk++) {
      double av[R], bv[C];
#pragma unroll
      for (int a = 0; a < R; a++) av[a] = As[((16 * R * wi + 16 * a + lr) * 33 + 4 * kk + lk + sh) & 4095];
#pragma unroll
      for (int b = 0; b < C; b++) bv[b] = Bs[((4 * kk + lk) * 129 + 16 * C * wj + 16 * b + lr + sh) & 4095];
#pragma unroll
      for (int a = 0; a < R; a++)
#pragma unroll
        for (int b = 0; b < C; b++) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a], bv[b], acc[a][b], 0, 0, 0);
    }
  }
  double s = 0;
  for (int a = 0; a < R; a++) for (int b = 0; b < C; b++) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
  if (s == 12345.678) out[0] = s;
}
template <int R, int C>
static void run(const char* name, int wgs_per_cu, double* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000, grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((k_loop<R, C>), dim3(grid), dim3(256), 0, 0, out, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_loop<R, C>), dim3(grid), dim3(256), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double flops = (double)grid * 4 * iters * 8 * R * C * 2048.0;
  printf("%s, %d workgroups per CU: %.2f ms, %.1f TFLOP/s\n", name, wgs_per_cu, ms, flops / ms * 1e-9);
}
int main() {
  double* out; hipMalloc(&out, 8);
  for (int rep = 0; rep < 2; rep++) {
    run<2, 2>("2 x 2 tiles per wave", 4, out);
    run<2, 2>("2 x 2 tiles per wave", 2, out);
    run<4, 2>("4 x 2 tiles per wave", 2, out);
    run<4, 2>("4 x 2 tiles per wave", 3, out);
    run<4, 4>("4 x 4 tiles per wave", 2, out);
    run<4, 4>("4 x 4 tiles per wave", 1, out);
  }
  return 0;
}
