// Micro-benchmark (not part of the product): cost of a grid-wide barrier inside one cooperative launch against the gap between
// dependent kernel launches on one stream.   hipcc --offload-arch=gfx950 -O3 -o gridsync_bench gridsync_bench.hip
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;

__global__ void __launch_bounds__(256) k_coop(double* x, int phases) {
  cg::grid_group g = cg::this_grid();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = x[i];
  for (int p = 0; p < phases; p++) {
    v = v * 1.0000001 + 1.0;
    x[i] = v;
    g.sync();
    v += x[(i + 256) % (gridDim.x * blockDim.x)] * 1e-9;
  }
  x[i] = v;
}

// hand-made barrier: one atomic counter per phase, spin with a bound (no hang if something goes wrong)
__global__ void __launch_bounds__(256) k_flag(double* x, int phases, unsigned* ctr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = x[i];
  for (int p = 0; p < phases; p++) {
    v = v * 1.0000001 + 1.0;
    x[i] = v;
    __threadfence();
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(&ctr[p], 1u);
      int spins = 0;
      while (__hip_atomic_load(&ctr[p], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x && ++spins < 2000000) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    v += x[(i + 256) % (gridDim.x * blockDim.x)] * 1e-9;
  }
  x[i] = v;
}

__global__ void __launch_bounds__(256) k_plain(double* x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  x[i] = x[i] * 1.0000001 + 1.0 + x[(i + 256) % (gridDim.x * blockDim.x)] * 1e-9;
}

int main() {
  const int phases = 34;
  for (int nb : {64, 256, 512, 1024}) {
    double* x; unsigned* ctr;
    hipMalloc(&x, (size_t)nb * 256 * sizeof(double)); hipMemset(x, 0, (size_t)nb * 256 * sizeof(double));
    hipMalloc(&ctr, 64 * sizeof(unsigned));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    int ph = phases;
    void* args[] = {&x, &ph};
    // cooperative
    hipError_t rc = hipLaunchCooperativeKernel((void*)k_coop, dim3(nb), dim3(256), args, 0, 0);
    hipDeviceSynchronize();
    if (rc != hipSuccess) printf("blocks %4d: cooperative launch refused (%s)\n", nb, hipGetErrorString(rc));
    else {
      hipEventRecord(e0, 0);
      for (int r = 0; r < 20; r++) hipLaunchCooperativeKernel((void*)k_coop, dim3(nb), dim3(256), args, 0, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d: cooperative kernel with %d grid.sync: %.1f us per launch, %.2f us per phase\n", nb, phases, ms * 1e3 / 20, ms * 1e3 / 20 / phases);
    }
    if (nb <= 512) {
      hipMemset(ctr, 0, 64 * sizeof(unsigned));
      hipLaunchKernelGGL(k_flag, dim3(nb), dim3(256), 0, 0, x, phases, ctr);
      hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      for (int r = 0; r < 20; r++) { hipMemsetAsync(ctr, 0, 64 * sizeof(unsigned), 0); hipLaunchKernelGGL(k_flag, dim3(nb), dim3(256), 0, 0, x, phases, ctr); }
      hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      printf("blocks %4d: atomic-counter barrier x %d: %.1f us per launch, %.2f us per phase\n", nb, phases, ms * 1e3 / 20, ms * 1e3 / 20 / phases);
    }
    hipEventRecord(e0, 0);
    for (int r = 0; r < 20; r++) for (int p = 0; p < phases; p++) hipLaunchKernelGGL(k_plain, dim3(nb), dim3(256), 0, 0, x);
    hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("blocks %4d: %d dependent launches: %.1f us, %.2f us per launch\n", nb, phases, ms * 1e3 / 20, ms * 1e3 / 20 / phases);
    hipFree(x); hipFree(ctr);
  }
  return 0;
}
