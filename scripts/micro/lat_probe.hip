// Micro-benchmark (not part of the product): dependent-chain latencies of the pieces of a tile-inversion block step on gfx950, in shader
// cycles (s_memtime), one wave (or four for the barrier) on an idle chip.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form -o lat_probe lat_probe.hip && ./lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define N 256
#define TICK(t, v) do { asm volatile("" : "+v"(v)); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory"); asm volatile("" : "+v"(v)); } while (0)
__device__ __forceinline__ double rcpn(double x) { double r = __builtin_amdgcn_rcp(x); return fma(fma(-x, r, 1.0), r, r); }

template <int CTRL>
__device__ __forceinline__ double dppd(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__global__ void __launch_bounds__(256) k_probe(double* out, long long* cyc, double seed) {
  __shared__ double L[64 * 33];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long t0, t1;
  double x = seed + lane * 1e-3;
  // 0: dependent v_fma_f64 chain
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < N; i++) x = fma(x, 1.0000001, 1e-9);
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[0] = t1 - t0;
  // 1: independent v_fma_f64 (8 chains)
  double y[8];
  for (int k = 0; k < 8; k++) y[k] = x + k;
  TICK(t0, y[0]);
#pragma unroll
  for (int i = 0; i < N / 8; i++)
#pragma unroll
    for (int k = 0; k < 8; k++) y[k] = fma(y[k], 1.0000001, 1e-9);
  for (int k = 0; k < 8; k++) asm volatile("" : "+v"(y[k]));
  TICK(t1, y[0]);
  for (int k = 0; k < 8; k++) x += y[k];
  if (threadIdx.x == 0) cyc[1] = t1 - t0;
  // 2: dependent rcp + newton
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) x = rcpn(x) + 1.5;
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[2] = t1 - t0;
  // 3: LDS write -> read round trip of one wave (other lane's value), dependent
  if (w == 0) {
    TICK(t0, x);
#pragma unroll
    for (int i = 0; i < 64; i++) {
      L[lane * 33 + (i & 31)] = x;
      x = L[((lane + 1) & 63) * 33 + (i & 31)] + 1e-9;
    }
    TICK(t1, x);
    if (threadIdx.x == 0) cyc[3] = t1 - t0;
  }
  // 4: dependent MFMA chain (accumulator), 5: MFMA -> VALU read -> MFMA operand chain
  d4 acc = {x, x, x, x};
  TICK(t0, acc[0]);
#pragma unroll
  for (int i = 0; i < 64; i++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, acc, 0, 0, 0);
  TICK(t1, acc[0]);
  if (threadIdx.x == 0) cyc[4] = t1 - t0;
  TICK(t0, acc[0]);
#pragma unroll
  for (int i = 0; i < 64; i++) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0] * 1e-3, 1e-3, acc, 0, 0, 0); }
  TICK(t1, acc[0]);
  if (threadIdx.x == 0) cyc[5] = t1 - t0;
  x += acc[0] + acc[1] + acc[2] + acc[3];
  // 6: four independent MFMAs per iteration (issue rate of one wave)
  d4 a4[4] = {acc, acc, acc, acc};
  TICK(t0, a4[0][0]);
#pragma unroll
  for (int i = 0; i < 16; i++)
#pragma unroll
    for (int k = 0; k < 4; k++) a4[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, a4[k], 0, 0, 0);
  for (int k = 0; k < 4; k++) asm volatile("" : "+v"(a4[k][0]));
  TICK(t1, a4[0][0]);
  if (threadIdx.x == 0) cyc[6] = t1 - t0;
  for (int k = 0; k < 4; k++) x += a4[k][0] + a4[k][3];
  // 13: four independent MFMAs followed by 32 independent v_fma_f64 (do the vector instructions run beside the matrix cores?), 14: the 32 alone
  {
    double z[8];
    for (int k = 0; k < 8; k++) z[k] = x + k;
    TICK(t0, z[0]);
#pragma unroll
    for (int k = 0; k < 4; k++) a4[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, a4[k], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) z[k] = fma(z[k], 1.0000001, 1e-9);
    for (int k = 0; k < 8; k++) asm volatile("" : "+v"(z[k]));
    TICK(t1, z[0]);
    if (threadIdx.x == 0) cyc[13] = t1 - t0;
    for (int k = 0; k < 4; k++) asm volatile("" : "+v"(a4[k][0]));
    TICK(t0, z[0]);
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) z[k] = fma(z[k], 1.0000001, 1e-9);
    for (int k = 0; k < 8; k++) asm volatile("" : "+v"(z[k]));
    TICK(t1, z[0]);
    if (threadIdx.x == 0) cyc[14] = t1 - t0;
    for (int k = 0; k < 8; k++) x += z[k];
    for (int k = 0; k < 4; k++) x += a4[k][1];
    // 15: the same with 32-bit vector instructions behind the MFMAs
    int iz[8];
    for (int k = 0; k < 8; k++) iz[k] = lane + k;
    TICK(t0, iz[0]);
#pragma unroll
    for (int k = 0; k < 4; k++) a4[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(1e-3, 1e-3, a4[k], 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int k = 0; k < 8; k++) iz[k] = iz[k] * 3 + 1;
    for (int k = 0; k < 8; k++) asm volatile("" : "+v"(iz[k]));
    TICK(t1, iz[0]);
    if (threadIdx.x == 0) cyc[15] = t1 - t0;
    for (int k = 0; k < 8; k++) x += iz[k];
    for (int k = 0; k < 4; k++) x += a4[k][2];
  }
  // 16: dependent quad sums (2 x (2 DPP movs + add)), 17: four quad broadcasts of one value + 3 adds, 18: v_cmp + ballot + branch round
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 32; i++) { x += dppd<0xB1>(x); x += dppd<0x4E>(x); }
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[16] = t1 - t0;
  x = x * 1e-300 + 1.0;
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 32; i++) { const double a = dppd<0x00>(x), b = dppd<0x55>(x), c = dppd<0xAA>(x), d = dppd<0xFF>(x); x = (a + b) * 0.25 + (c + d) * 0.25; }
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[17] = t1 - t0;
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 32; i++) { if (__builtin_amdgcn_ballot_w64(!(x < 1e300)) != 0) x = sqrt(x); x = fma(x, 1.0000001, 1e-9); }
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[18] = t1 - t0;
  // 7: workgroup barrier, four waves in step
  __syncthreads();
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) __builtin_amdgcn_s_barrier();
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[7] = t1 - t0;
  // 8: LDS write -> barrier -> read (four waves), dependent
  TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 64; i++) {
    L[(threadIdx.x & 63) * 33 + w] = x;
    __syncthreads();
    x = L[((threadIdx.x + 1) & 63) * 33 + ((w + 1) & 3)] + 1e-9;
    __syncthreads();
  }
  TICK(t1, x);
  if (threadIdx.x == 0) cyc[8] = t1 - t0;
  // 9: ds_bpermute chain, 10: readlane chain (2 per double)
  int iv = (int)x;
  TICK(t0, iv);
#pragma unroll
  for (int i = 0; i < 64; i++) iv = __builtin_amdgcn_ds_bpermute(((lane + 1) & 63) << 2, iv) + 1;
  TICK(t1, iv);
  if (threadIdx.x == 0) cyc[9] = t1 - t0;
  TICK(t0, iv);
#pragma unroll
  for (int i = 0; i < 64; i++) iv = __builtin_amdgcn_readlane(iv, (i * 7) & 63) + lane;
  TICK(t1, iv);
  if (threadIdx.x == 0) cyc[10] = t1 - t0;
  // 11: wall clock (100 MHz) against cycles over a fixed spin, to get the shader clock
  long long w0, w1;
  asm volatile("" : "+v"(x)); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0) : : "memory"); TICK(t0, x);
#pragma unroll
  for (int i = 0; i < 8 * N; i++) x = fma(x, 1.0000001, 1e-9);
  TICK(t1, x); asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) : : "memory");
  if (threadIdx.x == 0) { cyc[11] = t1 - t0; cyc[12] = w1 - w0; }
  out[threadIdx.x] = x + iv;
}

int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 256 * 8); hipMalloc(&cyc, 32 * 8);
  for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL(k_probe, dim3(1), dim3(256), 0, 0, out, cyc, 1.25); hipDeviceSynchronize(); }
  long long h[32]; hipMemcpy(h, cyc, 32 * 8, hipMemcpyDeviceToHost);
  printf("readcyclecounter ticks per 100 MHz wall tick: %.2f (s_memtime runs at %.0f MHz)\n", (double)h[11] / h[12], 100.0 * h[11] / h[12]);
  printf("dependent v_fma_f64: %.1f ticks each\n", h[0] / 256.0);
  printf("independent v_fma_f64 (8 chains): %.1f ticks each\n", h[1] / 256.0);
  printf("dependent rcp + newton + add: %.1f ticks per round\n", h[2] / 64.0);
  printf("LDS write -> read (one wave): %.1f ticks per round trip\n", h[3] / 64.0);
  printf("dependent mfma_f64_16x16x4 (accumulator chain): %.1f ticks each\n", h[4] / 64.0);
  printf("mfma -> v_mul -> mfma operand chain: %.1f ticks per round\n", h[5] / 64.0);
  printf("independent mfma x4: %.1f ticks each\n", h[6] / 64.0);
  printf("s_barrier (4 waves): %.1f ticks each\n", h[7] / 64.0);
  printf("LDS write -> barrier -> read -> barrier (4 waves): %.1f ticks per round\n", h[8] / 64.0);
  printf("ds_bpermute chain: %.1f ticks each\n", h[9] / 64.0);
  printf("readlane + add chain: %.1f ticks each\n", h[10] / 64.0);
  printf("4 MFMA then 32 independent v_fma_f64: %lld ticks; the 32 v_fma_f64 alone: %lld; 4 MFMA then 64 v_mad_u32: %lld\n", h[13], h[14], h[15]);
  printf("quad sum (2 stages): %.1f ticks each; 4 quad broadcasts + 2 add + 2 mul + add: %.1f; cmp + ballot + branch + fma: %.1f\n", h[16] / 32.0, h[17] / 32.0, h[18] / 32.0);
  return 0;
}
