// Micro-benchmark (not part of the product): the hop of a point-to-point dataflow chain inside ONE persistent launch -- what a block
// step of the Gauss-Jordan chain would cost if the workgroups kept their tiles and waited on the pivot workgroup's flag instead of on a
// kernel boundary.  Step k: workgroup pub(k) (which has consumed step k-1, like the owner of pivot tile k+1) "inverts" for `work` ns,
// stores an 8 KB tile, releases, raises the flag; EVERY workgroup polls the flag, acquires, loads the tile (and checks it).
//   hipcc --offload-arch=gfx950 -O3 -o flag_chain flag_chain.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define SPIN_LIMIT 4000000   // a lost flag ends the run with an error count instead of a hang

// V bit 0: back-off between polls; bit 1: 256 copies of the flag on lines of their own (workgroup b polls copy b & 255);
// bit 2: the tile is read with agent-scope (sc1) loads instead of normal loads behind an acquire fence; bit 3: only every 16th workgroup
// reads the tile (the pivot inverse goes to the panel workgroups only); bit 4: the tile is stored with agent-scope (sc1) stores and the
// publisher only waits for them (bit 5: s_waitcnt vmcnt(0); bit 6: a full agent-scope release fence behind the write-through stores)
template <int V>
__global__ void __launch_bounds__(256, 4) k_chain(double* __restrict__ slots, int* __restrict__ flag, int nsteps, int work_ns, int* __restrict__ err,
                                                  int stride) {
  const int b = blockIdx.x, t = threadIdx.x;
  __shared__ int s_bad;
  if (t == 0) s_bad = 0;
  double acc = 0.0;
  int* myflag = (V & 2) ? flag + 32 * (b & 255) : flag;
  for (int k = 0; k < nsteps; k++) {
    const int pub = (int)(((long long)k * stride + 3) % gridDim.x);
    double* slot = slots + (size_t)k * 1024;
    if (b == pub) {
      if (work_ns > 0) {
        const unsigned long long t0 = wall_clock64();   // 100 MHz
        while ((wall_clock64() - t0) * 10ull < (unsigned long long)work_ns) __builtin_amdgcn_s_sleep(1);
      }
      if (V & 16) {                 // write-through (sc1) stores + a wait for their completion instead of a write-back of the L2
        for (int e = t; e < 1024; e += 256) __hip_atomic_store(slot + e, (double)(k + 1) + acc * 1e-30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (V & 64) __threadfence();
        else { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); if (V & 32) __builtin_amdgcn_s_waitcnt(0); }   // (the fence alone emits NO wait: bit 4 without bit 5 / 6 publishes the flag next to the tile)
      } else {
        for (int e = t; e < 1024; e += 256) slot[e] = (double)(k + 1) + acc * 1e-30;
        __threadfence();              // release at agent scope: the tile is in memory before the flag
      }
      __syncthreads();
      if (V & 2) __hip_atomic_store(flag + 32 * t, k + 1, (V & 16) ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      else if (t == 0) __hip_atomic_store(flag, k + 1, (V & 16) ? __ATOMIC_RELAXED : __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0) {
      int spins = 0;
      while (__hip_atomic_load(myflag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < k + 1 && ++spins < SPIN_LIMIT) { if (V & 1) __builtin_amdgcn_s_sleep(8); }
      if (spins >= SPIN_LIMIT) { s_bad = 1; for (int c = 0; c < 256; c++) __hip_atomic_store(flag + 32 * c, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    if ((V & 8) && (b & 15) != 0 && b != (int)(((long long)(k + 1) * stride + 3) % gridDim.x)) continue;
    double s = 0.0;
    if (V & 4) {
      for (int e = t; e < 1024; e += 256) s += __hip_atomic_load(slot + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      __atomic_thread_fence(__ATOMIC_ACQUIRE);   // agent scope by default in HIP: invalidates what this XCD's L2 holds of other XCDs' data
      for (int e = t; e < 1024; e += 256) s += slot[e];
    }
    if (s != 4.0 * (double)(k + 1)) atomicAdd(err, 1);
    acc += s;
  }
  if (acc == -1.0) slots[0] = acc;
  if (t == 0 && s_bad) atomicAdd(err + 1, 1);
}

template <int V>
void run(double* slots, int* flag, int* err, int nsteps, hipEvent_t e0, hipEvent_t e1) {
  for (int nb : {4, 16, 64, 256, 1024}) {
    float best[2] = {1e30f, 1e30f}; int herr[2] = {0, 0};
    for (int w = 0; w < 2; w++)
      for (int rep = 0; rep < 4; rep++) {
        (void)hipMemset(slots, 0, (size_t)nsteps * 1024 * sizeof(double)); (void)hipMemset(flag, 0, 256 * 128); (void)hipMemset(err, 0, 64);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_chain<V>, dim3(nb), dim3(256), 0, 0, slots, flag, nsteps, w ? 3000 : 0, err, 37);
        (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best[w]) best[w] = ms;
        int h[2]; (void)hipMemcpy(h, err, 8, hipMemcpyDeviceToHost); herr[0] += h[0]; herr[1] += h[1];
      }
    printf("variant %2d  workgroups %4d: %6.2f us per hop; with 3 us of work in the publisher %6.2f  (wrong tiles %d, lost flags %d)\n", V, nb,
           best[0] * 1e3 / nsteps, best[1] * 1e3 / nsteps, herr[0], herr[1]);
  }
}

int main() {
  const int nsteps = 256;
  double* slots; int* flag; int* err;
  (void)hipMalloc(&slots, (size_t)nsteps * 1024 * sizeof(double));
  (void)hipMalloc(&flag, 256 * 128); (void)hipMalloc(&err, 64);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  run<0>(slots, flag, err, nsteps, e0, e1);
  run<6>(slots, flag, err, nsteps, e0, e1);
  run<4>(slots, flag, err, nsteps, e0, e1);
  run<20>(slots, flag, err, nsteps, e0, e1);
  run<52>(slots, flag, err, nsteps, e0, e1);
  run<84>(slots, flag, err, nsteps, e0, e1);
  run<14>(slots, flag, err, nsteps, e0, e1);
  return 0;
}
