// Host timing of the factorisation plan (not part of the product): 225 x 225 grid with a two-ring stencil, one 7 x 7 x 6 lattice body,
// 200 contact cliques.   g++ -O2 -std=c++17 -I thinshelllab_amd/csrc -o /tmp/plan_time scripts/micro/plan_time.cpp
#include <chrono>
#include <cstdio>
#include <random>
#include "direct_plan.hpp"
int main() {
  const int N = 224, W = N + 1, ncloth = W * W;
  const int bx = 7, by = 7, bz = 6, nbody = bx * by * bz, NV = ncloth + nbody;
  std::vector<std::vector<int>> adj(NV);
  for (int i = 0; i < W; i++)
    for (int j = 0; j < W; j++)
      for (int di = -2; di <= 2; di++)
        for (int dj = -2; dj <= 2; dj++) {
          if (abs(di) + abs(dj) > 3) continue;
          const int a = i + di, b = j + dj;
          if (a < 0 || b < 0 || a >= W || b >= W) continue;
          adj[i * W + j].push_back(a * W + b);
        }
  for (int x = 0; x < bx; x++) for (int y = 0; y < by; y++) for (int z = 0; z < bz; z++)
    for (int dx = -1; dx <= 1; dx++) for (int dy = -1; dy <= 1; dy++) for (int dz = -1; dz <= 1; dz++) {
      const int a = x + dx, b = y + dy, c = z + dz;
      if (a < 0 || b < 0 || c < 0 || a >= bx || b >= by || c >= bz) continue;
      adj[ncloth + (x * by + y) * bz + z].push_back(ncloth + (a * by + b) * bz + c);
    }
  for (auto& r : adj) std::sort(r.begin(), r.end());
  std::vector<int> rp(NV + 1, 0);
  for (int v = 0; v < NV; v++) rp[v + 1] = rp[v] + (int)adj[v].size();
  std::mt19937 rng(1);
  std::vector<int> cons;
  for (int e = 0; e < 200; e++) {
    const int i = 80 + rng() % 60, j = 80 + rng() % 60;
    cons.push_back(i * W + j); cons.push_back(i * W + j + 1); cons.push_back((i + 1) * W + j); cons.push_back(ncloth + rng() % nbody);
  }
  DirectPlan P;
  std::vector<DsGrid> G{{0, N, N}};
  std::vector<DsBlock> B{{ncloth, nbody}};
  auto t0 = std::chrono::steady_clock::now();
  P.sym.build_partition(NV, adj, G, B, 64);
  auto t1 = std::chrono::steady_clock::now();
  for (int rep = 0; rep < 8; rep++) {
    P.threads = rep < 2 ? 1 : rep < 4 ? 2 : rep < 6 ? 4 : 8;
    auto a = std::chrono::steady_clock::now();
    P.sym.build_tree(adj, cons.data(), 200, 4);
    auto b = std::chrono::steady_clock::now();
    const int rc = P.build(adj, rp, cons.data(), 200);
    auto c = std::chrono::steady_clock::now();
    printf("threads %d phases: tree %.2f desc+rel %.2f levels+worklists %.2f blockmap %.2f | ", P.threads, P.phase_ms[0], P.phase_ms[1], P.phase_ms[2], P.phase_ms[3]);
    printf("build_tree %.2f ms, build (tree + descriptors + maps) %.2f ms, rc %d, supernodes %d levels %d\n", 1e3 * std::chrono::duration<double>(b - a).count(),
           1e3 * std::chrono::duration<double>(c - b).count(), rc, P.sym.n_sn, P.n_levels);
  }
  printf("partition %.2f ms\n", 1e3 * std::chrono::duration<double>(t1 - t0).count());
  return 0;
}
