// ORACLE (test infrastructure only -- see oracle/README.md).
// Loss seeds of the reverse sweep and the rewards the trajectory optimisers maximise, restated loop for loop from
//   /root/reference/code/engine/analytic_grad_single.py:259-471   (Grad.get_loss_*: what the sweep starts from)
//   /root/reference/code/engine/analytic_grad_system.py:171-183   (the seeds of the system-identification Grad)
//   /root/reference/code/task_scene/Scene_*.py                    (compute_reward*: the scalar of every task)
// so that the GPU-side tests let BOTH sides seed themselves and compare the seed arrays and the reward before the sweep, instead of
// copying the product's seed into the oracle.  Taichi's parallel `for` over ndrange is restated as the serial loop in the order the
// kernel states it; where several iterations write the same entry (get_loss_balance / get_loss_side: every ball vertex writes the
// cloth-centre entry) the serial order keeps the LAST writer -- under Taichi's parallel loop which writer wins is unspecified, so
// this is one of the admissible outcomes, not a pinned one.
// Row indices of the folding seeds / rewards (6, 8) / (7, 9) are parameters: the reference hard-codes them for N = 15
// (analytic_grad_single.py:289-292, Scene_folding.py:138-143), refined cloths scale them by N / 15.
#include <cmath>
#include <cstring>
#include <string>

#include "tslo_engine.h"

namespace tslo {

// ti.cast(x / (M + 1), ti.i32): x and M + 1 are integers in the kernels, Taichi's `/` is true division, the cast truncates
static inline int row_of(int v, int M) { return (int)((double)v / (double)(M + 1)); }

// analytic_grad_single.py:259-471.  name without the "get_loss" prefix; a0 / a1: curve7 / curve8 (fold) or sys.target (bounce);
// rows: {row of p1, row of p2} pairs for the hinge seeds; target: NV x 3 (push).  Returns tt of get_loss_bounce, else 0; -1 unknown name.
int grad_get_loss(Grad& g, Scene& sys, const char* name_, double a0, double a1, const int* rows, const double* target) {
  const std::string name(name_);
  const int T = g.tot_timestep;
  if (name == "") {  // get_loss :259-263
    for (int i = 0; i < sys.cloths[0].NV; i++) for (int j = 0; j < T; j++) g.PG(j, i, 0) = -1;
    return 0;
  }
  if (name == "sheet") {  // :265-269
    for (int i = 0; i < sys.cloths[0].NV; i++) for (int j = 0; j < T - 1; j++) g.PG(j + 1, i, 0) = 1;
    return 0;
  }
  if (name == "book") {  // :274-278
    for (int i = 0; i < sys.cloths[0].NV; i++) for (int j = 0; j < T - 1; j++) g.PG(j + 1, i, 0) = -1;
    return 0;
  }
  if (name == "fold") {  // :280-294
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NF; i++)
      for (int l = 0; l < 3; l++)
        if (c.counter_face[i][l] > i) {
          const int p1 = c.f2v[i][l];
          const int p2 = c.f2v[c.counter_face[i][l]][c.counter_point[i][l]];
          if (row_of(p1, c.M) == rows[0] && row_of(p2, c.M) == rows[1]) g.AG(T - 1, 0, i, l) = a0;
          if (row_of(p1, c.M) == rows[2] && row_of(p2, c.M) == rows[3]) g.AG(T - 1, 0, i, l) = a1;
        }
    return 0;
  }
  if (name == "push") {  // :296-300
    const Cloth& c = sys.cloths[0];
    for (int j = 0; j < c.NV; j++) for (int k = 0; k < 3; k++) g.PG(T - 1, c.offset + j, k) = 2 * (g.PB(T - 1, c.offset + j, k) - target[j * 3 + k]);
    return 0;
  }
  if (name == "lift") {  // :302-312
    const int j = T - 1;
    const Elastic& e = sys.elastics[0];
    for (int i = 0; i < e.n_verts; i++) {
      g.PG(j, e.offset + i, 0) = (g.PB(j, e.offset + i, 0) - g.PB(0, e.offset + i, 0) + 0.012);
      g.PG(j, e.offset + i, 1) = (g.PB(j, e.offset + i, 1) - g.PB(0, e.offset + i, 1) + 0.012);
      g.PG(j, e.offset + i, 2) = (g.PB(j, e.offset + i, 2) - g.PB(0, e.offset + i, 2));
    }
    return 0;
  }
  if (name == "sep") {  // :314-321
    for (int i = 0; i < sys.cloths[0].NV; i++) for (int j = 0; j < T; j++) g.PG(j, sys.cloths[0].offset + i, 0) = 1;
    for (int i = 0; i < sys.cloths[1].NV; i++) for (int j = 0; j < T; j++) g.PG(j, sys.cloths[1].offset + i, 0) = -1;
    return 0;
  }
  if (name == "pick" || name == "card") {  // :323-327 / :384-388 (the same body)
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++) for (int j = 0; j < T; j++)
      if (row_of(i, c.M) == 8) g.PG(j, c.offset + i, 2) = -1;
    return 0;
  }
  if (name == "bounce") {  // :329-371 (plain Python in the reference)
    const Cloth& c = sys.cloths[0];
    const double tgt = a0;  // sys.target
    int tt = T - 1;
    double max_z = -1.0;
    for (int j = 40; j < T; j++) {
      double now_z = 0;
      for (int i = 0; i < c.M + 1; i++) now_z += g.PB(j, i + c.offset, 2);
      if (now_z > max_z) { max_z = now_z; tt = j; }
    }
    if (tt < T - 1) {
      double z_prev = 0.0, z_next = 0.0;
      for (int i = 0; i < c.M + 1; i++) { z_prev += g.PB(tt - 1, i + c.offset, 2); z_next += g.PB(tt + 1, i + c.offset, 2); }
      if (z_prev > z_next) { for (int i = 0; i < c.M + 1; i++) g.PG(tt - 1, c.offset + i, 2) = 2 * (g.PB(tt - 1, c.offset + i, 2) - tgt); }
      else { for (int i = 0; i < c.M + 1; i++) g.PG(tt + 1, c.offset + i, 2) = 2 * (g.PB(tt + 1, c.offset + i, 2) - tgt); }
    }
    for (int i = 0; i < c.M + 1; i++) g.PG(tt, c.offset + i, 2) = 2 * (g.PB(tt, c.offset + i, 2) - tgt);
    return tt;
  }
  if (name == "pick_fold") {  // :373-382
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NF; i++) for (int j = 0; j < T; j++)
      for (int l = 0; l < 3; l++)
        if (c.counter_face[i][l] > i) {
          const int p1 = c.f2v[i][l];
          const int p2 = c.f2v[c.counter_face[i][l]][c.counter_point[i][l]];
          if (row_of(p1, c.M) == 7 && row_of(p2, c.M) == 9) g.AG(j, 0, i, l) = -1;
        }
    return 0;
  }
  if (name == "slide_simple") {  // :390-393
    for (int i = 0; i < sys.cloths[0].NV; i++) g.PG(T - 1, sys.cloths[0].offset + i, 0) = 1;
    return 0;
  }
  if (name == "deliver") {  // :395-406
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++)
      for (int k = 0; k < 3; k++) g.PG(T - 1, c.offset + i, k) = 2 * (g.PB(T - 1, c.offset + i, k) - g.PB(69, c.offset + i, k) - 0.01);
    return 0;
  }
  if (name == "interact") {  // :408-420
    for (int i = 0; i < sys.cloths[0].NV; i++) g.PG(T - 1, sys.cloths[0].offset + i, 0) = 1;
    for (int i = 0; i < sys.elastics[3].n_verts; i++) g.PG(T - 1, sys.elastics[3].offset + i, 0) = -1 * 256.0 / 144.0;
    return 0;
  }
  if (name == "interact_1") {  // :422-426
    for (int i = 0; i < sys.elastics[3].n_verts; i++) g.PG(T - 1, sys.elastics[3].offset + i, 0) = 1;
    return 0;
  }
  if (name == "balance" || name == "side") {  // :428-443 / :445-460 (tt a quarter of the way along the cloth)
    const Cloth& c = sys.cloths[0];
    const Elastic& e = sys.elastics[0];
    const int tt = (name == "balance" ? (c.N + 1) / 2 : (c.N + 1) / 4) * (c.M + 1) + (c.M + 1) / 2;
    for (int i = 0; i < e.n_verts; i++)
      for (int j = 0; j < T - 1; j++) {
        g.PG(j + 1, e.offset + i, 0) = 2 * (g.PB(j + 1, e.offset + i, 0) - g.PB(j + 1, c.offset + tt, 0));
        g.PG(j + 1, e.offset + i, 1) = 2 * (g.PB(j + 1, e.offset + i, 1) - g.PB(j + 1, c.offset + tt, 1));
        g.PG(j + 1, c.offset + tt, 0) = -2 * (g.PB(j + 1, e.offset + i, 0) - g.PB(j + 1, c.offset + tt, 0));
        g.PG(j + 1, c.offset + tt, 1) = -2 * (g.PB(j + 1, e.offset + i, 1) - g.PB(j + 1, c.offset + tt, 1));
      }
    return 0;
  }
  // analytic_grad_system.py:171-183 (the Grad of the system-identification drivers)
  if (name == "system.slide") {  // :171-173
    for (int i = 0; i < sys.cloths[0].NV; i++) for (int j = 0; j < T - 1; j++) g.PG(j + 1, sys.cloths[0].offset + i, 0) = 1;
    return 0;
  }
  if (name == "system.card") {  // :175-177
    for (int i = 0; i < sys.cloths[0].NV; i++) g.PG(T - 1, sys.cloths[0].offset + i, 0) = 1;
    return 0;
  }
  if (name == "system.table") {  // :179-183 (the row is i / (N + 1) there, not / (M + 1))
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++) for (int j = 0; j < T - 1; j++)
      if ((int)((double)i / (double)(c.N + 1)) == 5 || (int)((double)i / (double)(c.N + 1)) == 10) g.PG(j + 1, c.offset + i, 2) = -1;
    return 0;
  }
  if (name == "throwing") {  // :462-471
    const Cloth& c = sys.cloths[0];
    const Elastic& e = sys.elastics[0];
    for (int i = 0; i < e.n_verts; i++) for (int j = 0; j < T - 1; j++) g.PG(j + 1, e.offset + i, 2) = -1;
    for (int i = 0; i < c.M; i++)
      for (int j = 0; j < T - 1; j++) {
        g.PG(j + 1, c.offset + i, 2) = 20 * g.PB(j + 1, c.offset + i, 2);
        g.PG(j + 1, c.offset + i + c.N * (c.M + 1), 2) = 20 * g.PB(j + 1, c.offset + i + c.N * (c.M + 1), 2);
      }
    return 0;
  }
  return -1;
}

// compute_reward* of the task scenes.  The kernels read the PER-BODY copies (cloths[0].pos, elastics[k].F_x), i.e. the state the last
// time_step left there.  name: "<scene>" or "<scene>.<variant>".  NaN for an unknown name.
double scene_reward(Scene& sys, const Grad* g, const char* name_, double a0, double a1, const int* rows, const double* target) {
  const std::string name(name_);
  double ret = 0.0;
  if (name == "folding" || name == "folding.8" || name == "folding.7") {  // Scene_folding.py:129-147 / :149-169 (curve7 -1, curve8 1) / :171-191 (1, -1)
    const double curve7 = name == "folding" ? a0 : (name == "folding.8" ? -1 : 1), curve8 = name == "folding" ? a1 : (name == "folding.8" ? 1 : -1);
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NF; i++)
      for (int l = 0; l < 3; l++)
        if (c.counter_face[i][l] > i) {
          const int p = c.f2v[c.counter_face[i][l]][c.counter_point[i][l]];
          if (row_of(c.f2v[i][l], c.M) == rows[0] && row_of(p, c.M) == rows[1]) ret += -c.ref_angle[i][l] * curve7;
          else if (row_of(c.f2v[i][l], c.M) == rows[2] && row_of(p, c.M) == rows[3]) ret += -c.ref_angle[i][l] * curve8;
        }
    return ret;
  }
  if (name == "lifting") {  // Scene_lifting.py:152-159
    const Elastic& e = sys.elastics[0];
    for (int i = 0; i < e.n_verts; i++) {
      ret -= std::pow(e.F_x[i][0] - e.F_ox[i][0] + 0.025 + 0.012, 2);
      ret -= std::pow(e.F_x[i][1] - e.F_ox[i][1] + 0.005 + 0.012, 2);
      ret -= std::pow(e.F_x[i][2] - e.F_ox[i][2] - 0.0003, 2);
    }
    return ret;
  }
  if (name == "balancing") {  // Scene_balancing.py:138-145
    const Cloth& c = sys.cloths[0];
    const Elastic& e = sys.elastics[0];
    const int tt = (c.N + 1) / 2 * (c.M + 1) + (c.M + 1) / 2;
    for (int i = 0; i < e.n_verts; i++) { ret -= std::pow(e.F_x[i][0] - c.pos[tt][0], 2); ret -= std::pow(e.F_x[i][1] - c.pos[tt][1], 2); }
    return ret;
  }
  if (name == "balancing.all") {  // Scene_balancing.py:147-154
    const Cloth& c = sys.cloths[0];
    const Elastic& e = sys.elastics[0];
    const int tt = (c.N + 1) / 2 * (c.M + 1) + (c.M + 1) / 2;
    Grad& G = *const_cast<Grad*>(g);
    for (int i = 0; i < e.n_verts; i++)
      for (int j = 0; j < G.tot_timestep; j++) {
        ret -= std::pow(G.PB(j, e.offset + i, 0) - G.PB(j, c.offset + tt, 0), 2);
        ret -= std::pow(G.PB(j, e.offset + i, 1) - G.PB(j, c.offset + tt, 1), 2);
      }
    return ret;
  }
  if (name == "balancing.throwing" || name == "balancing.throwing_RL") {  // Scene_balancing.py:156-167 / :169-179
    const Cloth& c = sys.cloths[0];
    const Elastic& e = sys.elastics[0];
    Grad* G = const_cast<Grad*>(g);
    for (int i = 0; i < e.n_verts; i++) ret += name == "balancing.throwing" ? G->PB(G->tot_timestep - 1, e.offset + i, 2) : e.F_x[i][2];
    for (int i = 0; i < c.M + 1; i++) {
      ret -= 10 * std::pow(c.pos[i][2] - 0.0, 2);
      ret -= 10 * std::pow(c.pos[i + c.N * (c.M + 1)][2] - 0.0, 2);
    }
    return ret;
  }
  if (name == "bouncing") {  // Scene_bouncing.py:107-114
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++)
      if (row_of(i, c.M) == 5 || row_of(i, c.M) == 10) ret += c.pos[i][2];
    return ret;
  }
  if (name == "card" || name == "sliding") {  // Scene_card.py:157-162 / Scene_sliding.py:115-120
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++) ret -= c.pos[i][0];
    return ret;
  }
  if (name == "forming") {  // Scene_forming.py:125-132
    const Cloth& c = sys.cloths[0];
    for (int j = 0; j < c.NV; j++) for (int k = 0; k < 3; k++) ret -= std::pow(c.pos[j][k] - target[j * 3 + k], 2);
    return ret;
  }
  if (name == "interact") {  // Scene_interact.py:149-156
    for (size_t i = 0; i < sys.cloths[0].pos.size(); i++) ret = ret - sys.cloths[0].pos[i][0];
    for (size_t i = 0; i < sys.elastics[3].F_x.size(); i++) ret = ret + sys.elastics[3].F_x[i][0] * 256.0 / 144.0;
    return ret;
  }
  if (name == "interact.1") {  // Scene_interact.py:158-163
    for (size_t i = 0; i < sys.elastics[3].F_x.size(); i++) ret = ret - sys.elastics[3].F_x[i][0];
    return ret;
  }
  if (name == "pick") {  // Scene_pick.py:119-126
    const Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NV; i++) if (row_of(i, c.M) == 8) ret += c.pos[i][2];
    return ret;
  }
  if (name == "pick.deliver") {  // Scene_pick.py:128-136
    const Cloth& c = sys.cloths[0];
    Grad* G = const_cast<Grad*>(g);
    for (int i = 0; i < c.NV; i++) for (int k = 0; k < 3; k++) ret -= std::pow(c.pos[i][k] - G->PB(69, i + c.offset, k) - 0.01, 2);
    return ret;
  }
  if (name == "pick.pick_fold" || name == "pick.pick_and_fold") {  // Scene_pick.py:138-152 / :154-172
    Cloth& c = sys.cloths[0];
    for (int i = 0; i < c.NF; i++)
      for (int l = 0; l < 3; l++)
        if (c.counter_face[i][l] > i) {
          const int p = c.f2v[c.counter_face[i][l]][c.counter_point[i][l]];
          if (row_of(c.f2v[i][l], c.M) == 7 && row_of(p, c.M) == 9) {
            ret += c.ref_angle[i][l];
            const double theta = c.compute_angle(i, c.counter_face[i][l], l);
            ret += 0.01 * theta;
          }
        }
    if (name == "pick.pick_and_fold")
      for (int i = 0; i < c.NV; i++) if (row_of(i, c.M) == 8) ret += c.pos[i][2];
    return ret;
  }
  return std::nan("");
}

}  // namespace tslo
