"""Open-loop trajectory parametrisation: counterpart of ``agent_trajopt``
(/root/reference/code/agent/traj_opt_single.py:5-109).  Plain host arithmetic on a (T, n_part, 6) tensor."""
import math

import torch

from ..engine.field import Field


class agent_trajopt:
    def __init__(self, tot_timestep, cnt, max_moving_dist=0.0005):
        self.traj = Field(torch.zeros((tot_timestep, cnt, 6), dtype=torch.float64))
        self.tmp_action = Field(torch.zeros((cnt, 6), dtype=torch.float64))
        self.delta_pos = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.delta_rot = Field(torch.zeros((cnt, 3), dtype=torch.float64))
        self.action_dim = 6 * cnt
        self.tot_timestep = tot_timestep
        self.max_moving_dist = max_moving_dist
        self.n_part = cnt

    def _delta(self, i, j):
        d = self.traj.t[i, j] - self.traj.t[i - 1, j]
        return d[0:3], d[3:6]

    # traj_opt_single.py:15-27 (the reference's inner loop runs k up to 6*cnt-1 over an axis of size 6: restated as k < 6)
    def fix_action(self, max_dist):
        t = self.traj.t
        for i in range(1, self.tot_timestep):
            for j in range(self.n_part):
                dp, dr = self._delta(i, j)
                moving_dist = math.sqrt(float(dp.dot(dp))) + math.sqrt(float(dr.dot(dr))) * max_dist
                weight = self.max_moving_dist / (moving_dist + 1e-8)
                if weight < 1.0:
                    t[i, j] = t[i - 1, j] + (t[i, j] - t[i - 1, j]) * weight

    # traj_opt_single.py:29-40
    def calculate_dist(self, frame, max_dist, j):
        dp, dr = self._delta(frame, j)
        return math.sqrt(float(dp.dot(dp))) + math.sqrt(float(dr.dot(dr))) * max_dist

    # traj_opt_single.py:42-48
    def get_action(self, step):
        for j in range(self.n_part):
            dp, dr = self._delta(step, j)
            self.delta_pos.t[j] = dp
            self.delta_rot.t[j] = dr

    # hand-scripted initial trajectories (traj_opt_single.py:50-109); rows beyond tot_timestep are out-of-range writes in the
    # reference and are dropped here
    def init_traj_forming(self):
        t = self.traj.t
        for i in range(1, min(20, t.shape[0])):
            t[i, 0, 2] = -0.00011 * i
            t[i, 0, 0] = t[i - 1, 0, 0] + 0.00023
        for i in range(20, min(35, t.shape[0])):
            t[i, 0, 2] = t[i - 1, 0, 2] - 0.0002
            t[i, 0, 0] = t[i - 1, 0, 0] + 0.00027
        for i in range(35, min(50, t.shape[0])):
            t[i, 0, 2] = t[i - 1, 0, 2]
            t[i, 0, 0] = t[i - 1, 0, 0] + 0.0002

    def init_traj_pick_fold(self):
        t = self.traj.t
        for i in range(min(8, t.shape[0])):
            t[i, 0, 2] = -0.0006 * i
            t[i, 1, 2] = -0.0006 * i
            t[i, 0, 0] = t[i - 1, 0, 0]
            t[i, 1, 0] = t[i - 1, 1, 0]
        for i in range(8, min(50, t.shape[0])):
            t[i, 0, 2] = t[i - 1, 0, 2]; t[i, 1, 2] = t[i - 1, 1, 2]
            t[i, 0, 0] = t[i - 1, 0, 0]; t[i, 1, 0] = t[i - 1, 1, 0]

    def init_traj_card(self):  # traj_opt_single.py:75-102 (i - 1 wraps to the last row for i = 0, like the Taichi field;
        t = self.traj.t        # rows beyond tot_timestep are out-of-range writes in the reference and are dropped here)
        for i in range(min(5, t.shape[0])):
            t[i, 0, 0] = t[i - 1, 0, 0] + 0.0003
            t[i, 1, 0] = t[i - 1, 1, 0] - 0.0003
        for lo, hi, dx, dz, dr in ((5, 20, 0.0001, 0.0003, 0.0), (20, 35, 0.0001, 0.0002, 0.0), (35, 50, 0.0002, 0.0005, 0.02), (50, 150, 0.0, 0.0, 0.0)):
            for i in range(lo, min(hi, t.shape[0])):
                t[i, 0, 0] = t[i - 1, 0, 0] + dx
                t[i, 0, 2] = t[i - 1, 0, 2] + dz
                t[i, 0, 4] = t[i - 1, 0, 4] + dr
                t[i, 1, 0] = t[i - 1, 1, 0]

    def init_traj_slide(self):
        t = self.traj.t
        for i in range(min(10, t.shape[0])):
            t[i, 0, 2] = -0.00035 * i
        for i in range(10, min(50, t.shape[0])):
            t[i, 0, 0] = t[i - 1, 0, 0] - 0.0005
            t[i, 0, 2] = t[i - 1, 0, 2]
