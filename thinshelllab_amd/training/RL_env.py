"""Episode wrapper of a task scene for policy-gradient training: counterpart of /root/reference/code/training/RL_env.py:31-236
(`Env(sys_name, time_step, reward_name, load_dir, task_name, Kb, mu, model)` with reset / step / get_observations /
compute_rewards / check_termination, action = 6 numbers per gripper part in [-1e-3, 1e-3], observation layout of
BaseScene.get_observation_kernel :1586-1619).  `gymnasium` is not installed in this image: when it is importable the class derives
from gym.Env and uses gymnasium.spaces.Box, otherwise a minimal Box with the same fields stands in, so stable-baselines style
trainers work where gymnasium exists and the forward path is testable where it does not.  Forward path only (no adjoint)."""
import importlib
import os

import numpy as np

try:
    import gymnasium as gym
    from gymnasium import spaces
    _Base = gym.Env
    Box = spaces.Box
except ImportError:
    _Base = object

    class Box:
        def __init__(self, low, high, shape, dtype=np.float32):
            self.low = np.full(shape, low, dtype=dtype); self.high = np.full(shape, high, dtype=dtype)
            self.shape = tuple(shape); self.dtype = np.dtype(dtype)

        def sample(self, rng=None):
            rng = rng or np.random.default_rng()
            return rng.uniform(self.low, self.high).astype(self.dtype)

        def contains(self, x):
            x = np.asarray(x)
            return x.shape == self.shape and bool(np.all(x >= self.low) and np.all(x <= self.high))


class Env(_Base):
    count = 0

    def __init__(self, sys_name, time_step, reward_name=None, load_dir=None, task_name=None, Kb=100.0, mu=5.0, model="PPO", target_pos=None,
                 save_root=None):
        super().__init__()
        from ..engine.geometry import projection_query
        self._contact = projection_query
        Scene = importlib.import_module(f"thinshelllab_amd.task_scene.Scene_{sys_name}")
        cloth_size = 0.1 if sys_name in ("folding", "forming") else 0.06
        self.sys_name = sys_name
        Env.count += 1
        self.sys = Scene.Scene(cloth_size=cloth_size, dense=20000) if sys_name == "interact" else Scene.Scene(cloth_size=cloth_size)
        self.target_pos = target_pos
        if sys_name == "forming" and target_pos is None:
            self.target_pos = np.load(os.path.join("..", "data", "forming_pos_save", "cloth_pos.npy"))
        self.sys.init_all()
        self.sys.cloths[0].Kb[None] = Kb
        self.sys.mu_cloth_elastic[None] = mu
        self.n_actions = self.sys.action_dim
        self.n_observations = self.sys.obs_dim
        self.action_space = Box(low=-0.001, high=0.001, shape=(self.n_actions,), dtype=np.float32)
        self.observation_space = Box(low=-1000, high=1000, shape=(self.n_observations,), dtype=np.float32)
        self.time_step = 0
        self.time_limit = time_step
        self.total_rewards = 0
        self.task_name = task_name
        self.reward_name = reward_name
        self.load_dir = load_dir
        self.rewards = []
        self.model = model
        self.save_dir = None
        if model is not None and save_root is not None:   # RL_env.py:99-109 (plot_data.npy of the episode rewards)
            self.save_dir = os.path.join(save_root, f"{task_name}_plot", f"{model}_{Env.count}")
            os.makedirs(self.save_dir, exist_ok=True)
        self.reset()

    # RL_env.py:111-159
    def step(self, action):
        real_rewards = self.compute_real_rewards()
        self.time_step += 1
        if self.time_step <= self.time_limit - 1 and self.task_name == "balance_RL":
            real_rewards -= 0.5
        n_part = self.sys.gripper.n_part
        a = np.asarray(action, dtype=np.float64).reshape(n_part, 6)
        self.sys.action(self.time_step, a[:, 0:3].copy(), a[:, 3:6].copy())
        self.sys.time_step(self._contact, self.time_step)
        obs = self.get_observations()
        rewards = self.compute_rewards()
        dones = self.check_termination()
        truncated = dones
        if truncated:
            obs = np.zeros_like(obs)
            rewards = 0
            self.rewards.append(real_rewards)
            if self.save_dir is not None and len(self.rewards) % 10 == 0:
                np.save(os.path.join(self.save_dir, "plot_data.npy"), np.array(self.rewards))
        else:
            self.total_rewards += rewards
        return obs, rewards, dones, truncated, {}

    # RL_env.py:161-178
    def reset(self, seed=None, options=None):
        self.sys.reset()
        if self.load_dir is not None:
            self.sys.load_all(self.load_dir)
        obs = self.get_observations()
        self.time_step = 0
        self.total_rewards = 0
        return obs, {}

    def get_observations(self):
        return np.asarray(self.sys.get_observation_kernel(), dtype=np.float64).reshape(-1)

    def compute_real_rewards(self):
        if self.reward_name is None:
            if self.sys_name == "forming":
                return self.sys.compute_reward(self.target_pos)
            if self.sys_name == "folding":
                return self.sys.compute_reward(1.0, -1.0)   # the reference calls compute_reward() here, which its folding scene does not accept
            return self.sys.compute_reward()
        func = getattr(self.sys, self.reward_name)
        if not callable(func):
            raise SystemExit(f"{self.reward_name}, not a callable function!!")
        return func()

    def compute_rewards(self):
        return float(np.exp(self.compute_real_rewards()))

    def check_termination(self):
        if self.time_step >= self.time_limit:
            return True
        return bool(self.sys.check_early_stop(self.time_step))

    def close(self):
        pass
