"""Embarrassingly-parallel scene batches: one process per GPU, one engine context per process.

The hot path does not shard inside a scene (a Newton solve couples every dof): independent
rollouts (trajectory-optimisation batch members) are placed one per rank and only their tiny results
(reward: 1 f64, gripper_grad: T x n_part x 6 f64) are exchanged once per optimisation iteration.  Backend "nccl" is
RCCL on ROCm; "gloo" is used by the CPU tests.  No data-path collective exists.
"""
import os

import torch


class Batch:
    def __init__(self, backend=None, device=None):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.dist = None
        self.device = device
        # under torch.distributed.run the process group is formed even for one rank, so that `--nproc-per-node 1` exercises the same
        # RCCL path (init, barrier, all_reduce / all_gather) as the 2/4/8-GPU launches
        if self.world > 1 or ("RANK" in os.environ and "MASTER_ADDR" in os.environ):
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            kw = {}
            if backend == "nccl":
                torch.cuda.set_device(self.local_rank)
                kw["device_id"] = torch.device("cuda", self.local_rank)
            if not dist.is_initialized():
                dist.init_process_group(backend, rank=self.rank, world_size=self.world, **kw)
            self.dist = dist
            self.backend = backend
        if self.device is None:
            self.device = torch.device("cuda", self.local_rank) if torch.cuda.is_available() else torch.device("cpu")

    def scene_ids(self, n_scenes):
        """scene s runs on rank s % world"""
        return [s for s in range(n_scenes) if s % self.world == self.rank]

    def barrier(self):
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def max_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value):
        if self.dist is None:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def share_population(self, n, evaluate):
        """fitness of n independent candidates (CMA-ES population, run_cmaes_all): candidate k runs on rank k % world, the values
        are combined with one all_reduce of an n-vector; every rank returns the full list"""
        f = torch.zeros(n, dtype=torch.float64, device=self.device)
        for k in self.scene_ids(n):
            f[k] = float(evaluate(k))
        if self.dist is not None:
            self.dist.all_reduce(f, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in f.cpu()]

    def gather_results(self, reward, gripper_grad):
        """all_gather of (reward, gripper_grad) -> list over ranks; the only exchange of a batched trajopt iteration"""
        g = gripper_grad.to(self.device, torch.float64).contiguous()
        r = torch.tensor([float(reward)], dtype=torch.float64, device=self.device)
        if self.dist is None:
            return [float(reward)], [g.cpu()]
        rs = [torch.zeros_like(r) for _ in range(self.world)]
        gs = [torch.zeros_like(g) for _ in range(self.world)]
        self.dist.all_gather(rs, r)
        self.dist.all_gather(gs, g)
        return [float(x.item()) for x in rs], [x.cpu() for x in gs]

    def close(self):
        if self.dist is not None and self.dist.is_initialized():
            self.dist.destroy_process_group()
