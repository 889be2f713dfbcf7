"""(mu/mu_w, lambda)-CMA-ES with the `ask` / `tell` / `result` / `stop` surface that
/root/reference/code/training/run_cmaes_all.py:91,170-186 uses from the third-party package `cma` (pycma; listed without a version
in the reference's pyproject.toml:11-27 and not vendored, not installed here).  The algorithm is restated from its published
description (N. Hansen, "The CMA Evolution Strategy: A Tutorial", 2016: default strategy parameters of Table 1, rank-one +
rank-mu covariance update, cumulative step-size adaptation); sample streams differ from pycma's, so runs are comparable in
distribution only.  Host-side numpy: the optimiser sees 6 * n_gripper * abs_step numbers per candidate, the engine does the rest."""
import math
from collections import namedtuple

import numpy as np

Result = namedtuple("Result", "xbest fbest evals_best evaluations iterations xfavorite stds")


class CMAEvolutionStrategy:
    def __init__(self, x0, sigma0, inopts=None):
        opts = dict(inopts or {})
        self.N = N = len(x0)
        self.mean = np.asarray(x0, dtype=np.float64).copy()
        self.sigma = float(sigma0)
        self.lam = int(opts.get("popsize", 4 + int(3 * math.log(N))))
        self.rng = np.random.default_rng(opts.get("seed", None))
        self.maxiter = opts.get("maxiter", None)
        self.tolfun = float(opts.get("tolfun", 1e-11))
        self.tolx = float(opts.get("tolx", 1e-11))
        mu = self.lam // 2
        w = math.log((self.lam + 1) / 2.0) - np.log(np.arange(1, mu + 1))
        self.weights = w / w.sum()
        self.mu = mu
        self.mueff = 1.0 / np.sum(self.weights ** 2)
        me = self.mueff
        self.cc = (4 + me / N) / (N + 4 + 2 * me / N)
        self.cs = (me + 2) / (N + me + 5)
        self.c1 = 2 / ((N + 1.3) ** 2 + me)
        self.cmu = min(1 - self.c1, 2 * (me - 2 + 1 / me) / ((N + 2) ** 2 + me))
        self.damps = 1 + 2 * max(0.0, math.sqrt((me - 1) / (N + 1)) - 1) + self.cs
        self.chiN = math.sqrt(N) * (1 - 1 / (4 * N) + 1 / (21 * N * N))
        self.pc = np.zeros(N); self.ps = np.zeros(N)
        self.C = np.eye(N); self.B = np.eye(N); self.D = np.ones(N)
        self.eigen_eval = 0
        self.counteval = 0
        self.iterations = 0
        self.xbest = None; self.fbest = math.inf; self.evals_best = 0
        self._last_f = None
        self._y = None

    def _update_eigen(self):
        # O(N^3) decomposition only every ~ N / (10 (c1 + cmu)) evaluations (tutorial, Sec. B.2)
        if self.counteval - self.eigen_eval <= self.lam / (self.c1 + self.cmu) / self.N / 10:
            return
        self.eigen_eval = self.counteval
        self.C = np.triu(self.C) + np.triu(self.C, 1).T
        d, B = np.linalg.eigh(self.C)
        self.D = np.sqrt(np.maximum(d, 1e-30)); self.B = B

    def ask(self):
        self._update_eigen()
        z = self.rng.standard_normal((self.lam, self.N))
        self._y = (z * self.D) @ self.B.T           # y_k ~ N(0, C)
        return [self.mean + self.sigma * y for y in self._y]

    def tell(self, X, fitnesses):
        f = np.asarray(fitnesses, dtype=np.float64)
        X = np.asarray(X, dtype=np.float64)
        if X.shape != (self.lam, self.N) or f.shape != (self.lam,):
            raise ValueError("tell: expected popsize candidates and as many fitness values")
        self.counteval += self.lam
        self.iterations += 1
        order = np.argsort(f, kind="stable")
        if f[order[0]] < self.fbest:
            self.fbest = float(f[order[0]]); self.xbest = X[order[0]].copy(); self.evals_best = self.counteval - self.lam + int(order[0]) + 1
        y = (X[order[:self.mu]] - self.mean) / self.sigma
        yw = self.weights @ y
        self.mean = self.mean + self.sigma * yw
        N = self.N
        invsqrtC_yw = self.B @ ((self.B.T @ yw) / self.D)
        self.ps = (1 - self.cs) * self.ps + math.sqrt(self.cs * (2 - self.cs) * self.mueff) * invsqrtC_yw
        hsig = np.linalg.norm(self.ps) / math.sqrt(1 - (1 - self.cs) ** (2 * self.counteval / self.lam)) / self.chiN < 1.4 + 2 / (N + 1)
        self.pc = (1 - self.cc) * self.pc + (math.sqrt(self.cc * (2 - self.cc) * self.mueff) * yw if hsig else 0.0)
        rank_mu = (y.T * self.weights) @ y
        self.C = ((1 - self.c1 - self.cmu) * self.C + self.c1 * (np.outer(self.pc, self.pc) + (0.0 if hsig else self.cc * (2 - self.cc)) * self.C)
                  + self.cmu * rank_mu)
        self.sigma *= math.exp((self.cs / self.damps) * (np.linalg.norm(self.ps) / self.chiN - 1))
        self._last_f = f[order]

    @property
    def result(self):
        return Result(self.xbest, self.fbest, self.evals_best, self.counteval, self.iterations, self.mean.copy(), self.sigma * np.sqrt(np.diag(self.C)))

    def stop(self):
        out = {}
        if self.maxiter is not None and self.iterations >= self.maxiter:
            out["maxiter"] = self.maxiter
        if self._last_f is not None and self._last_f[-1] - self._last_f[0] < self.tolfun and self.iterations > 10:
            out["tolfun"] = self.tolfun
        if self.sigma * float(np.max(self.D)) < self.tolx:
            out["tolx"] = self.tolx
        return out

    def disp(self):
        print(f"cmaes: iter {self.iterations} evals {self.counteval} fbest {self.fbest:.6e} sigma {self.sigma:.3e} "
              f"axis ratio {float(np.max(self.D) / np.min(self.D)):.2e}")
