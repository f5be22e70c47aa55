"""Test-only numpy/oracle backend for rapidcfd-dev_amd/parallel.py.

Implements the phases of mi_dpcg_phase on host arrays so that the exchange / reduction logic
of DistributedPCG can be exercised with the gloo backend on a CPU-only box.  The local Amul is
the oracle's; this file lives under tests/ and is never used by the product path.
"""
import copy

import numpy as np
import torch

SMALL, VSMALL, GREAT = 1e-20, 1e-300, 1e20


class NumpyOps:
    def __init__(self, orc, sub, precond="diagonal"):
        self.sub = sub
        self.n = sub.n_cells
        local = copy.copy(sub)
        local.interfaces = []
        self.S = orc.System([local])
        self.fc = [np.asarray(i.face_cells) for i in sub.interfaces]
        self.bou = [np.asarray(i.bou_coeffs) for i in sub.interfaces]
        self.n_ext = sum(len(f) for f in self.fc)
        self.off = np.concatenate([[0], np.cumsum([len(f) for f in self.fc])]).astype(int)
        nv = self.n + self.n_ext
        z = lambda: torch.zeros(nv, dtype=torch.float64)
        self.psi, self.src, self.pA, self.wA, self.rA = z(), z(), z(), z(), z()
        self.src[: self.n] = torch.from_numpy(sub.source)
        self.scal = torch.zeros(8, dtype=torch.float64)
        self.send = torch.zeros(max(self.n_ext, 1), dtype=torch.float64)
        self.rD = 1.0 / sub.diag if precond == "diagonal" else np.ones(self.n)
        self.st = {}

    def set_initial(self, psi0):
        self.psi.zero_()
        if psi0 is not None:
            self.psi[: self.n] = torch.from_numpy(np.asarray(psi0))

    def begin(self, tolerance, relTol, maxIter, minIter, history_len):
        self.st = dict(tol=tolerance, relTol=relTol, maxIter=maxIter, minIter=minIter, done=0, nIterations=0,
                       converged=0, singular=0, wArA=[GREAT, GREAT], hist=[])

    def _amul(self, x):
        xn = x.numpy()
        y = self.S.amul(xn[: self.n])
        for k, (fc, bou) in enumerate(zip(self.fc, self.bou)):
            np.subtract.at(y, fc, bou * xn[self.n + self.off[k]: self.n + self.off[k + 1]])
        return y

    def _sumA(self):
        y = self.S.sumA()
        for fc, bou in zip(self.fc, self.bou):
            np.subtract.at(y, fc, bou)
        return y

    def _pack(self, x):
        for k, fc in enumerate(self.fc):
            self.send[self.off[k]: self.off[k + 1]] = x[: self.n][torch.from_numpy(fc.astype(np.int64))]

    def _conv(self, res):
        s = self.st
        return res < s["tol"] or (s["relTol"] > SMALL and res < s["relTol"] * s["initialResidual"])

    def _final(self, it):
        s = self.st
        if s["done"]:
            return
        if self.scal[1] < 0:
            s["singular"], s["done"] = 1, 1
            return
        res = float(self.scal[1]) / s["normFactor"]
        s["finalResidual"] = res
        if len(s["hist"]) == it + 1:
            s["hist"].append(res)
        s["nIterations"] = it + 1
        s["converged"] = int(self._conv(res))
        if not ((it < s["maxIter"] and not s["converged"]) or (it + 1 < s["minIter"])):
            s["done"] = 1

    def phase(self, k, it=0, arg=0.0):
        s, n = self.st, self.n
        if k == 0:
            self._pack(self.psi)
        elif k == 1:
            self.wA[:n] = torch.from_numpy(self._amul(self.psi))
            self.rA[:n] = self.src[:n] - self.wA[:n]
            self.pA[:n] = torch.from_numpy(self._sumA())
            self.scal[3] = float(np.sum(self.psi[:n].numpy()))
        elif k == 2:
            t = arg * self.pA[:n].numpy()
            self.scal[4] = float(np.sum(np.abs(self.wA[:n].numpy() - t) + np.abs(self.src[:n].numpy() - t)))
            self.scal[1] = float(np.sum(np.abs(self.rA[:n].numpy())))
            r = self.rA[:n].numpy()
            self.scal[0] = float(np.sum((self.rD * r) * r))
        elif k == 3:
            s["normFactor"] = float(self.scal[4]) + SMALL
            res = float(self.scal[1]) / s["normFactor"]
            s["initialResidual"] = s["finalResidual"] = res
            s["hist"] = [res]
            s["converged"] = int(self._conv(res))
            s["done"] = 0 if (s["minIter"] > 0 or not s["converged"]) else 1
        elif k == 10:
            if it > 0:
                self._final(it - 1)
            if not s["done"]:
                wArA = float(self.scal[0])
                w = self.rD * self.rA[:n].numpy()
                if it == 0:
                    self.pA[:n] = torch.from_numpy(w)
                else:
                    beta = wArA / s["wArA"][(it & 1) ^ 1]
                    self.pA[:n] = torch.from_numpy(w + beta * self.pA[:n].numpy())
                s["wArA"][it & 1] = wArA
            self._pack(self.pA)
        elif k == 11:
            pass
        elif k == 12:
            self.wA[:n] = torch.from_numpy(self._amul(self.pA))
            self.scal[2] = float(np.sum(self.wA[:n].numpy() * self.pA[:n].numpy()))
        elif k == 13:
            if not s["done"]:
                wApA = float(self.scal[2])
                if abs(wApA) / s["normFactor"] < VSMALL:
                    self.scal[1] = -1.0
                else:
                    alpha = s["wArA"][it & 1] / wApA
                    self.psi[:n] += alpha * self.pA[:n]
                    self.rA[:n] -= alpha * self.wA[:n]
                    r = self.rA[:n].numpy()
                    self.scal[1] = float(np.sum(np.abs(r)))
                    self.scal[0] = float(np.sum((self.rD * r) * r))
        elif k == 14:
            self._final(it)
        else:
            raise ValueError(k)

    def status(self, history_len=0):
        s = self.st
        return dict(initialResidual=s["initialResidual"], finalResidual=s["finalResidual"], normFactor=s["normFactor"],
                    nIterations=s["nIterations"], converged=s["converged"], singular=s["singular"], done=s["done"],
                    history=np.array(s["hist"]))

    def solution(self):
        return self.psi[: self.n].numpy().copy()
