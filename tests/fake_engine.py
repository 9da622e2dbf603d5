"""Oracle-backed stand-in for deepblast_amd._engine.HipEngine -- TESTS ONLY.

Lets the CPU test-suite exercise the host logic (autograd wiring, gradient quirks, batch
sharding + gather) without a GPU.  The product never constructs this class: the real engine
raises when libsdp_hip.so or a ROCm device is missing.
"""
import numpy as np
import torch

from oracle import oracle


class OracleEngine:
    name = "oracle-fake"

    def max_cols(self):
        return 2048

    @staticmethod
    def _np(t):
        return t.detach().cpu().numpy()

    def _slices(self, B, N, M, lens):
        if lens is None:
            return [(N, M)] * B
        lens = np.asarray(lens.cpu() if isinstance(lens, torch.Tensor) else lens)
        return [(int(lens[b, 0]), int(lens[b, 1])) for b in range(B)]

    def forward(self, theta, A, variant, lens=None, exact_state=False):
        th, a = self._np(theta), self._np(A)
        B, N, M = th.shape
        Vt = np.zeros(B, np.float32)
        Qs = []
        for b, (n, m) in enumerate(self._slices(B, N, M, lens)):
            v, q = oracle.forward(np.ascontiguousarray(th[b:b + 1, :n, :m]),
                                  np.ascontiguousarray(a[b:b + 1, :n, :m]), variant)
            Vt[b] = v[0]
            Qs.append(q)
        state = torch.zeros(1)
        state._oracle_Q = Qs  # opaque, like the real engine's flat state
        return torch.from_numpy(Vt), state

    def backward(self, Et, state, shape, variant, lens=None, exact_state=False, pair_range=None, out=None, no_fill=False):
        B, N, M = shape
        et = self._np(Et).astype(np.float32).reshape(-1)
        et = np.broadcast_to(et, (B,)) if et.size == 1 else et
        lo, hi = (0, B) if pair_range is None else pair_range
        E = np.zeros((hi - lo, N, M), np.float32)
        if not hasattr(state, "_oracle_E") or pair_range is None:
            state._oracle_E = [None] * B
        for b in range(lo, hi):
            q = state._oracle_Q[b]
            e = oracle.backward(et[b:b + 1], q, variant)
            n, m = q.shape[1] - 2, q.shape[2] - 2
            E[b - lo, :n, :m] = e[0, 1:-1, 1:-1]
            state._oracle_E[b] = e
        if out is None:
            return torch.from_numpy(E)
        out[lo:hi] = torch.from_numpy(E)   # like the real engine: only the swept rows are written
        return out

    def adjoint_forward(self, state, Ztheta, ZA, variant, lens=None, ref=False):
        Z = self._np(Ztheta)
        B = Z.shape[0]
        Vtd = np.zeros(B, np.float32)
        Qds = []
        for b, q in enumerate(state._oracle_Q):
            n, m = q.shape[1] - 2, q.shape[2] - 2
            zt = np.zeros((1, n + 2, m + 2), np.float32)
            zt[0, 1:-1, 1:-1] = Z[b, :n, :m]
            za = np.zeros((1, n, m), np.float32) if ZA is None else np.ascontiguousarray(self._np(ZA)[b:b + 1, :n, :m])
            v, qd = oracle.adjoint_forward(q, zt, za)
            Vtd[b] = v[0]
            Qds.append(qd)
        sd = torch.zeros(1)
        sd._oracle_Qd = Qds
        return torch.from_numpy(Vtd), sd

    def adjoint_backward(self, E, state, state_d, variant, lens=None, ref=False):
        B, N, M = E.shape
        Ed = np.zeros((B, N, M), np.float32)
        En = self._np(E)
        for b, (q, qd) in enumerate(zip(state._oracle_Q, state_d._oracle_Qd)):
            n, m = q.shape[1] - 2, q.shape[2] - 2
            e = np.zeros((1, n + 2, m + 2), np.float32)   # the reference's E carries a zero border
            e[0, 1:-1, 1:-1] = En[b, :n, :m]
            ed = oracle.adjoint_backward(e, q, qd)
            Ed[b, :n, :m] = ed[0, 1:-1, 1:-1]
        return torch.from_numpy(Ed)

    def traceback(self, grad, lens=None, rule="cpu"):
        """Per-pair host walks (deepblast_amd/_dp.py::traceback) in the device kernel's output format."""
        from deepblast_amd._dp import traceback
        g = self._np(grad)
        B, N, M = g.shape
        cap = N + M + 2
        states = np.zeros((B, cap, 3), np.int32)
        counts = np.zeros(B, np.int32)
        for b, (n, m) in enumerate(self._slices(B, N, M, lens)):
            try:
                path = traceback(g[b, :n, :m], rule=rule)
            except IndexError:
                counts[b] = -1
                continue
            counts[b] = len(path)
            states[b, :len(path)] = np.asarray(path, np.int32)
        return torch.from_numpy(states), torch.from_numpy(counts)
