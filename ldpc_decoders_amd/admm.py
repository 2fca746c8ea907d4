"""ADMM LP decoding on the GPU -- mirror of the reference's ``src/admm.py`` (class ``ADMM``; ``ADMMA``, its neural-network
approximation of the projection, is not built).

Same constructor keywords (``mu``, ``eps``, ``max_iter``, ``allow_pseudo``; the rest of the CLI flags are ignored),
``id_keys``, ``decode(y, gamma) -> estimate`` and ``stats()`` (iteration histogram, saved under ``dec`` by the driver).
The iteration runs in ``ldpc_admm_decode`` (fp64, upstream operation order, bit-identical estimates and iteration counts);
``pseudo_to_cw`` (src/math_utils.py:28-34) is applied here.
"""
import numpy as np

from ._device import AdmmHandle, as_code


def pseudo_to_cw(x_, allow_pseudo, eps=1e-8):  # src/math_utils.py:28-34
    if allow_pseudo:
        x_[x_ < eps] = 0
        x_[1 - x_ < eps] = 1
        return x_
    return (x_ > .5).astype(int)


class ADMM:
    id_keys = ["mu", "eps", "max_iter", "allow_pseudo"]

    def __init__(self, parity_mtx, **kwargs):
        self.allow_pseudo = kwargs["allow_pseudo"]  # KeyError if absent, as upstream (src/admm.py:13-14)
        self.mu, self.max_iter, self.eps = kwargs["mu"], kwargs["max_iter"], kwargs["eps"]
        self.code = as_code(parity_mtx)
        self.handle = AdmmHandle(self.code, kwargs.get("device"))
        self.handle.mu, self.handle.eps, self.handle.allow_pseudo = self.mu, self.eps, bool(self.allow_pseudo)
        self.handle.on_iters = lambda it: self._count(it.cpu().numpy())  # device Monte-Carlo feeds the same histogram
        self.iter = np.zeros(2000, dtype=int)  # src/admm.py:36
        self.last_iters = None

    def stats(self):  # src/admm.py:38-40
        total = self.iter.sum()
        avg = self.iter @ np.arange(len(self.iter)) / total if total else 0.0  # (an intermediate progress line of a device run)
        return {"average": avg, "iter": self.iter.tolist()}

    def _count(self, iters):
        np.add.at(self.iter, np.minimum(np.asarray(iters), len(self.iter) - 1), 1)  # src/admm.py:49

    def decode(self, y, gamma):
        return self.decode_batch(np.atleast_2d(np.asarray(gamma, dtype=np.float64)))[0][0]

    def decode_batch(self, gamma):
        """gamma [B,n] (numpy or CUDA float64) -> (estimates [B,n] after pseudo_to_cw, iters [B]); numpy in -> numpy out."""
        import torch

        host = not hasattr(gamma, "is_cuda")
        g = torch.from_numpy(np.ascontiguousarray(gamma, dtype=np.float64)).cuda() if host else gamma
        x, iters, _ = self.handle.decode_device(g, self.mu, self.eps, self.max_iter)
        self.last_iters = iters.cpu().numpy()
        self._count(self.last_iters)
        if host:
            return pseudo_to_cw(x.cpu().numpy(), self.allow_pseudo), self.last_iters
        if self.allow_pseudo:
            x = torch.where(x < 1e-8, torch.zeros_like(x), x)
            x = torch.where(1 - x < 1e-8, torch.ones_like(x), x)
            return x, iters
        return (x > .5).to(torch.uint8), iters
