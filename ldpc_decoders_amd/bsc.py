"""Binary symmetric channel and its LLR decoders -- mirror of the reference's ``src/bsc.py:11-39``."""
import numpy as np

from . import admm, bpa


class Channel:
    name = "bsc"

    def __init__(self, p):
        self.param = self.p = p

    def send(self, x):
        return (x + (np.random.random(x.shape) < self.p)) % 2  # src/bsc.py:16


class LLR:
    channel = "bsc"

    def __init__(self, p, dec):
        self.param, self.llr, self.dec = p, np.log(1 - p) - np.log(p), dec  # src/bsc.py:21

    def priors(self, y):
        return self.llr * (1 - 2 * y)  # src/bsc.py:25

    def decode(self, y):
        return self.dec.decode(y, self.priors(np.asarray(y)))

    def decode_batch(self, y):
        if hasattr(y, "is_cuda"):
            import torch

            dt = torch.float64 if self.dec.precision == "f64" else torch.float32
            return self.dec.decode_batch(y, (self.llr * (1 - 2 * y.to(dt))).contiguous())
        return self.dec.decode_batch(y, self.priors(np.asarray(y)))


class SPA(LLR):
    id_keys = bpa.SPA.id_keys

    def __init__(self, p, _code, **kwargs):
        super().__init__(p, bpa.SPA(_code, **kwargs))


class MSA(LLR):
    id_keys = bpa.MSA.id_keys

    def __init__(self, p, _code, **kwargs):
        super().__init__(p, bpa.MSA(_code, **kwargs))


class ADMM(LLR):  # src/bsc.py:49-53
    id_keys = admm.ADMM.id_keys

    def __init__(self, p, _code, **kwargs):
        super().__init__(p, admm.ADMM(_code, **kwargs))
        self.stats = self.dec.stats

    def decode_batch(self, y):
        return self.dec.decode_batch(self.priors(np.asarray(y)))


from .ml import BscML as ML  # noqa: E402  (src/bsc.py: class ML)
