"""Binary erasure channel and its message-passing decoder -- mirror of the reference's ``src/bec.py:11-18,70-125``.

Behind the ``bec`` selector ``SPA`` and ``MSA`` are the same ternary {-1,0,+1} erasure decoder (not ``bpa``), with
the stopping-set exit of src/bec.py:120; symbols are {0, 1, 2 = erased}.
"""
import numpy as np

from . import admm
from ._device import DecoderHandle, as_code


class Channel:
    name = "bec"

    def __init__(self, p):
        self.param = self.p = p

    def send(self, x):
        tt = (np.random.random(x.shape) < self.p).astype(int)  # src/bec.py:17-18
        return np.clip(x + tt * 10, 0, 2)


class SPA:
    id_keys = ["max_iter"]
    channel = "bec"

    def __init__(self, p, _code, **kwargs):
        self.param = p
        self.max_iter = kwargs["max_iter"]
        self.code = as_code(_code)
        self.precision = "f32"  # integer arithmetic; the field only selects staging widths
        self.handle = DecoderHandle(self.code, "BEC", "f32", kwargs.get("backend") or "auto", kwargs.get("device"))
        self.last_iters = None

    def decode(self, y):
        y = np.asarray(y)
        xhat, iters = self.handle.decode_host(None, y, self.max_iter)
        self.last_iters = iters
        return xhat[0].astype(np.int64)

    def decode_batch(self, y):
        if hasattr(y, "is_cuda"):
            out = self.handle.decode_device(None, y, self.max_iter)
        else:
            out = self.handle.decode_host(None, y, self.max_iter)
        self.last_iters = out[1]
        return out


class MSA(SPA):
    pass


class ADMM:  # src/bec.py:38-45,58-62: LLR wrapper with +-1e8 for the known symbols, 0 for an erasure
    id_keys = admm.ADMM.id_keys
    channel = "bec"

    def __init__(self, p, _code, **kwargs):
        self.param, self.dec, safe_inf = p, admm.ADMM(_code, **kwargs), 1e8
        self.llr = np.array([safe_inf, -safe_inf, 0])  # 0 WP1, 1 WP1, 0 OR 1 WP0.5
        self.stats = self.dec.stats

    def decode(self, y):
        return self.dec.decode(y, self.llr[np.asarray(y)])

    def decode_batch(self, y):
        return self.dec.decode_batch(self.llr[np.asarray(y)])


from .ml import BecML as ML  # noqa: E402  (src/bec.py: class ML)
