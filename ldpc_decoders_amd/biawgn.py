"""BI-AWGN channel and its LLR decoders -- mirror of the reference's ``src/biawgn.py:10-42``."""
import numpy as np

from . import admm, bpa

noise_var = lambda snr_in_db: 10 ** (-snr_in_db / 10)  # noqa: E731  (src/biawgn.py:10)


class Channel:
    name = "biawgn"

    def __init__(self, snr_in_db):
        self.param = snr_in_db
        self.std_dev = np.sqrt(noise_var(snr_in_db))

    def send(self, x):  # {0,1} -> {-1,+1} + N(0, sigma); consumes numpy's global stream like upstream (src/biawgn.py:17-18)
        return (2 * x - 1) + np.random.normal(0, self.std_dev, x.shape)


class LLR:
    channel = "biawgn"

    def __init__(self, snr_in_db, dec):
        self.param, self.noise_var, self.dec = snr_in_db, noise_var(snr_in_db), dec

    def priors(self, y):
        return -2 * y / self.noise_var  # src/biawgn.py:28 ; positive == bit 0

    def decode(self, y):
        return self.dec.decode(y, self.priors(np.asarray(y)))

    def decode_batch(self, y):
        return self.dec.decode_batch(None if hasattr(y, "is_cuda") else y, self.priors(y))


class SPA(LLR):
    id_keys = bpa.SPA.id_keys

    def __init__(self, snr_in_db, _code, **kwargs):
        super().__init__(snr_in_db, bpa.SPA(_code, **kwargs))


class MSA(LLR):
    id_keys = bpa.MSA.id_keys

    def __init__(self, snr_in_db, _code, **kwargs):
        super().__init__(snr_in_db, bpa.MSA(_code, **kwargs))


class ADMM(LLR):  # src/biawgn.py:52-56
    id_keys = admm.ADMM.id_keys

    def __init__(self, snr_in_db, _code, **kwargs):
        super().__init__(snr_in_db, admm.ADMM(_code, **kwargs))
        self.stats = self.dec.stats

    def decode_batch(self, y):
        return self.dec.decode_batch(self.priors(np.asarray(y)))


from .ml import BiawgnML as ML  # noqa: E402  (src/biawgn.py: class ML)
