"""Maximum-likelihood decoding of the short codes on the GPU -- mirror of the reference's ``ML`` classes
(``src/biawgn.py:66-78``, ``src/bsc.py:63-75``, ``src/bec.py:21-36``).

Same constructor and ``decode(y) -> codeword`` call shapes, ``id_keys = []``.  The log-likelihood of every codeword
is evaluated by ``ldpc_ml_decode`` (fp64, the reference's operation order), which returns the SET of maximisers; the
random pick among them (``math_utils.arg_max_rand``, src/math_utils.py:72-74) is made here with the same
``np.random.choice`` call as upstream, so one-frame-at-a-time use consumes numpy's global stream exactly as the
reference does.  ``decode_batch`` on CUDA tensors picks on the device instead.
"""
import numpy as np

from ._device import MlHandle


class MLBase:
    id_keys = []
    channel = None

    def __init__(self, param, _code, **kwargs):
        self.param = param
        self.cb = _code.cb  # AttributeError for codes without a generator matrix, as upstream
        self.n = self.cb.shape[1]
        self.precision = kwargs.get("precision") or "f64"
        self.handle = MlHandle(self.cb, self.channel, self.precision, kwargs.get("device"))
        self.coef = self.coefficients(param)
        self.last_iters = None

    def coefficients(self, param):
        raise NotImplementedError

    def _to_device(self, y):
        import torch

        y = np.atleast_2d(np.asarray(y))
        if self.channel == "biawgn":
            y = np.ascontiguousarray(y, dtype=np.float64 if self.precision == "f64" else np.float32)
        else:
            yi = np.asarray(y)
            if not (((yi >= 0) & (yi <= 2)).all()):
                raise ValueError("symbols must be in {0, 1, 2}")
            y = np.ascontiguousarray(yi, dtype=np.uint8)
        return torch.from_numpy(y).to("cuda:%d" % self.handle.device)

    def tie_sets(self, y):
        """-> (list of maximiser index arrays, best log-likelihood [B]) for host frames y [B,n] / [n]."""
        out = self.handle.decode_device(self._to_device(y), self.coef)
        mask = out["tie_mask"].cpu().numpy().view(np.uint32)
        bits = ((mask[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).reshape(mask.shape[0], -1)[:, :self.handle.K]
        return [np.flatnonzero(b) for b in bits], out["best"].cpu().numpy()

    def decode(self, y):
        max_ind = self.tie_sets(y)[0][0]
        ind = np.random.choice(max_ind, 1)[0]  # src/math_utils.py:72-74
        return self.cb[ind]

    def decode_batch(self, y):
        """[B,n] -> (x_hat [B,n], iters = zeros).  numpy in: frame-by-frame ``np.random.choice`` picks (the reference's
        stream); CUDA tensor in: picks from a device draw, everything stays on the GPU."""
        if hasattr(y, "is_cuda"):
            import torch

            pick = torch.randint(-2 ** 31, 2 ** 31, (y.shape[0],), dtype=torch.int32, device=y.device)
            out = self.handle.decode_device(y, self.coef, pick=pick, want_mask=False)
            self.last_iters = torch.zeros(y.shape[0], dtype=torch.int32, device=y.device)
            return out["xhat"], self.last_iters
        sets, _ = self.tie_sets(y)
        xhat = np.stack([self.cb[np.random.choice(s, 1)[0]] for s in sets]).astype(np.uint8)
        self.last_iters = np.zeros(len(sets), dtype=np.int32)
        return xhat, self.last_iters


class BiawgnML(MLBase):
    channel = "biawgn"

    def coefficients(self, snr_in_db):
        self.noise_var = 10 ** (-snr_in_db / 10)  # src/biawgn.py:10
        return (2 * self.noise_var, 0.0)          # the divisor of src/biawgn.py:75


class BscML(MLBase):
    channel = "bsc"

    def coefficients(self, p):
        with np.errstate(divide="ignore"):
            self.log_p, self.log_1p = np.log(p), np.log(1 - p)  # src/bsc.py:67
        return (self.log_p, self.log_1p)


class BecML(BscML):  # src/bec.py:25
    channel = "bec"
