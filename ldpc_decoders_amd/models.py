"""Decoder registry -- mirror of the reference's ``src/models.py:3`` / ``src/utils.py:16``.

``models[channel]`` is a module exposing ``Channel`` and the decoder classes; ``getattr(models[ch], name)`` selects
one (src/main.py:11-12).  The BP decoders, the exhaustive ML decoder of the short codes and the ADMM LP decoder are built for the GPU; the other
upstream names (LP, ADMMA) resolve to a class that raises on construction.
"""
from . import bec, biawgn, bsc

decoder_names = ["ML", "SPA", "MSA", "LP", "ADMM", "ADMMA"]  # src/utils.py:16


def _unavailable(name):
    class _Unavailable:
        id_keys = []

        def __init__(self, *a, **k):
            raise NotImplementedError("decoder %s is outside the GPU belief-propagation path (SURVEY.md section 8); "
                                      "use SPA, MSA, ML or ADMM" % name)

    _Unavailable.__name__ = name
    return _Unavailable


for _mod in (bsc, bec, biawgn):
    for _name in decoder_names:
        if not hasattr(_mod, _name):
            setattr(_mod, _name, _unavailable(_name))

models = {"bsc": bsc, "bec": bec, "biawgn": biawgn}
