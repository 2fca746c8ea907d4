"""ldpc_decoders_amd -- MI355X (gfx950) belief-propagation LDPC decoding behind the decoder registry of
thadikari/ldpc_decoders (``models[channel].{SPA,MSA}``, ``main.py <channel> <code> <decoder>``).

Only the BP hot path is built: channel -> LLR -> flooding SPA/MSA (erasure decoder for the
BEC) with syndrome early exit -> error counting.  The arithmetic runs in hand-written HIP kernels reached through
the C ABI of ``include/ldpc_hip.h``; there is no CPU fallback -- importing a decoder without the built library
raises.
"""
__version__ = "0.1.0"
