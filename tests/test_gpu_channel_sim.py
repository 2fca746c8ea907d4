"""GPU tests of the device channel kernels, the error counters and the fused simulate() pass."""
import numpy as np
import pytest

import bp_oracle as O
import c_oracle as C
from helpers import golden_edges

pytestmark = pytest.mark.gpu


def _setup(name="1200_3_6_rand_ldpc_1", alg="MSA", precision="f32", backend="auto"):
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    return g, code, DecoderHandle(code, alg, precision, backend)


def test_philox_stream_bit_exact_via_discrete_channels():
    # BSC flips and BEC erasures are pure integer functions of the Philox words: must equal the oracle's stream exactly
    g, code, h = _setup()
    seed, stream, frame0, B = 0x1234ABCD5678, 3, (1 << 33) + 17, 9
    for ch, p in (("bsc", 0.11), ("bec", 0.37)):
        hh = h if ch == "bsc" else _setup(alg="BEC")[2]
        _, y = hh.channel_device(ch, p, 1, seed, stream, frame0, B)
        y = y.cpu().numpy()
        thr = int(np.ceil(p * 4294967296.0 - 0.5))
        for f in range(B):
            w = O.philox_frame_words(seed, stream, frame0 + f, g.n).astype(np.uint64)
            hit = w < thr
            want = np.where(hit, 0, 1) if ch == "bsc" else np.where(hit, 2, 1)
            assert (y[f] == want).all()


@pytest.mark.parametrize("prec,tol", [("f64", 1e-9), ("f32", 3e-5)])
def test_biawgn_priors_match_fp64_model(prec, tol):
    g, code, h = _setup(precision=prec)
    seed, stream, frame0, B, snr = 99, 1, 123456789, 6, 2.0
    pri, _ = h.channel_device("biawgn", snr, 0, seed, stream, frame0, B)
    pri = pri.cpu().numpy().astype(np.float64)
    var = O.biawgn_noise_var(snr)
    for f in range(B):
        z = O.device_biawgn_noise(seed, stream, frame0 + f, g.n)
        want = -2 * (-1 + np.sqrt(var) * z) / var
        assert np.max(np.abs(pri[f] - want) / (1 + np.abs(want))) <= tol


def test_biawgn_noise_statistics():
    g, code, h = _setup()
    snr, B = 1.0, 4096
    pri, _ = h.channel_device("biawgn", snr, 0, 7, 0, 0, B)
    var = O.biawgn_noise_var(snr)
    z = (pri.double().cpu().numpy() * var / -2 + 1) / np.sqrt(var)  # back to unit normals
    N = z.size
    assert abs(z.mean()) < 5 / np.sqrt(N)
    assert abs(z.var() - 1) < 5 * np.sqrt(2 / N)
    assert abs((z ** 4).mean() - 3) < 0.05
    assert abs(np.mean(np.abs(z) > 3) - 0.0026998) < 3e-4
    c = np.corrcoef(z[:, :-1].ravel()[:200000], z[:, 1:].ravel()[:200000])[0, 1]
    assert abs(c) < 0.01


@pytest.mark.parametrize("alg,channel,param,prec", [("MSA", "biawgn", 2.0, "f32"), ("MSA", "bsc", 0.04, "f32"), ("BEC", "bec", 0.4, "f32"),
                                                      ("MSA", "biawgn", 2.0, "f64")])
def test_simulate_equals_channel_decode_count_and_oracle(alg, channel, param, prec):
    import torch

    g, code, h = _setup(alg=alg, precision=prec)
    seed, stream, frame0, B, max_iter, bins = 42, 5, 1000, 700, 50, 51
    cnt = torch.zeros(4 + bins, dtype=torch.int64, device="cuda")
    h.simulate(channel, param, 0, seed, stream, frame0, B, max_iter, cnt, hist_bins=bins)
    cnt = cnt.cpu().numpy()
    pri, y = h.channel_device(channel, param, 0, seed, stream, frame0, B)
    if alg == "BEC":
        xo, io = C.bec_decode(g, y.cpu().numpy(), max_iter)
    else:
        dt = np.float64 if prec == "f64" else np.float32
        xo, io = C.bp_decode(g, alg, None if y is None else y.cpu().numpy().astype(dt), pri.cpu().numpy(), max_iter, dtype=dt)
    err = (xo != 0).sum(axis=1)
    assert cnt[0] == B and cnt[1] == (err > 0).sum() and cnt[2] == err.sum() and cnt[3] == io.sum()
    assert (cnt[4:] == np.bincount(np.minimum(io, bins - 1), minlength=bins)).all()


@pytest.mark.parametrize("name,codeword,B", [("1200_3_6_rand_ldpc_1", 0, 2048 + 700), ("1200_3_6_rand_ldpc_1", 1, 61), ("1200_rho_x5_rand_ldpc_5", 0, 4133)])
def test_streaming_erasure_simulate_counts_on_the_bit_planes(name, codeword, B):
    # ldpc_simulate on the streaming erasure decoder never forms a [B, n] byte array: the erasures are drawn straight into the bit planes and
    # the bit errors are counted straight from them.  Counters and histogram must equal channel kernel -> C oracle on the same Philox frames
    # (ragged batches: a partial lane word, a partial supertile; both codewords; regular and irregular degrees).
    import torch

    g, code, h = _setup(name=name, alg="BEC", backend="stream")
    seed, stream, frame0, max_iter, bins = 77, 2, (1 << 32) + 5, 50, 51
    cnt = torch.zeros(4 + bins, dtype=torch.int64, device="cuda")
    h.simulate("bec", 0.41, codeword, seed, stream, frame0, B, max_iter, cnt, hist_bins=bins)
    assert h.last_stats()[0] == "stream"
    cnt = cnt.cpu().numpy()
    _, y = h.channel_device("bec", 0.41, codeword, seed, stream, frame0, B)
    xo, io = C.bec_decode(g, y.cpu().numpy(), max_iter)
    err = (xo != codeword).sum(axis=1)
    assert cnt[0] == B and cnt[1] == (err > 0).sum() and cnt[2] == err.sum() and cnt[3] == io.sum()
    assert (cnt[4:] == np.bincount(np.minimum(io, bins - 1), minlength=bins)).all()
    # accumulation: a second call adds to the same counters
    cnt2 = torch.from_numpy(cnt.copy()).cuda()
    h.simulate("bec", 0.41, codeword, seed, stream, frame0, B, max_iter, cnt2, hist_bins=bins)
    assert (cnt2.cpu().numpy() == 2 * cnt).all()


def test_shard_invariance_of_counters():
    import torch

    g, code, h = _setup()
    B = 1000

    def run(f0, nb):
        c = torch.zeros(4, dtype=torch.int64, device="cuda")
        h.simulate("biawgn", 2.0, 0, 17, 0, f0, nb, 50, c)
        return c.cpu().numpy()

    whole = run(0, B)
    parts = run(0, 333) + run(333, 400) + run(733, 267)
    assert (whole == parts).all()


def test_full_size_properties():
    # BASELINE config 2 size: 65 536 frames of the n=1200 (3,6) code, min-sum, max_iter=50, device noise at 3 dB.
    # Size-independent properties: every frame that left early carries a codeword (zero syndrome); the all-zero word
    # is recovered for nearly all frames; counters agree with an independent recount; iteration counts are in range.
    import torch
    from ldpc_decoders_amd import bpa

    g, code, h = _setup()
    B, snr = 65536, 3.0
    pri, _ = h.channel_device("biawgn", snr, 0, 2024, 0, 0, B)
    xhat, iters = h.decode_device(pri, None, 50)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    h.simulate("biawgn", snr, 0, 2024, 0, 0, B, 50, cnt)
    xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
    assert it.min() >= 1 and it.max() <= 50
    early = it < 50
    syn = code.syndrome(xh[early][:4096])
    assert syn.sum() == 0
    err = (xh != 0).sum(axis=1)
    assert (err > 0).mean() < 0.05
    cnt = cnt.cpu().numpy()
    assert cnt[0] == B and cnt[1] == (err > 0).sum() and cnt[2] == err.sum() and cnt[3] == it.sum()
    # a sample of frames re-decoded by the CPU oracle: bit-exact
    idx = np.r_[0:64, B - 64:B]
    xo, io = C.bp_decode(g, "MSA", None, pri[idx].cpu().numpy(), 50, dtype=np.float32)
    assert (xh[idx] == xo).all() and (it[idx] == io).all()


@pytest.mark.parametrize("channel,param", [("biawgn", 1.9), ("bsc", 0.045)])
def test_fp64_simulate_on_the_lds_kernel_equals_streaming(channel, param):
    # fp64 Monte-Carlo: channel kernel -> fp64 LDS kernel in counting mode  ==  channel kernel -> streaming decode -> counting kernel
    # (identical priors, bit-identical arithmetic => identical counters and iteration histograms, any batch split)
    import torch
    from helpers import golden_edges
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    res = {}
    for be in ("fused", "stream"):
        h = DecoderHandle(code, "MSA", "f64", be)
        cnt = torch.zeros(4 + 41, dtype=torch.int64, device="cuda")
        for a, b in ((0, 1), (1, 700), (700, 3000)):
            h.simulate(channel, param, 1, 99, 3, 5000 + a, b - a, 40, cnt, hist_bins=41)
        assert h.last_stats()[0] == be
        res[be] = cnt.cpu().numpy()
    assert (res["fused"] == res["stream"]).all() and res["fused"][0] == 3000 and 0 < res["fused"][1] < 3000
    assert res["fused"][4:].sum() == 3000


@pytest.mark.parametrize("prec,dt", [("f64", np.float64), ("f32", np.float32)])
def test_whole_batch_against_the_oracle(prec, dt):
    # not a sample: EVERY frame of a 16 384-frame device-noise batch (config 2's code, min-sum, max_iter 50), at an SNR where nearly
    # every frame fails and at one where frames leave after 5..50 sweeps, re-decoded by the C oracle on the host threads --
    # decisions and iteration counts identical, frame for frame, on the LDS-resident backend
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    h = DecoderHandle(Code.from_edges(g.m, g.n, g.chk, g.var), "MSA", prec, "fused")
    B = 16384
    for snr in (1.0, 2.0):
        pri, _ = h.channel_device("biawgn", snr, 0, 77, 3, 0, B)
        xhat, iters = h.decode_device(pri, None, 50)
        xo, io = C.bp_decode(g, "MSA", None, pri.cpu().numpy(), 50, dtype=dt)
        xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
        assert h.last_stats()[0] == "fused"
        assert (it == io).all() and (xh == xo).all(), (prec, snr, int((it != io).sum()), int((xh != xo).any(axis=1).sum()))
        assert len(np.unique(it)) > (3 if snr == 1.0 else 20)


def test_whole_batch_erasure_decoder_against_the_oracle():
    # the same for the erasure decoder (integer arithmetic: exact), device-generated erasures at eps = 0.42 where frames end in
    # stopping sets as often as they decode: every one of 16 384 frames, symbols {0, 1, 2} and sweep counts, both backends
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    B = 16384
    ref = None
    for backend in ("fused", "stream"):
        h = DecoderHandle(code, "BEC", "f32", backend)
        _, y = h.channel_device("bec", 0.42, 0, 5, 1, 0, B)
        xhat, iters = h.decode_device(None, y, 50)
        xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
        assert h.last_stats()[0] == backend
        if ref is None:
            ref = C.bec_decode(g, y.cpu().numpy(), 50)
        assert (xh == ref[0]).all() and (it == ref[1]).all(), backend
    assert 0.05 < (ref[0] == 2).any(axis=1).mean() < 0.95  # stopping sets and complete decodes both present


@pytest.mark.parametrize("name,backend,B,stride,cw", [
    ("1200_3_6_rand_ldpc_1", "auto", 4096, 4096, 0),     # LDS-resident erasure decoder: ONE launch, positions refilled across round boundaries
    ("1200_3_6_rand_ldpc_1", "auto", 1000, 8000, 1),     # ragged rounds (1000 = 31 slabs + 8 frames), rounds of one rank of a sharded job
    ("1200_3_6_rand_ldpc_1", "auto", 37, 37, 0),         # rounds smaller than the 32 x (workgroups) positions of the chip: rows interleave everywhere
    ("7_4_hamming", "auto", 333, 333, 0),
    ("1200_3_6_rand_ldpc_1", "stream", 2048, 2048, 0),   # streaming erasure kernels: round by round behind the same entry point
])
def test_simulate_rounds_rows_equal_round_by_round_calls(name, backend, B, stride, cw):
    """ldpc_simulate_rounds: every counter row must be exactly what ldpc_simulate returns for that round alone (same Philox frames), however
    many rounds share the launch -- the host applies the stopping rule of src/main.py:37 to the rows in order."""
    import torch
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(name)
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    h = DecoderHandle(code, "BEC", "f32", backend)
    R, bins = 11, 51
    rows = torch.zeros((R, 4 + bins), dtype=torch.int64, device="cuda")
    h.simulate_rounds("bec", 0.41, cw, 77, 3, 500, B, R, stride, 50, rows, hist_bins=bins)
    one = torch.zeros((R, 4 + bins), dtype=torch.int64, device="cuda")
    for r in range(R):
        h.simulate("bec", 0.41, cw, 77, 3, 500 + r * stride, B, 50, one[r], hist_bins=bins)
    assert (rows == one).all(), (rows[:, :4].tolist(), one[:, :4].tolist())
    assert (rows[:, 0] == B).all() and int(rows[:, 1].sum()) > 0
    # accumulation (not overwrite), and a single round through the same entry point
    h.simulate_rounds("bec", 0.41, cw, 77, 3, 500, B, 1, stride, 50, rows[:1], hist_bins=bins)
    assert (rows[0] == 2 * one[0]).all()


def test_simulate_rounds_with_more_rounds_in_flight_than_accumulator_slots():
    # One slab of 32 frames per round and thousands of rounds: every workgroup is handed slabs of many different rounds in quick
    # succession, frames of three rounds would be in flight at once -- the kernel's two accumulator slots force the third slab to WAIT
    # (its ticket is held, `pending`) until one of the two rounds has no frame left.  Rows must still be exact, each against its own launch.
    import torch
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    h = DecoderHandle(Code.from_edges(g.m, g.n, g.chk, g.var), "BEC", "f32", "auto")
    R, B, bins = 6000, 32, 51
    rows = torch.zeros((R, 4 + bins), dtype=torch.int64, device="cuda")
    h.simulate_rounds("bec", 0.42, 0, 5, 2, 10_000, B, R, B, 50, rows, hist_bins=bins)
    assert (rows[:, 0] == B).all()
    one = torch.zeros((R, 4 + bins), dtype=torch.int64, device="cuda")
    for r in range(0, R, 7):  # every seventh round alone (857 launches)
        h.simulate("bec", 0.42, 0, 5, 2, 10_000 + r * B, B, 50, one[r], hist_bins=bins)
    assert (rows[::7] == one[::7]).all()
    # and the sum over all rows equals one long launch of the same frame range
    tot = torch.zeros(4 + bins, dtype=torch.int64, device="cuda")
    h.simulate("bec", 0.42, 0, 5, 2, 10_000, R * B, 50, tot, hist_bins=bins)
    assert (rows.sum(dim=0) == tot).all() and int(tot[1]) > 0


def test_simulate_rounds_of_an_llr_decoder_and_forced_flushes():
    # LLR decoders take the round-by-round path; the erasure kernel's 16-bit partial sums are flushed every 4096 frames of a workgroup
    import torch
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges("1200_3_6_rand_ldpc_1")
    code = Code.from_edges(g.m, g.n, g.chk, g.var)
    h = DecoderHandle(code, "MSA", "f32", "auto")
    rows = torch.zeros((3, 4 + 11), dtype=torch.int64, device="cuda")
    h.simulate_rounds("biawgn", 2.0, 0, 5, 1, 0, 1024, 3, 1024, 10, rows, hist_bins=11)
    one = torch.zeros_like(rows)
    for r in range(3):
        h.simulate("biawgn", 2.0, 0, 5, 1, r * 1024, 1024, 10, one[r], hist_bins=11)
    assert (rows == one).all()
    # a long launch of a small code: each of the 1024 workgroups handles > 4096 frames per slot -> forced flushes of the packed sums
    g2 = golden_edges("7_4_hamming")
    hb = DecoderHandle(Code.from_edges(g2.m, g2.n, g2.chk, g2.var), "BEC", "f32", "auto")
    B = 1 << 23
    big = torch.zeros((2, 4 + 20), dtype=torch.int64, device="cuda")
    hb.simulate_rounds("bec", 0.3, 0, 9, 0, 0, B, 2, B, 20, big, hist_bins=20)
    ref = torch.zeros((2, 4 + 20), dtype=torch.int64, device="cuda")
    hs = DecoderHandle(Code.from_edges(g2.m, g2.n, g2.chk, g2.var), "BEC", "f32", "stream")  # bit-plane counters of the streaming kernels: another code path
    for r in range(2):
        hs.simulate("bec", 0.3, 0, 9, 0, r * B, B, 20, ref[r], hist_bins=20)
    assert hb.last_stats()[0] == "fused" and hs.last_stats()[0] == "stream"
    assert (big == ref).all() and int(big[0, 0]) == B
