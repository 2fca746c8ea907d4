"""GPU tests of the Monte-Carlo driver layer (ldpc_decoders_amd.montecarlo / dist): the path main.py really takes, the pipelined
rounds, and RCCL itself on one GPU."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import golden_edges

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _handle(alg="MSA", prec="f32", backend="auto", code="1200_3_6_rand_ldpc_1"):
    from ldpc_decoders_amd._device import DecoderHandle
    from ldpc_decoders_amd.codes import Code

    g = golden_edges(code)
    return DecoderHandle(Code.from_edges(g.m, g.n, g.chk, g.var), alg, prec, backend)


@pytest.mark.parametrize("alg,prec,channel,param", [("MSA", "f32", "biawgn", 2.0), ("MSA", "f64", "biawgn", 2.0), ("SPA", "f32", "bsc", 0.06),
                                                     ("BEC", "f32", "bec", 0.40), ("MSA", "f64", "bsc", 0.04)])
def test_main_py_path_is_the_in_kernel_path(alg, prec, channel, param):
    # main.py builds its DeviceSimulator WITHOUT a histogram (hist_bins = 0).  That call must take the same in-kernel
    # channel -> decode -> count path that bench.py times (hist_bins = max_iter + 1), and both must equal the composition
    # channel kernel -> ldpc_decode -> ldpc_count_errors on the streaming backend (same precision => same bits).
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    outs = {}
    for name, backend, bins in (("main", "auto", 0), ("bench", "auto", 51), ("composed", "stream", 0)):
        h = _handle(alg, prec, backend)
        sim = DeviceSimulator(h, channel, 50, 0, 1234, hist_bins=bins)
        outs[name] = sim.run_round(param, 3, 100, 4096)[:4].tolist()
        if backend == "auto":
            assert h.fused_info()["waves_per_frame"] > 0
    assert outs["main"] == outs["bench"], outs
    if alg == "SPA" and prec == "f32":
        # fp32 sum-product: the LDS kernel visits a check's edges in the order its layout plan chose, the streaming kernel in H's
        # order; the (E, O) joins round differently in the last bit, which may flip a bit of a non-converging frame
        a, b = np.array(outs["main"], dtype=float), np.array(outs["composed"], dtype=float)
        assert a[0] == b[0] and np.all(np.abs(a[1:] - b[1:]) <= 0.02 * b[1:] + 2), outs
    else:
        assert outs["main"] == outs["composed"], outs
    assert outs["main"][0] == 4096 and outs["main"][3] > 4096


def test_pipelined_rounds_equal_the_synchronous_loop():
    from ldpc_decoders_amd.montecarlo import DeviceSimulator

    h = _handle()
    sim = DeviceSimulator(h, "biawgn", 50, 0, 99, hist_bins=51)
    got = sim.run_point(2.3, 1, min_wec=60, batch_per_rank=512)
    tot = np.zeros(55, dtype=np.int64)
    frame0 = 0
    while tot[1] < 60:
        tot += sim.run_round(2.3, 1, frame0, 512)
        frame0 += 512
    assert (got["tot"], got["wec"], got["bec"], got["iter_sum"]) == tuple(int(v) for v in tot[:4])
    assert got["hist"] == tot[4:].tolist() and got["tot"] >= 3 * 512  # several rounds, so the pipeline really overlapped


WORKER = r'''
import json, os, sys
sys.path[:0] = [%(root)r, %(root)r + "/tests", %(root)r + "/oracle"]
import torch
from ldpc_decoders_amd import dist
from ldpc_decoders_amd.montecarlo import DeviceSimulator
from test_gpu_montecarlo import _handle
comm = dist.init_from_env()
import torch.distributed as td
sim = DeviceSimulator(_handle(), "biawgn", 50, 0, 7, comm, hist_bins=51)
a = sim.run_point(2.2, 0, min_wec=40, batch_per_rank=4096)
# the block forms (several rounds per launch: [rounds, k] counters through ONE all-reduce): erasure decoder, exact-in-fp32 mode
simb = DeviceSimulator(_handle("BEC", "f32"), "bec", 50, 0, 7, comm, hist_bins=51)
b = simb.run_point(0.42, 0, min_wec=3000, batch_per_rank=2048)
sime = DeviceSimulator(_handle("MSA", "f32"), "biawgn", 50, 0, 7, comm, hist_bins=51, prior_grid=8)
e = sime.run_point(2.0, 0, min_wec=900, batch_per_rank=2048)
t = torch.arange(5, dtype=torch.int64, device="cuda") * (1 << 40)
comm.all_reduce_sum(t)
json.dump({"backend": comm.backend, "group": comm.group, "initialized": td.is_initialized(), "td_backend": td.get_backend() if td.is_initialized() else None,
           "point": a, "bec": b, "exact": e, "rpl": [simb.rounds_per_launch(), sime.rounds_per_launch()], "big": t.cpu().tolist(), "max": comm.max_float(1.5)},
          open(sys.argv[1], "w"))
dist.finalize()
'''


@pytest.mark.timeout(900)
def test_rccl_all_reduce_of_the_counters_on_one_gpu(tmp_path):
    # De-risks the multi-GPU run on a 1-GPU box: a process group of ONE rank on the "nccl" backend (= RCCL on ROCm), so the int64
    # SUM all-reduce of the device-resident counters, the async (stream-ordered) form used by the pipelined rounds, the MAX
    # reduce of bench.py's timing and the barrier all execute inside RCCL.  Counters must equal the run without a group.
    outs = {}
    for name, extra in (("plain", {}), ("rccl", {"LDPC_DIST_FORCE_GROUP": "1", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0",
                                                 "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29741"})):
        env = dict(os.environ, **extra)
        env.pop("LDPC_DIST_BACKEND", None)
        path = str(tmp_path / (name + ".json"))
        r = subprocess.run([sys.executable, "-c", WORKER % {"root": ROOT}, path], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[name] = json.load(open(path))
    assert outs["rccl"]["backend"] == "nccl" and outs["rccl"]["group"] and outs["rccl"]["td_backend"] == "nccl"
    assert not outs["plain"]["initialized"]
    assert outs["rccl"]["point"] == outs["plain"]["point"] and outs["rccl"]["point"]["wec"] >= 40
    assert outs["rccl"]["rpl"] == [32, 8] and outs["rccl"]["bec"] == outs["plain"]["bec"] and outs["rccl"]["exact"] == outs["plain"]["exact"]
    assert outs["rccl"]["bec"]["wec"] >= 3000 and outs["rccl"]["bec"]["tot"] % 2048 == 0 and outs["rccl"]["exact"]["wec"] >= 900
    assert outs["rccl"]["big"] == [i << 40 for i in range(5)] and outs["rccl"]["max"] == 1.5


@pytest.mark.parametrize("channel,param,alg", [("biawgn", 3.0, "SPA"), ("bsc", 0.05, "MSA"), ("bec", 0.3, "SPA")])
def test_random_codeword_per_frame_on_the_device(channel, param, alg):
    # --codeword -1 (src/main.py:38): every frame sends a random word of the code book.  The device channel picks word
    # floor(w K / 2^32), w = first Philox word of block 0xFFFFFFFE of the frame (checked against the oracle's integer stream), the
    # received values are that word + the same noise the all-zero path draws, and the counters equal a host recount.
    import torch

    import bp_oracle as O
    from ldpc_decoders_amd import _lib, codes
    from ldpc_decoders_amd._device import DecoderHandle

    code = codes.get_code("7_4_hamming")
    cb = np.ascontiguousarray(code.cb, dtype=np.uint8)
    K, n, B, seed, stream, frame0 = cb.shape[0], code.n, 4096, 99, 5, 1000
    bp_alg = "BEC" if channel == "bec" else alg
    h = DecoderHandle(code, bp_alg, "f64", "auto")
    lib = _lib.load()
    cbd = torch.from_numpy(cb).cuda()
    pri = None if channel == "bec" else torch.empty((B, n), dtype=torch.float64, device="cuda")
    y = None if channel == "biawgn" else torch.empty((B, n), dtype=torch.uint8, device="cuda")
    sent = torch.empty((B, n), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(lib.ldpc_channel_words(_lib.CHANNEL[channel], 1, param, cbd.data_ptr(), K, seed, stream, frame0, B, n,
                                      None if pri is None else pri.data_ptr(), None if y is None else y.data_ptr(), sent.data_ptr(), st))
    sent_h = sent.cpu().numpy()
    want_idx = np.array([(int(O.philox4x32(np.array([[0xFFFFFFFE, stream, (frame0 + f) & 0xFFFFFFFF, (frame0 + f) >> 32]], dtype=np.uint32),
                                           np.array([[seed & 0xFFFFFFFF, seed >> 32]], dtype=np.uint32))[0, 0]) * K) >> 32 for f in range(B)])
    assert (sent_h == cb[want_idx]).all() and len(np.unique(want_idx)) == K  # every word of the book is drawn
    # same noise as the all-zero path, shifted by the word: BI-AWGN priors differ by -2(2x)/sigma^2 per bit; BSC / BEC symbols by the word
    p0, y0 = h.channel_device(channel, param, 0, seed, stream, frame0, B)
    if channel == "biawgn":
        var = O.biawgn_noise_var(param)
        assert np.allclose(pri.cpu().numpy(), p0.cpu().numpy() - 4.0 * sent_h / var, rtol=1e-12, atol=1e-12)
    elif channel == "bsc":
        assert ((y.cpu().numpy() ^ y0.cpu().numpy()) == sent_h).all()
    else:
        yy, y00 = y.cpu().numpy(), y0.cpu().numpy()
        assert ((yy == 2) == (y00 == 2)).all() and (yy[yy != 2] == sent_h[yy != 2]).all()
    # whole path: DecoderHandle.simulate(codeword=-1) == decode of those very frames + host recount
    cnt = torch.zeros(4 + 11, dtype=torch.int64, device="cuda")
    h.simulate(channel, param, -1, seed, stream, frame0, B, 10, cnt, hist_bins=11)
    xhat, iters = h.decode_device(pri, y, 10)
    xh, it = xhat.cpu().numpy(), iters.cpu().numpy()
    err = (xh != sent_h).sum(axis=1)
    c = cnt.cpu().numpy()
    assert c[0] == B and c[1] == (err > 0).sum() and c[2] == err.sum() and c[3] == it.sum()
    assert (c[4:] == np.bincount(np.minimum(it, 10), minlength=11)).all()
    assert 0 < c[1] < B  # the operating point has both outcomes


def test_main_py_random_codeword_runs_on_the_device(tmp_path):
    # `main.py bsc 7_4_hamming MSA --codeword -1` without --exact: device Monte-Carlo; WER close to the all-zero word's (min-sum over
    # the BSC is symmetric in the sent word up to its tie rule)
    from ldpc_decoders_amd import main as M

    res = {}
    for cw in ("-1", "0"):
        r = M.main(["biawgn", "7_4_hamming", "SPA", "--codeword", cw, "--min-wec", "3000", "--max-iter", "10", "--params", "3", "--seed", "7",
                    "--data_dir", str(tmp_path), "--console"])
        res[cw] = r[3.0]
    assert res["-1"]["wec"] >= 3000 and res["0"]["wec"] >= 3000
    # fp32 sum-product over BI-AWGN is codeword-symmetric: both runs estimate the same word-error rate
    assert abs(res["-1"]["wer"] - res["0"]["wer"]) < 0.1 * res["0"]["wer"]


def test_kernel_name_and_profile_classes():
    # what the measurement tools rely on: the library names the LDS-resident kernel a decoder launches exactly as the code object (and
    # rocprofv3) does -- the key of its committed PMC counters --, and the HIP-event classes cover the whole streaming decode
    import sys

    import torch

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import kernel_resources
    from ldpc_decoders_amd._device import DecoderHandle

    h = _handle()
    built = {k.split("(")[0] for k in kernel_resources.kernels_of()}
    for sim in (False, True):
        name = h.kernel_name(sim)
        assert name.startswith("k_fused_") and name in built, name
    assert ", true, " in h.kernel_name(True) and ", false, " in h.kernel_name(False)
    hs = DecoderHandle(h.code, "MSA", "f32", "stream")
    assert hs.kernel_name(True) == ""
    hs.set_profiling(True)
    hs.read_profile(reset=True)
    cnt = torch.zeros(4, dtype=torch.int64, device="cuda")
    hs.simulate("biawgn", 2.0, 0, 5, 0, 0, 2048, 50, cnt)
    prof = hs.read_profile(reset=True)
    hs.set_profiling(False)
    cn, vn, tot = prof["stream_check_pass"], prof["stream_variable_pass"], prof["stream_decode_total"]
    assert cn[1] == vn[1] > 5 and tot[1] == 1 and tot[0] >= cn[0] + vn[0] > 0 and prof["fused_decode"][1] == 0, prof
