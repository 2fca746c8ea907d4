"""Test entry point of bench.py on CPU-only hosts: `python tests/bench_cpu_entry.py <bench.py flags>` == bench.main() with the decoder
handle replaced by a stand-in that draws the device's Philox noise (oracle model) and decodes with the C oracle, on gloo ranks.

Started WITHOUT WORLD_SIZE and with --gpus N > 1 it goes through bench.self_launch -- the path under test (tests/test_dist_cpu.py): the
parent starts torch.distributed.run on THIS script with the same argv, so every rank lands here again.  Test infrastructure only: the
product (bench.py run as a script) always builds the HIP handle and has no CPU path.
BENCH_TEST_BASELINE_CONFIGS=1 swaps the `baseline_configs` table for three tiny workloads and treats the command as the default one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import bp_oracle as O  # noqa: E402
import c_oracle as C  # noqa: E402


class OracleHandle:
    """Stand-in for _device.DecoderHandle: same simulate() contract, frames decoded by the CPU oracle."""

    def __init__(self, code, alg, precision, backend):
        self.code, self.alg = code, alg
        self.g = O.Edges(code.m, code.n, code.edge_chk, code.edge_var)

    def simulate(self, channel, param, codeword, seed, stream_id, frame0, B, max_iter, counters, flags=0, hist_bins=0):
        assert channel == "biawgn" and codeword == 0
        var = O.biawgn_noise_var(param)
        pri = np.stack([-2 * (-1 + np.sqrt(var) * O.device_biawgn_noise(seed, stream_id, f, self.g.n)) / var for f in range(frame0, frame0 + B)])
        xh, it = C.bp_decode(self.g, self.alg, None, pri, max_iter, nthreads=1)
        err = (xh != 0).sum(axis=1)
        counters[0] += B
        counters[1] += int((err > 0).sum())
        counters[2] += int(err.sum())
        counters[3] += int(it.sum())
        if hist_bins:
            counters[4:4 + hist_bins] += torch.from_numpy(np.bincount(np.minimum(it, hist_bins - 1), minlength=hist_bins))

    def last_stats(self):
        return "oracle", 0

    def kernel_name(self, simulate=False):
        return ""


if __name__ == "__main__":
    if os.environ.get("BENCH_TEST_BASELINE_CONFIGS") == "1":
        bench.BASELINE_CONFIGS = [
            ("tiny_spa", "sum-product on the (4,2) test code", ["--code", "4_2_test", "--decoder", "SPA", "--batch", "64", "--snr", "3.0", "--steps", "2", "--warmup", "1"]),
            ("tiny_msa", "min-sum on Hamming(7,4) at 4 dB", ["--code", "7_4_hamming", "--batch", "48", "--snr", "4.0", "--steps", "3", "--warmup", "1"]),
            ("tiny_f64", "min-sum fp64 on Hamming(7,4)", ["--code", "7_4_hamming", "--batch", "32", "--snr", "1.0", "--precision", "f64", "--steps", "2", "--warmup", "0"]),
        ]
        bench.is_default_workload = lambda args: True
    bench.main(make_handle=OracleHandle, device="cpu")
