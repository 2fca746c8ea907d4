"""Experiment driver -- the build's counterpart of the reference's ``src/main.py``.

    python -m ldpc_decoders_amd.main <channel> <code> <decoder> [--codeword C --min-wec W --params P.. --max-iter I ..]

Same positional/flag grammar, same logger names, same JSON result files (see ``utils``), so the arg-lines emitted by
the reference's ``simulations.py`` run unchanged and ``graph.py`` reads the outputs.  Differences, all additive:
frames are decoded in batches on the GPU (``--batch`` per round and rank, ``--seed`` for the device noise, ``--precision`` /
``--backend`` for the kernels, ``--max-frames`` to stop a parameter whose error rate is too low to reach ``--min-wec``); ``--exact``
selects the reference-exact mode (host noise, sequential stopping rule, ``--np-seed``); under ``torchrun`` each rank drives one GPU
and the counters are all-reduced once per round.
"""
import logging
import time
from collections import OrderedDict

import numpy as np

from . import codes, dist, utils
from .models import models
from .montecarlo import DeviceSimulator, run_point_exact


def test(args, comm=None):
    comm = comm or dist.Comm()
    model = models[args.channel]
    dec_fac = getattr(model, args.decoder)
    id_keys = ["channel", "code", "decoder", "codeword", "min_wec"] + dec_fac.id_keys
    id_val = [vars(args)[key] for key in id_keys]
    log = logging.getLogger(".".join(utils.strl(id_val)))
    code = codes.get_code(args.code)
    code_n = code.get_n()
    saver = utils.Saver(args.data_dir, list(zip(id_keys, id_val))) if comm.is_root else None
    # --codeword -1 (a random word of the code book per frame, src/main.py:38): on the device for the BP decoders, the reference's
    # sequential loop on host noise for the others
    exact = bool(args.exact) or (args.codeword == -1 and args.decoder not in ("SPA", "MSA"))
    if exact and comm.world > 1:
        raise SystemExit("--exact / --codeword -1 follow the reference's sequential rule and run on a single rank")
    if exact and args.np_seed is not None:
        np.random.seed(args.np_seed)
    kwargs = dict(vars(args))
    # fp32 message arithmetic in the throughput mode -- except min-sum over the BSC: every LLR is +-L there, the decoder is
    # tie-dominated and only the reference's fp64 arithmetic reproduces its curves
    tie_dominated = args.channel == "bsc" and args.decoder == "MSA"
    kwargs["precision"] = args.precision or ("f64" if (exact or tie_dominated) else "f32")
    if kwargs["precision"] == "f16" and (args.decoder not in ("SPA", "MSA") or args.channel == "bec" or exact):
        raise SystemExit("--precision f16 (fp16 storage of the messages): the LLR decoders SPA / MSA over biawgn / bsc, device-noise mode")
    results = OrderedDict()

    for pi, param in enumerate(args.params):
        log.info("Starting parameter: %f" % param)
        channel = model.Channel(param)
        decoder = dec_fac(param, code, **kwargs)
        state = dict(t=time.time())

        def log_status(c, final=False):
            tot, wec, bec = c["tot"], c["wec"], c["bec"]
            wer, ber = (wec / tot, bec / (tot * code_n)) if tot else (0., 0.)
            keys = ["tot", "wec", "wer", "bec", "ber"]
            vals = [int(tot), int(wec), float(wer), int(bec), float(ber)]
            if comm.is_root:
                log.info(", ".join("%s:%s" % (k.upper(), v) for k, v in zip(keys, vals)))
            if hasattr(decoder, "stats"):  # e.g. the ADMM iteration histogram (src/main.py:34)
                keys.append("dec"), vals.append(decoder.stats())
            if c.get("capped"):  # an addition: the point was stopped by --max-frames before reaching --min-wec
                keys.append("capped"), vals.append(True)
            if comm.is_root:
                saver.add(param, OrderedDict(zip(keys, vals)))
            return OrderedDict(zip(keys, vals))

        def progress(tot, wec, bec, hist=None):
            if time.time() - state["t"] > args.log_freq:
                state["t"] = time.time()
                if hist is not None and state.get("hist_into") is not None:  # the decoder's own histogram follows the reduced counters
                    into, bins = state["hist_into"]
                    into[:] = 0
                    into[:bins] = hist[:bins]
                log_status(dict(tot=tot, wec=wec, bec=bec))

        inner = decoder if hasattr(decoder, "handle") else getattr(decoder, "dec", None)
        on_device = hasattr(getattr(inner, "handle", None), "simulate")  # BP and ML: channel + decode + count on the GPU
        if not exact and not on_device and comm.world > 1:
            raise SystemExit("decoder %s runs its Monte-Carlo loop on host noise and a single rank" % args.decoder)
        if exact or not on_device:
            pick = None
            if args.codeword == -1:
                pick = lambda: code.cb[np.random.choice(code.cb.shape[0], 1)[0]]  # noqa: E731  (src/main.py:38)
                x = code.cb[0]
            else:
                x = np.zeros(code_n, dtype=np.int64) + args.codeword
            # one frame per call keeps numpy's stream in the reference's order (send, [ML pick], send, ..)
            if exact:
                chunk = 1 if (code_n < 64 or args.decoder in ("ML", "ADMM")) else 32
            else:  # a decoder without a device Monte-Carlo path: host noise, frames decoded in batches
                chunk = max(1, min(args.batch, 4096))
            c = run_point_exact(channel, decoder, x, args.min_wec, chunk=chunk, on_progress=progress, pick_word=pick)
        else:
            handle = decoder.handle if hasattr(decoder, "handle") else decoder.dec.handle
            if args.codeword == -1 and getattr(code, "cb", None) is None:
                raise SystemExit("--codeword -1 draws from the code book, which only the small codes have (as upstream)")
            # a decoder that reports statistics (ADMM: iteration histogram, src/main.py:34) gets them from the REDUCED counters, so
            # that `dec` describes the same frames as tot/wec/bec -- the whole job, not rank 0's shard
            own_hist = hasattr(decoder, "stats") and hasattr(inner, "iter")
            bins = min(len(inner.iter), args.max_iter + 1 if args.max_iter > 0 else len(inner.iter)) if own_hist else 0
            if own_hist:
                handle.on_iters = None
                state["hist_into"] = (inner.iter, bins)  # intermediate result files then carry a real histogram, not zeros
            grid = getattr(args, "prior_grid", None)
            if grid is not None and (args.decoder != "MSA" or args.channel != "biawgn" or kwargs["precision"] != "f32"):
                raise SystemExit("--prior-grid: fp32 min-sum over BI-AWGN (biawgn <code> MSA, without --precision f64)")
            sim = DeviceSimulator(handle, args.channel, args.max_iter, args.codeword, args.seed, comm, hist_bins=bins, prior_grid=grid)
            c = sim.run_point(param, stream_id=pi, min_wec=args.min_wec, batch_per_rank=args.batch, on_progress=progress,
                              max_frames=(args.max_frames or None))
            if own_hist:
                inner.iter[:] = 0
                inner.iter[:bins] = c["hist"]
            if grid is not None and comm.is_root:
                log.info("prior grid 2^-%d: %d of %d frames were beyond the fp32 exactness guard and decoded again in fp64" % (grid, sim.redone, c["tot"]))
        results[param] = log_status(c, final=True)
    log.info("Done!")
    return results


def main(argv=None):
    args = utils.setup_parser(codes.get_code_names(), models.keys(), utils.decoder_names).parse_args(argv)
    comm = dist.init_from_env()
    log_level = logging.DEBUG if args.debug else logging.INFO
    if args.console:
        utils.setup_console_logger(log_level)
    else:
        utils.make_dir_if_not_exists(args.data_dir)
        utils.setup_file_logger(args.data_dir, "test", log_level)
    if comm.is_root:
        print(vars(args))
    try:
        return test(args, comm)
    finally:
        dist.finalize()


if __name__ == "__main__":
    # np.random is deliberately left unseeded, as upstream (src/main.py:68); use --np-seed / --seed for repeatability
    main()
