#!/usr/bin/env python3
"""Benchmark of the LSTM-GAN G+D training hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one Gen-A training iteration (reference source/trainer.py:142-194) on one batch of
synthetic windows: window gather from the HBM-resident series, critic step (critic=1, batch =
forecast_len windows, fresh RMSprop), discriminator phase (+Adam), generator phase (+Adam) and,
for N > 1, the gradient / loss-sum all-reduces.  Workload = BASELINE.json configs[1]: 32 variables
(+4 time features), window 128, hidden 256, 1 layer, batch 512 PER GPU (weak scaling).  Operands,
results and accumulation are float32; products run as split-precision bf16 MFMA (hi/lo, three MFMAs per
product, ~1e-6 of an fp32 reference): the parity mode IS the measured mode.

Rank 0 prints ONE JSON line (contract in the task description) carrying `roofline` (dominant
kernel, HIP-event timed on its own stream inside the timed region) and, at N=1, `cpu_baseline`
(the stock-PyTorch CPU port of the same iteration on the host cores, bounded sample).
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

CONFIGS = {
    # name: V, time features, To, Tf, H, L, batch per GPU, series length
    "cfg1": dict(V=7, NT=4, To=48, Tf=48, H=64, L=1, B=1024, N=20000),
    "cfg2": dict(V=32, NT=4, To=128, Tf=128, H=256, L=1, B=512, N=100000),
    # BASELINE.json configs[3], long context (not the headline; the critic batch is forecast_len = 1024 windows)
    "cfg4": dict(V=32, NT=4, To=1024, Tf=1024, H=512, L=2, B=64, N=12000),
    # BASELINE.json configs[4], wide multivariate cleaning (cleaner path only: no training step).  A step = one launch
    # of B windows: gather -> G forward -> band -> adjust.  --rows sets the series length (10M in BASELINE.json;
    # the default keeps the run short)
    "cfg5": dict(V=512, NT=4, To=48, Tf=48, H=64, L=1, B=8192, N=1000000, mode="clean"),
}
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algo_bytes_per_window(V, NT, To, Tf, H, L, **_):
    """per LSTM layer fwd+bwd = 4*T*(3*in + 12*h) bytes; one iteration runs every G
    layer 2x and every D layer 3x (critic excluded, heads/losses/optimizers excluded)."""
    E = V + NT
    g = 0
    for (i, h) in ((E, H // 4), (H // 4, H // 2), (H // 2, H)):
        g += 4 * To * (3 * i + 12 * h) + (L - 1) * 4 * To * (3 * h + 12 * h)
    d = 4 * Tf * (3 * V + 12 * H) + (L - 1) * 4 * Tf * (3 * H + 12 * H)
    return 2 * g + 3 * d


class _Log:
    def info(self, *a, **k):
        pass

    warn = debug = info


def pmc_traffic(family):
    """HBM bytes per launch of a kernel family from the latest committed rocprofv3 PMC summary
    (profiles/r*_traffic.json, made by tools/gpu_profile.sh + tools/pmc_summary.py from separate
    FETCH_SIZE / WRITE_SIZE passes of this same command; counters cannot be read inside this process)."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json"))
                   if re.fullmatch(r"r\d+_traffic\.json", os.path.basename(f)))   # not the per-config ones (r03_cfg4_..., r03_clean_...)
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            fam = json.load(f)["families"].get(family)
        return (round(fam["hbm_bytes_per_launch"], 1), os.path.basename(files[-1])) if fam else (None, None)
    except Exception:
        return None, None


def pmc_mfma():
    """bf16 MFMA issue rate of the GEMM family from the latest committed counter summary (profiles/r*_mfma.json:
    tools/gpu_profile.sh's SQ_INSTS_VALU_MFMA_MOPS_BF16 pass of this same command + tools/mfma_summary.py)."""
    import glob
    import re
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r*_mfma.json"))
                   if re.fullmatch(r"r\d+_mfma\.json", os.path.basename(f)))
    if not files:
        return None
    try:
        with open(files[-1]) as f:
            z = json.load(f)
        fam = z["families"]["gemm"]
        big = max((k for k in z["kernels"] if "gemm" in k["kernel"]), key=lambda k: k["avg_us"] * k["launches"])
        return {"bound": "mfma", "kernel": "gemm_bf16x3_kernel (all instantiations of one step)",
                "achieved": round(fam["bf16_tflops"], 1), "peak": z["peak_bf16_tflops"], "unit": "TFLOP/s",
                "frac": round(fam["mfma_frac"], 4), "fp32_equivalent_tflops": round(fam["bf16_tflops"] / 3, 1),
                "largest_instantiation": {"kernel": big["kernel"], "bf16_tflops": round(big["bf16_tflops"], 1),
                                          "mfma_busy": None if big["mfma_busy"] is None else round(big["mfma_busy"], 4)},
                "source": os.path.basename(files[-1]) + " (rocprofv3 counter pass; every fp32-grade product is three "
                          "bf16 MFMAs, so useful flops are a third of the issue rate)"}
    except Exception:
        return None


def build(cfg, device, seed=31):
    from timeband_amd import synthetic
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.loss import TIMEBANDLoss
    from timeband_amd.metric import TIMEBANDMetric
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.trainer import TIMEBANDTrainer
    from timeband_amd.utils.initiate import seeding
    seeding(seed)
    V, To, Tf, H, L, B, N = (cfg[k] for k in ("V", "To", "Tf", "H", "L", "B", "N"))
    log = _Log()
    core = dict(logger=log, device=device, batch_size=B, observed_len=To, forecast_len=Tf, directory="", data_name="synthetic")
    values = synthetic.make_series(N, V, seed=seed)
    tfeat = synthetic.time_features(N + Tf)
    ds = TIMEBANDDataset.from_arrays(values, tfeat, {**core, "stride": 1, "split_rate": 0.9, "min_valid_scale": 2,
                                                     "sliding_step": 0})
    trainset, validset = ds.prepare_dataset()
    models = TIMEBANDModel({**core, "save_opt": False, "load_opt": False, "best_score": 50.0, "hidden_dim": H,
                            "layers_num": L, "models_path": "/tmp/timeband_models"})
    models.initiate(ds.dims)
    losses = TIMEBANDLoss({**core, "l1_weight": 50, "l2_weight": 50, "gp_weight": 50})
    metric = TIMEBANDMetric(core)
    trainer = TIMEBANDTrainer({**core, "models": dict(reload=False, reload_interval=50, save_interval=10),
                               "critic": 1, "epochs": 1,
                               "learning_rate": dict(base=2e-4, decay=0.9, gammaD=1.0, gammaG=1.0),
                               "band_width": 1, "missing_gamma": 0.2, "anomaly_gamma": 0.0, "print_interval": 1},
                              ds, models, metric, losses, None)
    trainer.new_epoch()
    return ds, trainset, trainer


def run_clean(args, cfg, device):
    """cfg5: the cleaning / imputation pass (reference source/cleaner.py:55-128) over a device-resident series."""
    from timeband_amd import _lib, ops, synthetic
    from timeband_amd.dataset import TIMEBANDDataset
    from timeband_amd.model import TIMEBANDModel
    from timeband_amd.postprocess import BandAdjust
    from timeband_amd.utils.initiate import seeding
    from timeband_amd.utils.lstm_layer import LSTMGenerator
    V, To, Tf, H, L, B = (cfg[k] for k in ("V", "To", "Tf", "H", "L", "B"))
    N = args.rows or cfg["N"]
    seeding(31)
    lib = _lib.load()
    core = dict(logger=_Log(), device=device, batch_size=B, observed_len=To, forecast_len=Tf, directory="", data_name="synthetic")
    t0 = time.perf_counter()
    ds = TIMEBANDDataset.from_arrays(synthetic.LazySeries(N, V, 31, device), synthetic.LazyTimeFeatures(N + Tf),
                                     {**core, "stride": 1, "split_rate": 0.9, "min_valid_scale": 2, "sliding_step": 0})
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    netG = LSTMGenerator(V + cfg["NT"], V, To, Tf, H, L, device).to(device)
    full = ds.prepare_dataset(split=False)
    nlaunch = len(full) // B
    if nlaunch < 1:
        raise SystemExit("series too short for one launch")
    post = BandAdjust(ds, "cleaner", 1.0, 0.2, 0.3, keep_outputs=True)
    state = {"i": 0, "sweep": post.sweep(netG, ds.series_dev, ds.stride)}

    def flush():                   # the last pushed batch's band / adjust (the sweep is a two-stage pipeline)
        with torch.no_grad():
            state["sweep"].flush()

    def step():
        if state["i"] == nlaunch:
            flush()
            state["i"] = 0
            post.reset()
        i0 = state["i"] * B
        state["i"] += 1
        with torch.no_grad():
            if args.gathered:      # the round-2 path: materialise the encoder windows, project every (window, step) row
                post.step(netG(full.gather(i0, B)["encoded"])[:, :Tf])
            else:                  # the cleaner's path (TIMEBANDCleaner.clean_arrays): windows read in place, pipelined
                first = full.row0 + To + (full.w0 + i0) * ds.stride
                state["sweep"].push(first, B)

    kids = {"lstm_fwd": _lib.K_LSTM_FWD, "gemm": _lib.K_GEMM, "band": _lib.K_BAND, "gather": _lib.K_GATHER}

    def read(k):
        n, ms, by = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
        lib.tb_prof_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
        lib.tb_prof_enable(k, 0)
        return dict(launches=n.value, total_ms=ms.value, algo_bytes=by.value)

    for _ in range(args.warmup):
        step()
    flush()
    torch.cuda.synchronize()
    ops.check_errors(device)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    flush()                        # inside the timed region: every pushed batch is post-processed
    t_enq = time.perf_counter() - t0   # the host has queued everything; what follows is the GPU catching up
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ops.check_errors(device)
    extra = min(5, args.steps)
    for k in kids.values():
        lib.tb_prof_enable(k, 1)
    for _ in range(extra):
        step()
    flush()
    torch.cuda.synchronize()
    prof = {name: read(k) for name, k in kids.items()}
    E = V + cfg["NT"]
    # SURVEY.md 8(d), clean path: 4 * sum over LSTM layers of To * (in + h) bytes per window
    abw = 4 * To * sum(i + h for i, h in ((E, H // 4), (H // 4, H // 2), (H // 2, H))) * L
    value = B * args.steps / dt
    dom = max(prof, key=lambda k: prof[k]["total_ms"])
    # HBM bytes of one launch from the committed counter summary (profiles/r*_clean_traffic.json: tools/clean_profile.sh)
    import glob
    traffic = traffic_src = None
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_clean_traffic.json")))
    if files and not args.gathered:
        try:
            with open(files[-1]) as f:
                traffic, traffic_src = round(json.load(f)["hbm_bytes_per_step"], 1), os.path.basename(files[-1])
        except Exception:
            pass
    out = {"metric": "cleaning windows/sec", "value": round(value, 1), "unit": "windows/s", "n_gpus": 1, "rccl_ranks": 0,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 4),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "rows_per_s": round(value, 1), "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 4),
           "config": {"workload": f"cfg5: cleaning pass ({'window gather + ' if args.gathered else 'windows in place, '}G forward + band + adjust), V={V} E={E} window={To} "
                                  f"hidden={H} {B} windows/launch, series of {N} rows resident in HBM",
                      "series_rows": N, "windows": len(full), "dataset_setup_s": round(setup_s, 2)},
           "roofline": {"bound": "hbm", "kernel": "whole cleaning launch (all kernels of one step)",
                        "achieved": round(abw * value / 1e9, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(abw * value / 1e9 / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_source": traffic_src,
                        "traffic_note": "per launch of B windows, all kernels of the step",
                        "algo_bytes_per_window": abw, "dominant_family": dom},
           "kernel_ms_per_step": {k: round(v["total_ms"] / max(1, extra), 4) for k, v in prof.items()}}
    print(json.dumps(out), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="cfg2", choices=sorted(CONFIGS))
    ap.add_argument("--rows", type=int, default=0, help="series length (cfg5; default: the config's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--full", action="store_true", help="also time band/adjust/scores inside the step")
    ap.add_argument("--gathered", action="store_true", help="cfg5: gather the encoder windows (round-2 path) instead of reading them in place")
    ap.add_argument("--share-gpass", action="store_true",
                    help="NOT the headline schedule: one generator forward/backward per iteration instead of the "
                         "reference's two identical forwards (see TIMEBANDTrainer.share_gpass)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started plainly with --gpus N: launch the N ranks (one process per GPU) as children BEFORE anything
        # here touches the GPU, and pass their exit code on
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
        raise SystemExit(subprocess.call(cmd))

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no GPU visible)")
    if local >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has no GPU (LOCAL_RANK {local}, {torch.cuda.device_count()} visible)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import torch.distributed as dist
    if world > 1:
        from timeband_amd import dp as _dp
        _dp.init("nccl", rank=rank, world_size=world, device_id=device)      # sets the RCCL channel cap first
        assert dist.get_world_size() == args.gpus

    cfg = CONFIGS[args.config]
    if cfg.get("mode") == "clean":
        if world > 1:
            raise SystemExit("cfg5 (cleaning) is a single-GPU measurement here")
        return run_clean(args, cfg, device)
    ds, trainset, trainer = build(cfg, device)
    B = cfg["B"]
    if args.share_gpass:
        trainer.share_gpass = True
    from timeband_amd import _lib, ops
    lib = _lib.load()
    nbatches = len(trainset) // (B * world)
    if nbatches < 1:
        raise SystemExit("series too short for this batch / world size")
    from timeband_amd.dataset import WindowSet
    # the product's own sharding: global batches of B * world windows, this rank's contiguous slice of each
    # (weak scaling: B windows per GPU); whole batches only, cycled
    loader = WindowSet(ds, trainset.w0, nbatches * B * world, trainset.row0).loader(B * world, rank, world)
    state = {"it": iter(loader), "seen": 0}

    def step(i):
        if state["seen"] == nbatches:
            state["it"], state["seen"] = iter(loader), 0
            trainer.join_post()
            trainer.post.reset()
        batch = next(state["it"])
        state["seen"] += 1
        part = batch.get("part")
        fake = trainer.iteration(batch["encoded"], batch["decoded"], training=True,
                                 n_global=None if part is None else part[1])
        if args.full:
            trainer.post_process(fake, part)     # band / adjust / scores: on one rank under the next iteration (side stream)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    kids = {"lstm_fwd": _lib.K_LSTM_FWD, "lstm_bwd": _lib.K_LSTM_BWD, "gemm": _lib.K_GEMM}

    def read(k, off=True):
        n, ms, by = ctypes.c_int(0), ctypes.c_double(0), ctypes.c_double(0)
        lib.tb_prof_read(k, ctypes.byref(n), ctypes.byref(ms), ctypes.byref(by))
        if off:
            lib.tb_prof_enable(k, 0)
        return dict(launches=n.value, total_ms=ms.value, algo_bytes=by.value)

    # Warm-up with every family's HIP-event timer on: it also decides which LSTM family is the dominant one.
    # The timed region then carries only THAT family's events (an event pair per launch of all three
    # families costs ~0.35 ms per step); the per-family table comes from a few extra, untimed steps.
    for k in kids.values():
        lib.tb_prof_enable(k, 1)
    for i in range(args.warmup):
        step(i)
    barrier()
    ops.check_errors(device)
    warm = {name: read(k) for name, k in kids.items()}
    dom = max(("lstm_fwd", "lstm_bwd"), key=lambda k: warm[k]["total_ms"]) if args.warmup > 0 else "lstm_fwd"
    lib.tb_prof_enable(kids[dom], 1)
    from timeband_amd import dp as _dpm
    _dpm.comm_timing(world > 1)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dt_rank = time.perf_counter() - t0           # this rank alone, before the barrier
    barrier()
    dt = time.perf_counter() - t0
    ops.check_errors(device)
    comm_wait_ms = _dpm.comm_wait_ms()
    _dpm.comm_timing(False)
    p = read(kids[dom])
    extra = min(5, args.steps)
    for k in kids.values():
        lib.tb_prof_enable(k, 1)
    for i in range(extra):
        step(args.warmup + args.steps + i)
    barrier()
    ops.check_errors(device)
    prof = {name: read(k) for name, k in kids.items()}
    # the dominant family alone on one stream (TB_OVERLAP=0): its launch time without a co-running kernel
    mask = trainer.overlap_mask
    trainer.overlap_mask = mask & 48      # one stream; the twin forward and the joint backward (not stream matters) stay as they are
    step(args.warmup + args.steps + extra)
    barrier()
    lib.tb_prof_enable(kids[dom], 1)
    for i in range(extra):
        step(args.warmup + args.steps + extra + 1 + i)
    barrier()
    ops.check_errors(device)
    solo = read(kids[dom])
    trainer.overlap_mask = mask
    per_rank = None
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # per-rank spread (host jitter becomes all-reduce skew) and exposed communication: the time a stream that
        # needed reduced gradients had to wait for them
        mine = torch.tensor([dt_rank / args.steps * 1e3, comm_wait_ms / args.steps], dtype=torch.float64, device=device)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        ms = [float(a[0]) for a in allr]
        cw = [float(a[1]) for a in allr]
        per_rank = {"ms_per_step_min": round(min(ms), 4), "ms_per_step_max": round(max(ms), 4),
                    "ms_per_step": [round(v, 4) for v in ms],
                    "exposed_comm_ms_per_step_max": round(max(cw), 4), "exposed_comm_ms_per_step": [round(v, 4) for v in cw],
                    "note": "per rank, own clock up to its own device sync (before the closing barrier); exposed = time "
                            "the optimizer steps waited for the bucketed gradient all-reduces (events on their stream)"}

    if rank == 0:
        windows = B * world * args.steps
        value = windows / dt
        achieved = p["algo_bytes"] / (p["total_ms"] * 1e-3) / 1e9 if p["total_ms"] > 0 else 0.0
        abw = algo_bytes_per_window(**cfg)
        traffic, traffic_src = pmc_traffic(f"{dom}_kernel") if args.config == "cfg2" else (None, None)
        out = {
            "metric": "G+D training windows/sec", "value": round(value, 2), "unit": "windows/s",
            "n_gpus": dist.get_world_size() if world > 1 else 1,
            "rccl_ranks": dist.get_world_size() if world > 1 else 0, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.config}: LSTM-GAN G+D iteration (critic=1 + D phase + G phase), "
                                   f"V={cfg['V']} E={cfg['V'] + cfg['NT']} window={cfg['To']} hidden={cfg['H']} "
                                   f"layers={cfg['L']} batch={B}/GPU", "global_batch": B * world,
                       "parallelism": f"dp{world}", "full_iteration": bool(args.full),
                       **({"shared_generator_pass": True} if args.share_gpass else {})},
            "roofline": {"bound": "hbm", "kernel": f"{dom}_kernel (persistent recurrent, all LSTM layers)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "traffic_source": traffic_src,
                         "launches": p["launches"],
                         "avg_launch_ms": round(p["total_ms"] / max(1, p["launches"]), 4),
                         "algo_bytes_per_launch": round(p["algo_bytes"] / max(1, p["launches"]), 1),
                         "single_stream": {"avg_launch_ms": round(solo["total_ms"] / max(1, solo["launches"]), 4),
                                           "achieved": round(solo["algo_bytes"] / max(solo["total_ms"], 1e-9) / 1e6, 2),
                                           "frac": round(solo["algo_bytes"] / max(solo["total_ms"], 1e-9) / 1e6
                                                         / HBM_PEAK_GBS, 5),
                                           "note": f"{extra} untimed steps on one stream (TB_OVERLAP & 48): no co-running kernel"}},
            "step_hbm_roofline": {"algo_bytes_per_window": abw,
                                  "frac": round(abw * value / world / (HBM_PEAK_GBS * 1e9), 5)},
            "roofline_gemm": pmc_mfma() if args.config == "cfg2" else None,
            "kernel_ms_per_step": {k: round(v["total_ms"] / max(1, extra), 4) for k, v in prof.items()},
            "kernel_launches_per_step": {k: round(v["launches"] / max(1, extra), 2) for k, v in prof.items()},
            "algo_bytes_per_launch": {k: round(v["algo_bytes"] / max(1, v["launches"]), 1) for k, v in prof.items()
                                      if v["algo_bytes"] > 0},
            "kernel_ms_per_step_source": f"{extra} extra untimed steps with every family's event timers on",
        }
        if per_rank is not None:
            out["ranks"] = per_rank
        if world == 1 and not args.no_cpu_baseline:
            from oracle.torch_cpu_port import time_cpu_baseline
            threads = min(16, os.cpu_count() or 1)
            iters = 3 if args.config == "cfg2" else 10
            wps, spi, nthr = time_cpu_baseline(cfg["V"], cfg["V"] + cfg["NT"], cfg["To"], cfg["Tf"], cfg["H"], cfg["L"],
                                               B, iters=iters, warmup=1, threads=threads)
            out["cpu_baseline"] = {"value": round(wps, 2), "unit": "windows/s", "cores": nthr, "kind": "port",
                                   "sample": f"{iters} timed + 1 warm-up iterations of the same workload "
                                             f"(batch {B}, critic batch {cfg['Tf']}), stock PyTorch CPU ops, "
                                             f"{spi:.2f} s/iteration"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
