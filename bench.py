#!/usr/bin/env python3
"""Throughput of the batched LDPC decoder on MI355X: decoded info Gbit/s.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one decode launch over a batch of synthetic frames resident in HBM.  The default
workload is the CCSDS (512,256) code, 2^20 frames per launch and per GPU, the reference's own
ldpc.py decoder (rule R0, at most 10 iterations with a syndrome check at the top of each), float64
arithmetic (bit-exact with the reference), at an SNR low enough that practically every frame uses
the whole iteration budget — so "10 BP iterations" are really executed.  Frames are independent:
every rank decodes its own batch (weak scaling) and the ranks meet only in the all-reduce of the
error counters.

Rank 0 prints one JSON line.  Beside the headline value it carries
  roofline      algorithmic HBM bytes per launch / measured kernel time, against 8 TB/s
  lds_roofline  algorithmic LDS bytes of the executed iterations against the nominal LDS rate
  valu_roofline vector instructions issued (counter pass on file) against the nominal issue rate
  cpu_baseline  the CPU restatement (oracle/bp_oracle.c, OpenMP) timed on a bounded sample of the
                same LLRs on this host, which doubles as a parity check of that sample
  other         secondary workloads (early exit at an operating SNR, float32, rules R1 and R3, BG2-shaped 5G matrix, NR BG1
                Z=384 layered min-sum), aggregated over all ranks like the headline
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E peak
LDS_PEAK_GBS = 128 * 256 * 2.4   # 128 B/clk/CU x 256 CUs x 2.4 GHz = 78.6 TB/s (nominal)


def lds_bytes_per_frame_iter(code, rule, itemsize):
    """Algorithmic LDS traffic of one iteration of one frame: flooding extrinsic rules read and
    write one message per edge in both phases (16 E), R0 12 E + 4 n, layered min-sum reads and
    writes one posterior per edge (8 E) plus its compressed check state (24 m)."""
    E, n = code.num_edges, code.n
    per4 = (12 * E + 4 * n) if rule == 0 else (8 * E + 24 * code.m if rule == 5 else 16 * E)
    return per4 * itemsize // 4


def measured_traffic(args, frames):
    """HBM bytes per launch from the hardware counters (FETCH_SIZE / WRITE_SIZE, separate rocprofv3
    passes, corrected as profiles/*_hbm_traffic.json says), scaled to this launch's frame count;
    None when no counter run is on file for this workload."""
    key = f"{args.code}/{args.rule}/{args.dtype}/f32llr"
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            rec = json.load(open(path))["workloads"].get(key)
        except (OSError, ValueError, KeyError):
            rec = None
        if rec and rec.get("early_exit", True) == (not args.fixed):
            best = rec["hbm_bytes_per_frame"] * frames
    return best


def valu_roofline(args, executed_frame_iters, kernel_ms):
    """Vector-issue bound: wave-instructions per second against one instruction per SIMD every 4
    cycles (256 CUs x 4 SIMDs x 2.4 GHz / 4).  The instruction count per frame-iteration comes from
    the SQ_INSTS_VALU pass on file for this workload (profiles/*_sq_counters.json)."""
    key = f"{args.code}/{args.rule}/{args.dtype}/f32llr"
    per = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_sq_counters.json"))):
        try:
            rec = json.load(open(path))["workloads"].get(key)
        except (OSError, ValueError, KeyError):
            rec = None
        if rec and rec.get("early_exit", True) == (not args.fixed):
            per = rec["valu_wave_instructions_per_frame_iteration"]
    if per is None:
        return None
    peak = 256 * 4 * 2.4e9 / 4 / 1e9
    achieved = per * executed_frame_iters / (kernel_ms * 1e-3) / 1e9
    return {"bound": "valu", "achieved": achieved, "peak": peak, "unit": "G wave-instructions/s",
            "frac": achieved / peak, "valu_per_frame_iteration": per}


def timed_steps(fn, steps, warmup, torch, dist, world):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64,
                         device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])
    return wall, dev_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="frames per launch per GPU (0: 2^20 for the "
                    "small codes, as many as give about 3 GiB of LLRs for the large ones)")
    ap.add_argument("--code", default="ccsds512")
    ap.add_argument("--rule", default="R0")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--snr", type=float, default=-3.0, help="Es/N0 in dB of the synthetic channel")
    ap.add_argument("--max-iters", type=int, default=10)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--fixed", action="store_true", help="no early exit: every frame runs max-iters updates")
    ap.add_argument("--cpu-sample", type=int, default=0, help="frames of the CPU baseline (0 = auto)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-other", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the decoder has no CPU path")
    # rehearsal on a one-GPU box: LDPC_BENCH_BACKEND=gloo puts every rank on device 0 and runs the
    # collectives through host memory; the real multi-GPU run uses RCCL, one rank per GPU
    backend = os.environ.get("LDPC_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local = 0
    torch.cuda.set_device(local)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local)

    from ldpc_amd import codes, config, sweep
    from ldpc_amd.decoder import Decoder

    code = codes.load_packaged(args.code)
    G = codes.load_packaged_generator(args.code)   # a k x n matrix, or a structured encoder for BG1
    k_info = G.k if callable(G) else G.shape[0]
    if args.frames <= 0:
        args.frames = 1 << 20 if code.n <= 1024 else max(1024, (3 << 30) // (4 * code.n) // 1024 * 1024)
    cfg = config.PRESETS[args.rule].with_(max_iters=args.max_iters)
    if args.fixed:
        cfg = cfg.with_(early_exit=False)
    compute = np.float64 if args.dtype == "f64" else np.float32
    dec = Decoder(code)
    batch = sweep.make_batch(code, G, args.frames, args.snr, cfg, seed=[7, rank], device=device)
    kernel_name = [""]

    def step():
        kernel_name[0] = sweep.decode_batch(dec, batch, cfg, compute_dtype=compute, kernel=args.kernel)

    wall, dev_ms = timed_steps(step, args.steps, args.warmup, torch, dist, world)
    frames_total = args.frames * world * args.steps
    value = frames_total * k_info / wall / 1e9

    # error counters of the last step, summed over ranks: the one collective of the path
    batch.counters.zero_()
    sweep.count_errors(dec, batch)
    sweep.allreduce_counters(batch.counters)
    torch.cuda.synchronize()
    stats = sweep.summarize(batch.counters.cpu().numpy(), code.n)

    kernel_ms = dev_ms / args.steps
    hbm_bytes = (5 * code.n + 5) * args.frames                  # f32 LLR in, u8 hard + i32 + u8 out
    # iterations actually run: failures ran max_iters but report max-1; fixed mode reports max_iters
    executed = stats["iterations"] + (0 if args.fixed else stats["not_converged"])
    lds_bytes = lds_bytes_per_frame_iter(code, cfg.rule, 8 if args.dtype == "f64" else 4) * executed / world
    roofline = {"bound": "hbm", "achieved": hbm_bytes / (kernel_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "traffic": measured_traffic(args, args.frames), "kernel": kernel_name[0],
                "kernel_ms": kernel_ms, "bytes_per_frame": 5 * code.n + 5}
    roofline["frac"] = roofline["achieved"] / roofline["peak"]
    lds = {"bound": "lds", "achieved": lds_bytes / (kernel_ms * 1e-3) / 1e9, "peak": LDS_PEAK_GBS,
           "unit": "GB/s", "bytes_per_frame_iter": lds_bytes_per_frame_iter(code, cfg.rule, 8 if args.dtype == "f64" else 4)}
    lds["frac"] = lds["achieved"] / lds["peak"]

    out = {
        "metric": "decoded info Gbit/s", "value": value, "unit": "Gbit/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": wall / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"{args.code} (n={code.n}, k={k_info}) x {args.frames} frames/launch/GPU, "
                               f"rule {args.rule} ({ {'R0': 'ldpc.py decoder', 'R1': 'bpsk.py decoder', 'LNMS': 'layered min-sum, build extension'}.get(args.rule, 'sum-product') }), "
                               f"max {args.max_iters} iterations, {'no early exit' if args.fixed else 'syndrome check per iteration'}, Es/N0 {args.snr} dB",
                   "code": args.code, "frames_per_launch_per_gpu": args.frames, "rule": args.rule,
                   "max_iters": args.max_iters, "early_exit": not args.fixed, "es_n0_db": args.snr,
                   "llr_dtype": "f32", "kernel": kernel_name[0], "sharding": f"frames x{world}"},
        "mean_iters": stats["mean_iters"], "fer": stats["fer"], "ber": stats["ber"],
        "counters": {k: stats[k] for k in sweep.COUNTER_NAMES},
        "roofline": roofline, "lds_roofline": lds,
    }
    vr = valu_roofline(args, executed / world, kernel_ms)
    if vr:
        out["valu_roofline"] = vr

    if rank == 0 and world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(code, cfg, batch, k_info, args)
    if not args.no_other and args.code == "ccsds512":
        out["other"] = other_workloads(code, G, dec, args, torch, dist, device, world, rank)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def cpu_baseline(code, cfg, batch, k_info, args):
    """CPU restatement on a bounded sample of the same LLRs, all host threads; also checks that
    the GPU's results for that sample are identical."""
    from oracle import bp_oracle
    bp_oracle.build_c_oracle()
    threads = min(bp_oracle.c_max_threads(), len(os.sched_getaffinity(0)), 64)
    if cfg.rule == 5:
        decode = lambda x: bp_oracle.c_decode_lnms(code, cfg, x, nthreads=threads, use_float=args.dtype == "f32",
                                                   want_posterior=False)
    else:
        decode = lambda x: bp_oracle.c_decode(code, cfg, x, nthreads=threads, want_posterior=False)
    probe = (64 if code.n <= 1024 else 2) * threads
    L = batch.llr[:probe].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    decode(L)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    sample = args.cpu_sample or int(min(batch.llr.shape[0], max(probe, rate * 15.0)))
    L = batch.llr[:sample].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    hard, iters, conv, _ = decode(L)
    dt = time.perf_counter() - t0
    same = (np.array_equal(hard, batch.hard[:sample].cpu().numpy())
            and np.array_equal(iters, batch.iters[:sample].cpu().numpy())
            and np.array_equal(conv, batch.converged[:sample].cpu().numpy().astype(bool)))
    return {"value": sample * k_info / dt / 1e9, "unit": "Gbit/s", "cores": threads, "kind": "port",
            "sample": f"first {sample} frames of the same LLR batch, "
                      f"{'float32' if (cfg.rule == 5 and args.dtype == 'f32') else 'float64'}, oracle/bp_oracle.c with "
                      f"{threads} OpenMP threads, {dt:.1f} s",
            "gpu_results_identical_on_sample": bool(same) if (args.dtype == "f64" or cfg.rule == 5) else None}


def other_workloads(code, G, dec, args, torch, dist, device, world, rank):
    """Secondary workloads, 3 timed launches each on every rank; Gbit/s over all ranks, `frames`
    per launch and per GPU."""
    from ldpc_amd import codes, config, sweep
    from ldpc_amd.decoder import Decoder
    res = {}

    def run(label, code_, G_, dec_, rule, dtype, snr, early, frames, k_info):
        cfg = config.PRESETS[rule].with_(max_iters=args.max_iters, early_exit=early)
        batch = sweep.make_batch(code_, G_, frames, snr, cfg, seed=[8, len(label), rank], device=device)
        compute = np.float64 if dtype == "f64" else np.float32
        wall, dev_ms = timed_steps(lambda: sweep.decode_batch(dec_, batch, cfg, compute_dtype=compute),
                                   3, 1, torch, dist, world)
        sweep.count_errors(dec_, batch)
        sweep.allreduce_counters(batch.counters)
        torch.cuda.synchronize()
        st = sweep.summarize(batch.counters.cpu().numpy(), code_.n)
        res[label] = {"gbit_s": frames * world * 3 * k_info / wall / 1e9, "mean_iters": st["mean_iters"],
                      "fer": st["fer"], "ber": st["ber"], "frames": frames, "code": code_.name, "rule": rule,
                      "dtype": dtype, "es_n0_db": snr, "early_exit": early}

    frames = min(args.frames, 1 << 19)
    k = G.shape[0]
    # the step in front of the decoder: BPSK + AWGN + LLRs generated on the device (n bytes in, 4n out)
    from ldpc_amd import channel
    cfg0 = config.PRESETS["R0"]
    probe = sweep.make_batch(code, G, frames, args.snr, cfg0, seed=[9, rank], device=device, native_noise=True)
    gen = lambda: channel.channel_llr_device(probe.ref.data_ptr(), code.n, frames, code.n, args.snr, cfg0, 7, 0,
                                             probe.llr.data_ptr(), np.float32,
                                             stream=torch.cuda.current_stream().cuda_stream)
    wall, _ = timed_steps(gen, 5, 1, torch, dist, world)
    res["channel_llr_f32"] = {"gb_s": 5 * code.n * frames * 5 * world / wall / 1e9, "frames": frames, "code": code.name,
                              "ms_per_launch": wall / 5 * 1e3, "bytes_per_frame": 5 * code.n}
    del probe
    for label, rule, dtype, snr, early in [
        ("R0_f64_es_n0_1dB", "R0", "f64", 1.0, True),
        ("R0_f32_worst_case", "R0", "f32", args.snr, True),
        ("R1_f64_worst_case", "R1", "f64", args.snr, True),
        ("R1_f64_es_n0_1dB", "R1", "f64", 1.0, True),
        ("R1_f32_worst_case", "R1", "f32", args.snr, True),
        ("R0_f64_fixed_10", "R0", "f64", args.snr, False),
    ]:
        run(label, code, G, dec, rule, dtype, snr, early, frames, k)
    # the build-authored NR BG1 Z=384 code under row-layered min-sum (float; see ldpc_amd/nr_bg1.py)
    bg1 = codes.load_packaged("nr_bg1_z384")
    enc1 = codes.load_packaged_generator("nr_bg1_z384")
    dec1 = Decoder(bg1)
    run("BG1_Z384_LNMS_f32_fixed_10", bg1, enc1, dec1, "LNMS", "f32", -3.0, False, 16384, bg1.n - bg1.m)
    run("BG1_Z384_LNMS_f32_es_n0_-3dB", bg1, enc1, dec1, "LNMS", "f32", -3.0, True, 16384, bg1.n - bg1.m)
    run("BG1_Z384_LNMS_f32_es_n0_0dB", bg1, enc1, dec1, "LNMS", "f32", 0.0, True, 16384, bg1.n - bg1.m)
    # sum-product (tanh rule of channel.py) on the headline code, and the flooding rules on the BG2-shaped
    # 5G matrix of the reference (all-zero codeword: these rules are symmetric in the codeword)
    run("R3_f32_fixed_10", code, G, dec, "R3", "f32", args.snr, False, frames // 4, k)
    run("R3_f64_fixed_10", code, G, dec, "R3", "f64", args.snr, False, frames // 8, k)
    bg2 = codes.load_packaged("nr_bg2_z40")
    dec2 = Decoder(bg2)
    k2 = bg2.n - bg2.m
    for label, rule, dtype in [("BG2_Z40_R0_f64_fixed_10", "R0", "f64"), ("BG2_Z40_R0_f32_fixed_10", "R0", "f32"),
                               ("BG2_Z40_R1_f64_fixed_10", "R1", "f64"), ("BG2_Z40_R1_f32_fixed_10", "R1", "f32"),
                               ("BG2_Z40_R3_f32_fixed_10", "R3", "f32")]:
        run(label, bg2, None, dec2, rule, dtype, -2.0, False, 32768, k2)
    run("BG2_Z40_R1_f32_es_n0_-4dB", bg2, None, dec2, "R1", "f32", -4.0, True, 32768, k2)
    return res


if __name__ == "__main__":
    main()
