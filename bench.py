#!/usr/bin/env python3
"""Throughput of the batched LDPC decoder on MI355X: decoded info Gbit/s at 10 BP iterations.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...         (starts its N ranks itself, one per GPU, through torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one decode launch over a batch of synthetic frames resident in HBM.  Frames are
independent: every rank decodes its own batch (weak scaling) and the ranks meet only in the
all-reduce of the error counters.

Headline (`value`): CCSDS (512,256), 2^20 frames per launch and per GPU, the reference's own ldpc.py
decoder (rule R0), exactly 10 iterations per frame (no early exit), float64 arithmetic (bit-exact
with the reference).  The same kernel with the reference's per-iteration syndrome check at an
operating SNR is reported beside it as `operating_point`, and with that check at an SNR where no
frame converges as `worst_case_early_exit`.

Rank 0 prints one JSON line.  Beside the headline it carries
  roofline      algorithmic HBM bytes per launch / measured kernel time, against 8 TB/s
  lds_roofline  measured: share of the time the LDS arrays were busy (counter pass on file, only while the
                kernel source is unchanged since); beside it the LDS bytes the formulation moves per
                executed iteration against the LDS rates of the instruction widths involved
  valu_roofline vector instructions issued (counter pass on file, only while the kernel source is
                unchanged since) against the nominal issue rate
  cpu_baseline  the CPU restatement (oracle/bp_oracle.c, OpenMP) timed on a bounded sample of the
                same LLRs on this host, which doubles as a parity check of that sample
  bg1           the other half of the metric: NR BG1 Z=384 (build-authored table) under row-layered
                min-sum, exactly 10 sweeps, in float32 and float64, each with its own rooflines and
                CPU baseline, plus early-exit operating points
  other         secondary workloads, aggregated over all ranks like the headline
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
# LDS rates with every CU streaming (GB/s): 4-byte reads, 8/16-byte reads, writes (38-51 TB/s measured
# for ds_write_b32...b128; the midpoint is used)
LDS_READ32_GBS, LDS_READ64_GBS, LDS_WRITE_GBS = 75e3, 150e3, 45e3
VALU_PEAK_GIPS = 256 * 4 * 2.4 / 4   # one wave-instruction per SIMD every 4 cycles at 2.4 GHz


def lds_bytes_per_frame_iter(code, rule, itemsize):
    """Algorithmic LDS traffic of one iteration of one frame as (bytes read, bytes written), for
    4-byte numbers scaled by the item size: flooding extrinsic rules read and write one message per
    edge in both phases (8 E + 8 E); R0 reads 8 E and writes 4 E + 4 n; layered min-sum reads and
    writes one posterior per edge and its compressed check state (4 E + 12 m each way)."""
    E, n, m = code.num_edges, code.n, code.m
    if rule == 0:
        r, w = 8 * E, 4 * E + 4 * n
    elif rule == 5:
        r, w = 4 * E + 12 * m, 4 * E + 12 * m
    else:
        r, w = 8 * E, 8 * E
    return r * itemsize // 4, w * itemsize // 4


def _profile_record(kind, key, fixed, kernel):
    """Newest counter record on file for this workload whose kernel source is unchanged since."""
    from ldpc_amd.build import kernel_source_digest
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", f"*_{kind}.json"))):
        try:
            rec = json.load(open(path))["workloads"].get(key)
        except (OSError, ValueError, KeyError):
            rec = None
        if rec and rec.get("early_exit", True) == (not fixed) and rec.get("kernel") == kernel \
                and rec.get("kernel_source_sha16") == kernel_source_digest(kernel):
            best = dict(rec, source=os.path.basename(path))
    return best


def rooflines(code, cfg, dtype, frames, kernel_ms, executed, kernel, code_name, rule_name, fused=False, packed=False):
    """HBM, LDS and vector-issue rooflines of one launch."""
    itemsize = {"f64": 8, "f32": 4, "f16": 2}[dtype]
    key = f"{code_name}/{rule_name}/{dtype}/" + (("fused_packed" if packed else "fused") if fused else "f32llr") + ("/graph" if kernel.startswith("bg_lnms") else "")
    fixed = not cfg.early_exit
    sec = kernel_ms * 1e-3
    # decode: f32 LLR in, u8 hard + i32 + u8 out; fused sweep: the transmitted codeword (u8) in, nothing out
    per_frame = ((code.n + 7) // 8 if packed else code.n) if fused else 5 * code.n + 5
    hbm_bytes = per_frame * frames
    tr = _profile_record("hbm_traffic", key, fixed, kernel)
    roof = {"bound": "hbm", "achieved": hbm_bytes / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "traffic": tr["hbm_bytes_per_frame"] * frames if tr else None,
            "traffic_source": tr["source"] if tr else None,
            "kernel": kernel, "kernel_ms": kernel_ms, "bytes_per_frame": per_frame}
    roof["frac"] = roof["achieved"] / roof["peak"]
    rd, wr = lds_bytes_per_frame_iter(code, cfg.rule, itemsize)
    qc = getattr(code, "qc", None)
    kept = 0.0
    if qc is not None and qc.mb <= 4 and kernel.startswith(("spec_r0_", "spec_r1_", "spec_oms_", "fused_spec_")):
        # the single-wave kernels keep edges of shift 0 inside the lane (registers): that share of the
        # algorithmic edge traffic never reaches LDS and is not counted as moved
        kept = float((qc.edge_shift == 0).mean())
    rd, wr = rd * (1.0 - kept), wr * (1.0 - kept)
    read_rate = LDS_READ64_GBS if itemsize == 8 else LDS_READ32_GBS
    floor = executed * (rd / (read_rate * 1e9) + wr / (LDS_WRITE_GBS * 1e9))
    lds = {"bound": "lds", "achieved": (rd + wr) * executed / sec / 1e9, "unit": "GB/s",
           "read_bytes_per_frame_iter": rd, "write_bytes_per_frame_iter": wr, "edge_share_kept_in_registers": kept,
           "rates_gb_s": {"read": read_rate, "write": LDS_WRITE_GBS}}
    lds["peak"] = (rd + wr) * executed / floor / 1e9      # rate at which this read/write mix could move at best
    lds["frac_by_formula"] = lds["achieved"] / lds["peak"]
    # measured: LDS-array cycles (SQ_LDS_IDX_ACTIVE, bank conflicts included) over the cycles the compute
    # units were busy, from the counter pass on file for this kernel source
    lc = _profile_record("lds_counters", key, fixed, kernel)
    if lc:
        lds.update({"measured_busy_frac": lc["lds_array_busy_frac"],
                    "lds_array_cycles_per_frame_iteration": lc["lds_array_cycles_per_frame_iteration"],
                    "lds_instructions_per_frame_iteration": lc["lds_instructions_per_frame_iteration"],
                    "bank_conflict_share_of_lds_cycles": lc["SQ_LDS_BANK_CONFLICT"] / max(lc["SQ_LDS_IDX_ACTIVE"], 1.0),
                    "source": lc["source"]})
    lds["frac"] = lds.get("measured_busy_frac", lds["frac_by_formula"])
    out = {"roofline": roof, "lds_roofline": lds}
    sq = _profile_record("sq_counters", key, fixed, kernel)
    if sq:
        per = sq["valu_wave_instructions_per_frame_iteration"]
        ach = per * executed / sec / 1e9
        out["valu_roofline"] = {"bound": "valu", "achieved": ach, "peak": VALU_PEAK_GIPS, "unit": "G wave-instructions/s",
                                "frac": ach / VALU_PEAK_GIPS, "valu_per_frame_iteration": per, "source": sq["source"]}
    return out


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


class Runner:
    """One (code, decoder) pair and the measurement of a workload on it."""

    def __init__(self, name, torch, dist, device, world, rank):
        from ldpc_amd import codes
        from ldpc_amd.decoder import Decoder
        self.name, self.torch, self.dist, self.device, self.world, self.rank = name, torch, dist, device, world, rank
        self.code = codes.load_packaged(name)
        self.G = codes.load_packaged_generator(name) if name in ("ccsds512", "wifi648", "nr_bg1_z384") else None
        self.k = (self.G.k if callable(self.G) else self.G.shape[0]) if self.G is not None else self.code.n - self.code.m
        self.dec = Decoder(self.code)

    def run(self, rule, dtype, snr, early, frames, steps, warmup, max_iters=10, seed=7, kernel=0, with_rooflines=False,
            fused=False, packed=False):
        """`fused`: a step is channel + decode + error accounting in one launch (ldpc_sweep_batch) on
        codewords resident in HBM, instead of the decode of LLRs resident in HBM."""
        from ldpc_amd import config, sweep
        cfg = config.PRESETS[rule].with_(max_iters=max_iters, early_exit=early)
        batch = sweep.make_batch(self.code, self.G, frames, snr, cfg, seed=[seed, self.rank], device=self.device)
        compute = {"f64": np.float64, "f32": np.float32, "f16": np.float16}[dtype]
        kname = [""]
        ref_words = sweep.pack_bits(batch.ref) if fused and packed else batch.ref   # transmitted words as bits: n / 8 bytes per frame

        def step():
            if fused:
                kname[0] = sweep.sweep_fused(self.dec, ref_words, snr, cfg, [seed, self.rank], batch.counters, compute_dtype=compute,
                                             packed_n=self.code.n if packed else None)
            else:
                kname[0] = sweep.decode_batch(self.dec, batch, cfg, compute_dtype=compute, kernel=kernel)

        wall, dev_ms = timed_steps(step, steps, warmup, self.torch, self.dist, self.world)
        # error counters of the last step, summed over ranks: the one collective of the path
        batch.counters.zero_()
        if fused:
            step()
        else:
            sweep.count_errors(self.dec, batch)
        sweep.allreduce_counters(batch.counters)
        self.torch.cuda.synchronize()
        st = sweep.summarize(batch.counters.cpu().numpy(), self.code.n)
        res = {"value": frames * self.world * steps * self.k / wall / 1e9, "unit": "Gbit/s", "ms_per_step": wall / steps * 1e3,
               "kernel": kname[0], "kernel_ms": dev_ms / steps, "code": self.name, "rule": rule, "dtype": dtype,
               "frames_per_launch_per_gpu": frames, "max_iters": max_iters, "early_exit": early, "es_n0_db": snr,
               "mean_iters": st["mean_iters"], "fer": st["fer"], "ber": st["ber"],
               "counters": {k: st[k] for k in sweep.COUNTER_NAMES}}
        if with_rooflines:
            # iterations actually run: failures ran max_iters but report max-1; fixed mode reports max_iters
            executed = (st["iterations"] + (st["not_converged"] if early else 0)) / self.world
            res.update(rooflines(self.code, cfg, dtype, frames, dev_ms / steps, executed, kname[0], self.name, rule, fused, packed))
        return res, batch, cfg


def cpu_baseline(code, cfg, batch, k_info, dtype, seconds, cpu_sample=0):
    """CPU restatement on a bounded sample of the same LLRs, all host threads of this process's
    share; also checks that the GPU's results for that sample are identical."""
    from oracle import bp_oracle
    bp_oracle.build_c_oracle()
    threads = min(bp_oracle.c_max_threads(), len(os.sched_getaffinity(0)), 64)
    if cfg.rule == 5:
        decode = lambda x: bp_oracle.c_decode_lnms(code, cfg, x, nthreads=threads, use_float=dtype == "f32",
                                                   want_posterior=False)
    else:
        decode = lambda x: bp_oracle.c_decode(code, cfg, x, nthreads=threads, want_posterior=False)
    probe = (64 if code.n <= 1024 else 2) * threads
    L = batch.llr[:probe].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    decode(L)
    rate = probe / max(time.perf_counter() - t0, 1e-6)
    sample = cpu_sample or int(min(batch.llr.shape[0], max(probe, rate * seconds)))
    L = batch.llr[:sample].cpu().numpy().astype(np.float64)
    t0 = time.perf_counter()
    hard, iters, conv, _ = decode(L)
    dt = time.perf_counter() - t0
    same = (np.array_equal(hard, batch.hard[:sample].cpu().numpy())
            and np.array_equal(iters, batch.iters[:sample].cpu().numpy())
            and np.array_equal(conv, batch.converged[:sample].cpu().numpy().astype(bool)))
    exact = dtype != "f16" and (dtype == "f64" or cfg.rule == 5)
    return {"value": sample * k_info / dt / 1e9, "unit": "Gbit/s", "cores": threads, "kind": "port",
            "sample": f"first {sample} frames of the same LLR batch, "
                      f"{'float32' if (cfg.rule == 5 and dtype == 'f32') else 'float64'}, oracle/bp_oracle.c with "
                      f"{threads} OpenMP threads, {dt:.1f} s",
            "gpu_results_identical_on_sample": bool(same) if exact else None}


def launch_plan(gpus, env, argv, script=None, device_count=None, port=None):
    """What `bench.py --gpus N` does with the process it was started as.

    ("run", world, rank, local): this process is one rank (WORLD_SIZE names the job, set by
    torch.distributed.run) or the whole one-GPU job.  ("spawn", cmd): started plainly with
    --gpus N > 1 — the ranks do not exist yet, so this process (which must not have touched the GPU)
    starts `python -m torch.distributed.run --nproc-per-node N bench.py <same arguments>` and relays
    its exit code; rank 0 of that job prints the JSON line.  A job whose size differs from --gpus,
    or (under RCCL, one rank per GPU) exceeds the GPUs of the node, is an error rather than a
    silently smaller run."""
    if gpus < 1:
        raise SystemExit(f"bench.py: --gpus {gpus}: need at least one GPU")
    backend = env.get("LDPC_BENCH_BACKEND", "nccl")
    if backend == "nccl" and device_count is not None and gpus > device_count:
        raise SystemExit(f"bench.py: --gpus {gpus} but this node shows {device_count} GPU(s); one rank per GPU under RCCL "
                         f"(LDPC_BENCH_BACKEND=gloo rehearses the launch with every rank on GPU 0)")
    if "WORLD_SIZE" in env:
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            raise SystemExit(f"bench.py: --gpus {gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {gpus} "
                             f"(or run `python bench.py --gpus {gpus}` plainly, which starts the ranks itself)")
        return ("run", world, int(env.get("RANK", "0")), int(env.get("LOCAL_RANK", "0")))
    if gpus == 1:
        return ("run", 1, 0, 0)
    if port is None:
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), script or os.path.abspath(__file__)] + list(argv)
    return ("spawn", cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=0, help="frames per launch per GPU (0: 2^20 for the "
                    "small codes, 16384 for the NR graphs)")
    ap.add_argument("--code", default="ccsds512")
    ap.add_argument("--rule", default="R0")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "f16"], help="arithmetic type; f16 = packed half precision, two frames per lane (reduced-precision fast mode of R1 and LNMS)")
    ap.add_argument("--snr", type=float, default=1.0, help="Es/N0 in dB of the synthetic channel")
    ap.add_argument("--max-iters", type=int, default=10)
    ap.add_argument("--kernel", type=int, default=0)
    ap.add_argument("--early-exit", action="store_true", help="the reference's syndrome check at the top of every "
                    "iteration instead of exactly max-iters updates per frame")
    ap.add_argument("--fixed", action="store_true", help="(default) exactly max-iters updates per frame")
    ap.add_argument("--cpu-sample", type=int, default=0, help="frames of the CPU baseline (0 = auto)")
    ap.add_argument("--packed-ref", action="store_true", help="with --fused: the transmitted codewords as bits (n / 8 bytes per frame)")
    ap.add_argument("--fused", action="store_true", help="a step is channel + decode + error accounting in one launch "
                    "(ldpc_sweep_batch) instead of the decode alone")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-other", action="store_true", help="headline workload only (no bg1, no other)")
    args = ap.parse_args()

    import torch   # importing torch and counting devices does not initialise HIP
    if torch.cuda.device_count() == 0:
        raise SystemExit("bench.py needs an MI355X: the decoder has no CPU path")
    plan = launch_plan(args.gpus, os.environ, sys.argv[1:], device_count=torch.cuda.device_count())
    if plan[0] == "spawn":
        import subprocess
        raise SystemExit(subprocess.run(plan[1]).returncode)
    import torch.distributed as dist

    _, world, rank, local = plan
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

    head = Runner(args.code, torch, dist, device, world, rank)
    code = head.code
    if args.frames <= 0:
        args.frames = 1 << 20 if code.n <= 1024 else 16384
    early = bool(args.early_exit)
    res, batch, cfg = head.run(args.rule, args.dtype, args.snr, early, args.frames, args.steps, args.warmup,
                               max_iters=args.max_iters, kernel=args.kernel, with_rooflines=True, fused=args.fused, packed=args.packed_ref)
    rule_text = {'R0': 'ldpc.py decoder', 'R1': 'bpsk.py decoder', 'LNMS': 'layered min-sum, build extension'}.get(args.rule, 'sum-product')
    out = {
        "metric": "decoded info Gbit/s", "value": res["value"], "unit": "Gbit/s", "n_gpus": world,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": res["ms_per_step"],
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
        "data": "synthetic",
        "config": {"workload": f"{args.code} (n={code.n}, k={head.k}) x {args.frames} frames/launch/GPU, "
                               f"rule {args.rule} ({rule_text}), "
                               f"{'at most' if early else 'exactly'} {args.max_iters} iterations"
                               f"{' (syndrome check per iteration)' if early else ' per frame (no early exit)'}, Es/N0 {args.snr} dB",
                   "code": args.code, "frames_per_launch_per_gpu": args.frames, "rule": args.rule,
                   "max_iters": args.max_iters, "early_exit": early, "es_n0_db": args.snr,
                   "llr_dtype": "f32", "kernel": res["kernel"], "sharding": f"frames x{world}",
                   "step": "channel + decode + error accounting in one launch" if args.fused else "decode of LLRs resident in HBM"},
        "mean_iters": res["mean_iters"], "fer": res["fer"], "ber": res["ber"], "counters": res["counters"],
        "roofline": res["roofline"], "lds_roofline": res["lds_roofline"],
    }
    if "valu_roofline" in res:
        out["valu_roofline"] = res["valu_roofline"]
    if rank == 0 and world == 1 and not args.no_cpu and not args.fused:
        out["cpu_baseline"] = cpu_baseline(code, cfg, batch, head.k, args.dtype, 10.0, args.cpu_sample)
        out["vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    del batch

    if not args.no_other and args.code == "ccsds512" and args.rule == "R0" and not early and not args.fused:
        short = lambda r: {k: r[k] for k in ("value", "unit", "ms_per_step", "kernel", "mean_iters", "fer", "ber",
                                             "es_n0_db", "early_exit", "frames_per_launch_per_gpu", "dtype")}
        # the same kernel running the reference's loop (syndrome check at the top of every iteration)
        r, b, _ = head.run(args.rule, args.dtype, 1.0, True, args.frames, 3, 1, seed=8)
        out["operating_point"] = short(r)
        del b
        r, b, _ = head.run(args.rule, args.dtype, -3.0, True, args.frames, 3, 1, seed=8)
        out["worst_case_early_exit"] = short(r)
        del b
        out["bg1"] = bg1_workloads(torch, dist, device, world, rank, args)
        out["other"] = other_workloads(head, args, torch, dist, device, world, rank)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


def bg1_workloads(torch, dist, device, world, rank, args):
    """NR BG1 Z=384 (build-authored table, ldpc_amd/nr_bg1.py) under row-layered min-sum: exactly 10
    sweeps per frame in float32 and in float64, each with rooflines and (one GPU) a CPU baseline,
    and the early-exit operating points of the float32 kernel."""
    bg1 = Runner("nr_bg1_z384", torch, dist, device, world, rank)
    frames = 16384
    out = {"workload": f"nr_bg1_z384 (n={bg1.code.n}, k={bg1.k}; BG1 connectivity, synthetic shifts) x {frames} "
                       f"frames/launch/GPU, rule LNMS (row-layered normalised min-sum, build extension), exactly 10 sweeps",
           "metric": "decoded info Gbit/s",
           "headline_dtype": "f32",
           "headline_dtype_reason": "layered min-sum is a build extension: no reference arithmetic to reproduce; float32 and "
                                    "float64 decode 131072 waterfall frames to the same words, 1 frame differing in flag and sweep "
                                    "count (tests/test_gpu_precision.py); double holds one frame per compute unit instead of two"}
    for dtype in ("f32", "f64"):
        r, batch, cfg = bg1.run("LNMS", dtype, 0.0, False, frames, 5, 1, seed=9, with_rooflines=True)
        r.pop("counters")
        if rank == 0 and world == 1 and not args.no_cpu:
            r["cpu_baseline"] = cpu_baseline(bg1.code, cfg, batch, bg1.k, dtype, 5.0)
        out[dtype] = r
        del batch
    # the same rule on the kernel that takes lifting size, shift table and active block rows at run time (what any
    # other 38.212 table, lifting size or rate-matched sub-matrix runs on; bit-identical results)
    r, batch, cfg = bg1.run("LNMS", "f32", 0.0, False, frames, 5, 1, seed=9, kernel=3, with_rooflines=True)
    r.pop("counters")
    out["f32_run_time_shifts"] = r
    del batch
    # reduced-precision fast mode: packed half precision, two frames per lane (validated statistically only)
    r, batch, cfg = bg1.run("LNMS", "f16", 0.0, False, frames, 5, 1, seed=9, with_rooflines=True)
    r.pop("counters")
    r["note"] = "reduced precision: half-precision posteriors, 8-bit check magnitudes; no CPU restatement, error rates checked against float32"
    out["f16x2"] = r
    del batch
    for dt in ("f32", "f16"):
        for label, snr in ((f"{dt}_early_exit_es_n0_-3dB", -3.0), (f"{dt}_early_exit_es_n0_0dB", 0.0)):
            r, batch, _ = bg1.run("LNMS", dt, snr, True, frames, 3, 1, seed=9)
            out[label] = {k: r[k] for k in ("value", "unit", "ms_per_step", "kernel", "mean_iters", "fer", "ber")}
            del batch
    return out


def other_workloads(head, args, torch, dist, device, world, rank):
    """Secondary workloads, 3 timed launches each on every rank; Gbit/s over all ranks, `frames`
    per launch and per GPU."""
    from ldpc_amd import channel, config, sweep
    res = {}

    def run(label, runner, rule, dtype, snr, early, frames):
        r, b, _ = runner.run(rule, dtype, snr, early, frames, 3, 2, max_iters=args.max_iters, seed=8 + len(label))
        res[label] = {"gbit_s": r["value"], "mean_iters": r["mean_iters"], "fer": r["fer"], "ber": r["ber"],
                      "frames": frames, "code": runner.name, "rule": rule, "dtype": dtype, "es_n0_db": snr,
                      "early_exit": early, "kernel": r["kernel"]}
        del b

    code = head.code
    frames = min(args.frames, 1 << 19)
    # the step in front of the decoder: BPSK + AWGN + LLRs generated on the device (n bytes in, 4n out)
    cfg0 = config.PRESETS["R0"]
    probe = sweep.make_batch(code, head.G, frames, args.snr, cfg0, seed=[9, rank], device=device, native_noise=True)
    gen = lambda: channel.channel_llr_device(probe.ref.data_ptr(), code.n, frames, code.n, args.snr, cfg0, 7, 0,
                                             probe.llr.data_ptr(), np.float32,
                                             stream=torch.cuda.current_stream().cuda_stream)
    wall, _ = timed_steps(gen, 5, 1, torch, dist, world)
    res["channel_llr_f32"] = {"gb_s": 5 * code.n * frames * 5 * world / wall / 1e9, "frames": frames, "code": code.name,
                              "ms_per_launch": wall / 5 * 1e3, "bytes_per_frame": 5 * code.n}
    del probe
    for label, rule, dtype, snr, early in [
        ("R0_f32_fixed_10", "R0", "f32", 1.0, False),
        ("R1_f64_fixed_10", "R1", "f64", 1.0, False),
        ("R1_f32_fixed_10", "R1", "f32", 1.0, False),
        ("R1_f64_es_n0_1dB", "R1", "f64", 1.0, True),
        ("R1_f16x2_fixed_10", "R1", "f16", 1.0, False),      # packed half precision, two frames per lane: reduced precision
        ("R1_f16x2_es_n0_1dB", "R1", "f16", 1.0, True),
        ("R3_f32_fixed_10", "R3", "f32", 1.0, False),
    ]:
        run(label, head, rule, dtype, snr, early, frames if rule != "R3" else frames // 4)
    run("R3_f64_fixed_10", head, "R3", "f64", 1.0, False, frames // 8)
    # the small code of Matrices/H.txt (802.11n (648,324)), 65 536 frames, min-sum
    wifi = Runner("wifi648", torch, dist, device, world, rank)
    run("WIFI648_R1_f64_fixed_10", wifi, "R1", "f64", 1.0, False, 65536)
    run("WIFI648_R1_f32_fixed_10", wifi, "R1", "f32", 1.0, False, 65536)
    run("WIFI648_R1_f64_es_n0_1dB", wifi, "R1", "f64", 1.0, True, 65536)
    del wifi
    # the flooding rules on the BG2-shaped 5G matrix of the reference (all-zero codeword: these rules are
    # symmetric in the codeword)
    bg2 = Runner("nr_bg2_z40", torch, dist, device, world, rank)
    for label, rule, dtype in [("BG2_Z40_R0_f64_fixed_10", "R0", "f64"), ("BG2_Z40_R0_f32_fixed_10", "R0", "f32"),
                               ("BG2_Z40_R1_f64_fixed_10", "R1", "f64"), ("BG2_Z40_R1_f32_fixed_10", "R1", "f32"),
                               ("BG2_Z40_R3_f32_fixed_10", "R3", "f32")]:
        run(label, bg2, rule, dtype, -2.0, False, 32768)
    run("BG2_Z40_R1_f32_es_n0_-4dB", bg2, "R1", "f32", -4.0, True, 32768)
    # ... and on its first 1400 rows, the matrix the reference's channel.py decodes (channel.py:35)
    cut = Runner("nr_bg2_z40_r1400", torch, dist, device, world, rank)
    run("BG2_Z40_rows1400_R1_f64_fixed_10", cut, "R1", "f64", -2.0, False, 32768)
    run("BG2_Z40_rows1400_R1_f64_es_n0_-4dB", cut, "R1", "f64", -4.0, True, 32768)
    del cut
    res.update(link_workloads(head, bg2, args, torch, dist, device, world, rank))
    return res


def link_workloads(head, bg2, args, torch, dist, device, world, rank):
    """The steps around the decoder: the Monte-Carlo loop body as one launch (ldpc_sweep_batch) beside
    the same work as three launches, the mixed CCSDS + 5G sum-product batch under the Doppler-shifted
    channel, the device encoder and 5G rate recovery."""
    from ldpc_amd import channel, config, encoder, nr_rate_match, sweep
    res = {}
    cfo5 = channel.doppler_cfo(4000.0, compensated=0.999)   # the carrier offset of the mixed CCSDS + 5G workload below
    for label, rn, rule, dtype, snr, frames, chan in (
            ("sweep_R0_f64_es_n0_1dB", head, "R0", np.float64, 1.0, min(args.frames, 1 << 19), {}),
            ("sweep_R1_f32_es_n0_1dB", head, "R1", np.float32, 1.0, min(args.frames, 1 << 19), {}),
            ("sweep_R1_f16x2_es_n0_1dB", head, "R1", np.float16, 1.0, min(args.frames, 1 << 19), {}),   # packed half precision: reduced precision
            # the large codes stage their LLRs in a scratch slice per workgroup: the fused launch saves the LLR
            # array and its HBM traffic, not time - channel and decode are bound by vector issue alike
            ("sweep_BG2_Z40_R1_f64_es_n0_-4dB", bg2, "R1", np.float64, -4.0, 32768, {}),
            # the two halves of the mixed CCSDS + 5G workload as whole Monte-Carlo steps: sum-product rule R3 under the Doppler-shifted channel
            ("sweep_R3_f32_doppler_es_n0_1dB", head, "R3", np.float32, 1.0, min(args.frames, 1 << 18), {"cfo": cfo5}),
            ("sweep_BG2_Z40_R3_f32_doppler_es_n0_-4dB", bg2, "R3", np.float32, -4.0, 16384, {"cfo": cfo5})):
        code, dec = rn.code, rn.dec
        cfg = config.PRESETS[rule].with_(max_iters=args.max_iters)
        cw = sweep.make_codewords(code, rn.G, frames, [31, rank], device)
        ctr = torch.zeros((5,), dtype=torch.int64, device=device)

        def fused():
            sweep.sweep_fused(dec, cw, snr, cfg, [32, rank], ctr, compute_dtype=dtype, **chan)

        def three():
            bt = sweep.noisy_batch(code, cw, snr, cfg, [32, rank], device, **chan)
            sweep.decode_batch(dec, bt, cfg, compute_dtype=dtype)
            sweep.count_errors(dec, bt)

        w1, _ = timed_steps(fused, 3, 2, torch, dist, world)
        w3, _ = timed_steps(three, 3, 2, torch, dist, world)
        res[label] = {"fused_gbit_s": frames * world * 3 * rn.k / w1 / 1e9, "three_launch_gbit_s": frames * world * 3 * rn.k / w3 / 1e9,
                      "frames": frames, "code": code.name, "rule": rule, "es_n0_db": snr, "early_exit": True,
                      "hbm_bytes_per_frame_fused": code.n, "hbm_bytes_per_frame_three_launches": 11 * code.n + 10}
        del cw
    # mixed workload: CCSDS and 5G frames of one link in one call, sum-product rule R3, Doppler-shifted BPSK
    cfg = config.R3.with_(max_iters=args.max_iters)
    cfo = cfo5
    parts = []
    for runner, fr, snr in ((head, 1 << 16, 1.0), (bg2, 1 << 13, -4.0)):
        cwp = sweep.make_codewords(runner.code, runner.G, fr, [33, rank], device)
        parts.append((runner.dec, sweep.noisy_batch(runner.code, cwp, snr, cfg, [34, rank], device, cfo=cfo)))
    ctr = torch.zeros((5,), dtype=torch.int64, device=device)
    step = lambda: sweep.decode_mixed(parts, cfg, compute_dtype=np.float32, counters=ctr)
    w, _ = timed_steps(step, 3, 2, torch, dist, world)
    info_bits = (1 << 16) * head.k + (1 << 13) * bg2.k
    ctr.zero_()
    step()
    torch.cuda.synchronize()
    st = sweep.summarize(ctr.cpu().numpy(), 1)
    res["mixed_ccsds_bg2_R3_f32_doppler"] = {"gbit_s": info_bits * world * 3 / w / 1e9, "frames": [1 << 16, 1 << 13],
                                             "codes": [head.name, bg2.name], "rule": "R3", "dtype": "f32",
                                             "cfo_cycles_per_symbol": cfo, "es_n0_db": [1.0, -4.0], "fer": st["fer"],
                                             "mean_iters": st["mean_iters"], "early_exit": True}
    # rooflines of the step (two decode kernels side by side on two streams, then the error accounting):
    # HBM from the algorithmic bytes of decode (5 n + 5 per frame) and accounting (2 n + 5 per frame) over the
    # step time; vector issue from the instruction counts on file for the two sum-product kernels
    # (they are bound by their transcendentals: tanh and atanh per edge and iteration)
    step_s = w / 3
    mixed_bytes = sum(fr * (7 * r.code.n + 10) for r, fr in ((head, 1 << 16), (bg2, 1 << 13)))
    roof = {"bound": "hbm", "achieved": mixed_bytes / step_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None,
            "bytes_per_step": mixed_bytes, "step_ms": step_s * 1e3}
    roof["frac"] = roof["achieved"] / roof["peak"]
    res["mixed_ccsds_bg2_R3_f32_doppler"]["roofline"] = roof
    valu = 0.0
    have = True
    for (runner, fr), key, kname in (((head, 1 << 16), "ccsds512/R3/f32/f32llr", "spec_spa_f32"), ((bg2, 1 << 13), "nr_bg2_z40/R3/f32/f32llr", "spec_spaw_f32")):
        sq = _profile_record("sq_counters", key, True, kname)   # counted in fixed-work mode: per executed frame-iteration
        have = have and sq is not None
        if sq:
            valu += sq["valu_wave_instructions_per_frame_iteration"] * fr
    if have:   # iterations executed: mean over both codes' frames (failures run one more than they report)
        executed_per_frame = st["mean_iters"] + st["fer"]
        ach = valu * executed_per_frame / step_s / 1e9
        res["mixed_ccsds_bg2_R3_f32_doppler"]["valu_roofline"] = {
            "bound": "valu", "achieved": ach, "peak": VALU_PEAK_GIPS, "unit": "G wave-instructions/s", "frac": ach / VALU_PEAK_GIPS,
            "note": "instruction counts per frame-iteration from the fixed-work counter passes of the two kernels, "
                    "iterations from this run's counters (mean over both codes)"}
    del parts
    # encoder and rate recovery of the 5G-structured table at Z = 384 (HBM-bound byte work)
    from ldpc_amd import codes
    bg1 = codes.load_packaged("nr_bg1_z384")
    enc = encoder.device_encoder(bg1)
    fr = 16384
    msg = torch.randint(0, 2, (fr, enc.k), device=device, dtype=torch.uint8)
    cwb = torch.empty((fr, enc.n), dtype=torch.uint8, device=device)
    stream = torch.cuda.current_stream().cuda_stream
    w, _ = timed_steps(lambda: enc.encode_ptr(msg.data_ptr(), fr, cwb.data_ptr(), stream), 3, 2, torch, dist, world)
    res["encode_bg1_z384"] = {"gbit_s": fr * enc.k * world * 3 / w / 1e9, "gb_s": fr * (enc.k + enc.n) * world * 3 / w / 1e9,
                              "frames": fr, "bytes_per_frame": enc.k + enc.n}
    r = nr_rate_match.RateMatch(1, 384, 2 * 8448)
    llr = torch.randn((fr, r.E), device=device, dtype=torch.float32)
    n_act = r.n_active()
    out = torch.empty((fr, n_act), device=device, dtype=torch.float32)
    w, _ = timed_steps(lambda: r.recover_device(llr, n_act, out=out), 3, 2, torch, dist, world)
    res["rate_recover_bg1_rate_half"] = {"gb_s": fr * 4 * (r.E + n_act) * world * 3 / w / 1e9, "frames": fr,
                                         "bytes_per_frame": 4 * (r.E + n_act), "E": r.E, "n_active": n_act}
    del out
    return res


if __name__ == "__main__":
    main()
