"""Host-side pieces that need no GPU: LLR producers, the shim's bookkeeping, frame sharding and
the counter all-reduce over two gloo ranks."""
import os
import socket

import numpy as np
import pytest

from ldpc_amd import channel, config, sweep
from oracle import bp_oracle


def test_llr_producer_equals_the_oracle_formulas():
    rng = np.random.default_rng(0)
    y = rng.standard_normal((3, 200))
    for name, cfg in config.PRESETS.items():
        a = channel.llr_from_samples(y, [1.0, 2.5, -1.0], cfg)
        b = bp_oracle.llr_from_samples(y, [1.0, 2.5, -1.0], cfg.rule)
        assert np.array_equal(a, b), name
    raw = channel.llr_from_samples(y, 0.0, config.R1, compat_prefix=False)
    assert np.array_equal(raw, 4 * y)


def test_bpsk_polarities():
    cw = np.array([[0, 1, 1, 0]])
    rng = np.random.default_rng(1)
    y0 = channel.bpsk_awgn(cw, 60.0, config.R0.plus_is_bit, rng)     # ldpc.py:248: bit 1 -> +1
    y1 = channel.bpsk_awgn(cw, 60.0, config.R1.plus_is_bit, rng)     # pyldpc convention: bit 0 -> +1
    assert np.array_equal(np.sign(y0), [[-1, 1, 1, -1]])
    assert np.array_equal(np.sign(y1), [[1, -1, -1, 1]])


def test_presets_carry_the_reference_constants():
    assert (config.R0.max_iters, config.R0.scale) == (10, 0.75)
    assert (config.R1.max_iters, config.R1.scale) == (30, 0.75)
    assert (config.R2.max_iters, config.R2.clip_tanh_arg, config.R2.clip_prod, config.R2.clip_post) == (50, 20.0, 0.999999999, 30.0)
    assert (config.R3.max_iters, config.R3.scale, config.R3.clip_prod, config.R3.clip_post) == (50, 0.85, 0.99999999, 50.0)
    assert (config.R4.max_iters, config.R4.clip_post) == (40, 500.0)
    assert config.R0.plus_is_bit == 1 and config.R3.plus_is_bit == 0


def test_shard_ranges_partition_the_frames():
    for total in (0, 1, 7, 1 << 20):
        for world in (1, 2, 3, 8):
            parts = [sweep.shard_range(total, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1


def test_summary_of_counters():
    s = sweep.summarize(np.array([200, 63, 0, 1118, 1492]), 512)
    assert s["fer"] == 63 / 200 and s["ber"] == 1118 / (200 * 512) and s["mean_iters"] == 7.46


def _rank_main(rank, world, port, out):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = sweep.shard_range(1000, rank, world)
    # each rank "decodes" its own frames: counters derived from the frame indices it owns
    idx = np.arange(a, b)
    local = torch.tensor([len(idx), int((idx % 7 == 0).sum()), 0, int(idx.sum() % 1000), int(idx.sum())], dtype=torch.int64)
    total = sweep.allreduce_counters(local.clone())
    if rank == 0:
        np.save(out, total.numpy())
    dist.destroy_process_group()


def test_counter_allreduce_two_ranks(tmp_path):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "tot.npy")
    mp.spawn(_rank_main, args=(2, port, out), nprocs=2, join=True)
    tot = np.load(out)
    idx = np.arange(1000)
    assert tot[0] == 1000 and tot[1] == (idx % 7 == 0).sum() and tot[4] == idx.sum()
    a0 = np.arange(*sweep.shard_range(1000, 0, 2)).sum() % 1000
    a1 = np.arange(*sweep.shard_range(1000, 1, 2)).sum() % 1000
    assert tot[3] == a0 + a1


def test_allreduce_is_a_noop_without_a_process_group():
    import torch
    c = torch.tensor([1, 2, 3, 4, 5], dtype=torch.int64)
    assert sweep.allreduce_counters(c) is c and c.tolist() == [1, 2, 3, 4, 5]


def test_shim_bookkeeping_without_gpu():
    from ldpc_amd import shim
    H = shim.make_regular_ldpc(24, 3, 6, seed=1)
    assert H.shape == (12, 24) and (H.sum(0) == 3).all() and (H.sum(1) == 6).all()
    e = shim.LdpcPyEncoder(4, 8, 648, readDataMatrix=True)
    assert e.H.shape == (324, 648) and e.m == 324.0            # 648 * 4/8, ldpc.py:30
    q = shim.LdpcPyEncoder(4, 8, 16, readDataMatrix=True, compat_loop_bounds=True)
    assert q.m == 8.0 and q._config(config.R0).num_active_checks == 8      # the ccsds.py:18 object
    b = shim.BpskPyEncoder(4, 8, 648, readDataMatrix=True)
    assert b.m == 324 and b._config(config.R1).num_refresh_vars == -1
    c = shim.ChannelPyEncoder(4, 5, 648, readDataMatrix=True, compat_loop_bounds=True)
    assert c.H.shape == (1400, 2080) and c._config(config.R3).num_refresh_vars == 648   # channel.py:646
    msg = np.random.default_rng(0).integers(0, 2, e.G.shape[1])
    np.random.seed(3)
    y = e.encode(msg, 2.0)
    assert e.isValidCodeword(e.originalEncoded) and y.shape == (648,) and e.SNR == 2.0
    assert e.encode(msg[:-1], 2.0) is None                     # ldpc.py:43-45
    np.random.seed(3)
    y2 = e.addNoiseBPSK(1.0, e.originalEncoded)
    np.random.seed(3)
    ref = 2 * e.originalEncoded - 1 + np.sqrt(1 / (10 ** 0.1 * 2)) * np.random.randn(648)
    assert np.array_equal(y2, ref)
