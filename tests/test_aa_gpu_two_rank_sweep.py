"""Frame-sharded Monte-Carlo sweep on two ranks against the same sweep on one rank.

Runs first (file name) and only ever touches the GPU in freshly spawned child processes: every
rank is a new interpreter on device 0, the collective (the all-reduce of the error counters, the
only one of the path; ldpc.py:637-667 is the loop it serves) goes through gloo.  The summed
counters of the two-rank run must EQUAL those of the one-rank run: a frame's codeword and noise
depend on its absolute number only.
"""
import json
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FRAMES = 50_000          # not a multiple of the seeding chunk: ragged ranges on both ranks
SNRS = (0.5, 2.0)


def _rank_main(rank, world, port, out_dir, rule, code_name, dtype):
    import torch
    import torch.distributed as dist

    from ldpc_amd import codes, config, sweep
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    code = codes.load_packaged(code_name)
    G = codes.load_packaged_generator(code_name)
    cfg = config.PRESETS[rule]
    comp = np.float64 if dtype == "f64" else np.float32
    local = sweep.ber_sweep(code, G, cfg, SNRS, FRAMES, dev, comp, seed=31, batch=1 << 15, reduce=False)
    total = sweep.ber_sweep(code, G, cfg, SNRS, FRAMES, dev, comp, seed=31, batch=1 << 15)
    with open(os.path.join(out_dir, f"w{world}_r{rank}.json"), "w") as f:
        json.dump({"local": local, "total": total}, f)
    if world > 1:
        dist.destroy_process_group()


def _run(world, out_dir, rule, code_name, dtype):
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_rank_main, args=(world, port, out_dir, rule, code_name, dtype), nprocs=world, join=True)
    return [json.load(open(os.path.join(out_dir, f"w{world}_r{r}.json"))) for r in range(world)]


@pytest.mark.parametrize("rule,code_name,dtype", [("R0", "ccsds512", "f64"), ("R1", "wifi648", "f32")])
def test_two_rank_sweep_equals_one_rank(tmp_path, rule, code_name, dtype):
    from ldpc_amd import sweep
    one = _run(1, str(tmp_path), rule, code_name, dtype)[0]
    two = _run(2, str(tmp_path), rule, code_name, dtype)
    names = sweep.COUNTER_NAMES
    for si in range(len(SNRS)):
        ref = [one["total"][si][k] for k in names]
        assert ref[0] == FRAMES and one["local"][si] == one["total"][si]
        for r in range(2):
            assert [two[r]["total"][si][k] for k in names] == ref           # every rank holds the sum
        parts = np.array([[two[r]["local"][si][k] for k in names] for r in range(2)])
        assert parts[:, 0].tolist() == [FRAMES // 2, FRAMES // 2]           # each rank decoded its half only
        assert parts.sum(axis=0).tolist() == ref
    # the sweep saw errors and convergence (the comparison is not trivially 0 == 0)
    assert one["total"][0]["not_converged"] > 0 and one["total"][0]["bit_errors"] > 0
    assert one["total"][1]["iterations"] > 0
