"""`bench.py --gpus N` must decode on N ranks whichever way it is started (the Monte-Carlo loop it
shards is /root/reference ldpc.py:637-667: frames are independent, counters are summed)."""
import os
import sys

import pytest

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def test_single_gpu_runs_in_process():
    assert bench.launch_plan(1, {}, [], device_count=1) == ("run", 1, 0, 0)


def test_plain_start_with_several_gpus_spawns_the_ranks():
    kind, cmd = bench.launch_plan(8, {}, ["--gpus", "8", "--steps", "5"], script="/x/bench.py", device_count=8, port=29511)
    assert kind == "spawn"
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=8" in cmd and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "8", "--steps", "5"]


def test_rank_of_a_launched_job_runs_as_that_rank():
    env = {"WORLD_SIZE": "4", "RANK": "3", "LOCAL_RANK": "3"}
    assert bench.launch_plan(4, env, [], device_count=8) == ("run", 4, 3, 3)


def test_world_size_must_equal_gpus():
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.launch_plan(8, {"WORLD_SIZE": "2", "RANK": "0"}, [], device_count=8)
    with pytest.raises(SystemExit, match="WORLD_SIZE=2"):
        bench.launch_plan(1, {"WORLD_SIZE": "2", "RANK": "0"}, [], device_count=8)


def test_more_ranks_than_gpus_is_refused_under_rccl_only():
    with pytest.raises(SystemExit, match="shows 1 GPU"):
        bench.launch_plan(2, {}, [], device_count=1)
    kind, _ = bench.launch_plan(2, {"LDPC_BENCH_BACKEND": "gloo"}, [], device_count=1)
    assert kind == "spawn"
