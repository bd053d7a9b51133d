"""bench.py's N > 1 path on the GPU: two ranks, the real kernels, the split-conference exchange
(mi_mixer_partial_sum -> int32 all-reduce -> mi_mixer_finalize with explicit stream events) checked bit for bit
against the single-GPU mix.  With two visible GPUs the collective is RCCL ("nccl"); on a one-GPU box both ranks
share the device and the collective runs over gloo (RCCL refuses two ranks on one device) -- the control flow and
the kernels are the same."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, env_extra):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **env_extra)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--streams", "4096",
           "--steps", "8", "--warmup", "8", "--min-timed-s", "0.05", "--no-extras", "--no-cpu-baseline",
           "--worst-ticks", "200", "--zero-ticks", "32", "--roofline-ticks", "8"]
    p = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def _check(line, world, backend):
    assert line["n_gpus"] == world and line["scaling"] == "weak"
    assert line["config"]["streams_per_gpu"] == 4096
    # two ranks time-sharing one GPU with a host-staged collective may miss the 10 ms budget: the line must then say so
    assert line["value"] == (world * 4096 if line["config"]["fits"] else 0)
    if backend == "nccl":
        assert line["config"]["fits"] and line["value"] == world * 4096
    sc = line["config"]["split_conferences"]
    assert sc["mix_bit_exact_vs_single_gpu"] is True and sc["backend"] == backend and sc["members_per_rank"] == 32 // world
    assert sc["allreduce_alone_us"] > 0
    assert ("gloo (TEST BACKEND)" if backend == "gloo" else "mi_exchange_allreduce_i32 (RCCL)") in line["config"]["parallelism"]
    assert len(json.dumps(line)) < 6000
    assert line["roofline"]["frac"] > 0


def test_bench_two_ranks_share_one_gpu_over_gloo():
    line = _run(2, {"MSMI355X_BENCH_BACKEND": "gloo", "MSMI355X_BENCH_DEVICE": "0"})
    _check(line, 2, "gloo")


def test_bench_two_ranks_rccl():
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs (RCCL refuses two ranks on one device)")
    line = _run(2, {})
    _check(line, 2, "nccl")
