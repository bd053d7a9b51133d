"""bench.py's multi-rank control flow on a box without a GPU: two ranks over gloo, the device side replaced by
tests/bench_cpu_double.py (virtual clock, plain-torch conference mix).  What is checked is what the 8-GPU run depends on
and no single-GPU run exercises: every rank agrees on the leg count (the slowest rank's capacity), on the step count,
the exchange runs every tick, the split conferences' mix is verified, rank 0 prints one line with the whole-job value,
and failures exit non-zero on every rank instead of hanging."""
import json
import os
import socket
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOUBLE = os.path.join(ROOT, "tests", "bench_cpu_double.py")
COMMON = ["--steps", "16", "--warmup", "8", "--sweep-lo", "16384", "--sweep-hi", "65536", "--no-extras", "--no-cpu-baseline", "--worst-ticks", "200"]


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run(nranks, extra_env=None, args=()):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    env.pop("MSMI355X_BENCH_BACKEND", None)
    fd, detail = tempfile.mkstemp(prefix="bench_detail_", suffix=".json")
    os.close(fd)
    os.unlink(detail)
    args = (*args, "--detail", detail)
    try:
        r = _run(nranks, env, args)
        r.detail = json.load(open(detail)) if os.path.exists(detail) else None
        return r
    finally:
        if os.path.exists(detail):
            os.unlink(detail)


def _run(nranks, env, args):
    if nranks == 1:
        cmd = [sys.executable, DOUBLE, "--gpus", "1", *COMMON, *args]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nranks}", "--master-addr", "127.0.0.1",
               "--master-port", str(free_port()), DOUBLE, "--gpus", str(nranks), *COMMON, *args]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    # the probed port can be taken between free_port() and the launcher's bind, and a loaded box can time the rendezvous out: that
    # is the test's transport, not the control flow under test -- once more, on another port
    if nranks > 1 and r.returncode != 0 and any(w in r.stderr for w in ("Address already in use", "EADDRINUSE", "RendezvousConnectionError", "RendezvousTimeoutError",
                                                                        "Connection refused", "connect() timed out")):
        cmd[cmd.index("--master-port") + 1] = str(free_port())
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    return r


REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def the_line(r):
    """stdout is exactly ONE line, JSON, short enough for the driver to keep whole (it keeps the last 7 999 characters; round 4's
    26.5 KB line lost its head and the round went unmeasured), with the contract's keys; the full record is in the detail file"""
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and "peer ranks" not in ln]  # (the test transport's own chatter, interleaved between ranks, its newlines)
    assert len(lines) == 1 and lines[0].startswith("{"), (r.stdout[-2000:], r.stderr[-2000:])  # ONE line, from rank 0 only
    assert len(lines[0]) < 6000, len(lines[0])
    d = json.loads(lines[0])
    assert all(k in d for k in REQUIRED), sorted(d)
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["detail"].endswith(".json")
    assert r.detail is not None and r.detail["value"] == d["value"] and r.detail["ms_per_step"] == d["ms_per_step"]
    return d


def test_one_rank_sweep_settles_on_the_doubles_capacity():
    r = run(1)
    assert r.returncode == 0, r.stderr[-2000:]
    d = the_line(r)
    assert d["n_gpus"] == 1 and d["value"] == 22528  # 10 ms at 24 000 legs, sweep granularity 2048
    assert d["config"]["fits"] and d["config"]["worst_tick_ms"] < 10.0
    assert d["steps"] % 8 == 0 and d["steps"] >= 16 and r.detail["steps_requested"] == 16
    assert [p["streams"] for p in r.detail["config"]["capacity_sweep"]][:2] == [16384, 24576]
    assert "split_conferences" not in d["config"]


def test_final_line_is_short():
    """The line the driver parses carries numbers and short identifiers only -- also when every optional section is as large as a
    real run makes it: bench.short_line() over round 4's full 26.5 KB record (profiles/r04_bench_final.json) stays under 6 000
    characters and keeps metric / value / config.workload / roofline / cpu_baseline / plugin_path / scaler."""
    sys.path.insert(0, ROOT)
    import bench
    full = json.load(open(os.path.join(ROOT, "profiles", "r04_bench_final.json")))
    assert len(json.dumps(full)) > 20000
    s = json.dumps(bench.short_line(full, "bench_detail.json"), separators=(",", ":"))
    assert len(s) < bench.LINE_LIMIT <= 6000, len(s)
    d = json.loads(s)
    assert d["value"] == 124928 and d["roofline"]["frac"] == 0.7001 and d["roofline"]["bound"] == "hbm" and d["roofline"]["traffic"] > 0
    assert d["cpu_baseline"]["cores"] == 1 and d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] == 99.4
    assert d["plugin_path"]["legs"] == 51200 and d["scaler"]["frac"] == 0.5848 and d["video_pcie_inclusive"]["frames_per_s"] == 15452.5
    assert all(k in d for k in REQUIRED)
    longest = max((len(v) for v in _strings(d)), default=0)
    assert longest <= 230, longest   # no paragraph rides in a string value


def _strings(o):
    if isinstance(o, str):
        yield o
    elif isinstance(o, dict):
        for v in o.values():
            yield from _strings(v)
    elif isinstance(o, list):
        for v in o:
            yield from _strings(v)


def test_two_ranks_agree_on_the_slower_ranks_capacity_and_exchange_every_tick():
    r = run(2, {"DOUBLE_SLOW_RANK1": "1.25"})  # rank 1's device carries 19 200 legs in 10 ms -> 18 432
    assert r.returncode == 0, r.stderr[-3000:]
    d = the_line(r)
    sc = d["config"]["split_conferences"]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    per_rank = d["config"]["streams_per_gpu"]
    # 18 432 is the slower rank's sweep result; with the 64 split conferences' 16 local members the rig rounds to whole
    # conferences, and the deployed tick (exchange included) may step down by 2048 once
    assert per_rank in (18432, 18432 - 2048) and d["value"] == 2 * per_rank
    assert sc["count"] == 64 and sc["members_per_rank"] == 16 and sc["mix_bit_exact_vs_single_gpu"] is True
    assert sc["backend"] == "gloo" and "TEST BACKEND" in d["config"]["parallelism"] and "TEST BACKEND" in r.detail["config"]["parallelism"]
    assert sc["allreduce_bytes_per_tick"] == 64 * 480 * 4 and sc["allreduce_alone_us"] is not None
    assert d["steps"] % 8 == 0
    assert d["config"]["worst_tick_ms"] < 10.0


def test_a_slow_exchange_steps_the_leg_count_down_until_the_deployed_tick_fits():
    """The sweep measures the chain alone; the deployed tick carries the exchange too.  With 1.5 ms of exchange per tick
    the sweep's 22 528 legs (9.4 ms) no longer fit: the run must come down (2048 at a time, then faster) and still report
    a count that fits, not zero."""
    r = run(2, {"DOUBLE_EXCHANGE_MS": "1.5"})
    assert r.returncode == 0, r.stderr[-3000:]
    d = the_line(r)
    per_rank = d["config"]["streams_per_gpu"]
    assert d["config"]["fits"] and d["config"]["worst_tick_ms"] < 10.0
    assert 16384 <= per_rank <= 20480 and d["value"] == 2 * per_rank
    assert abs(d["config"]["worst_tick_ms"] - (per_rank / 2400.0 + 1.5)) < 0.05


def test_a_wrong_partial_sum_fails_every_rank():
    r = run(2, {"DOUBLE_BREAK_RANK": "1"}, args=["--streams", "8192"])
    assert r.returncode != 0
    assert "differs from the single-GPU mix" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_gpus_flag_without_the_launcher_is_a_usage_error():
    r = subprocess.run([sys.executable, DOUBLE, "--gpus", "2", *COMMON], capture_output=True, text=True, timeout=120, cwd=ROOT,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 2 and "needs torch.distributed.run" in r.stderr


def test_ranks_that_do_not_divide_a_conference_are_refused():
    r = run(3, args=["--streams", "4096"])
    assert r.returncode != 0 and "do not divide a 32-party conference" in r.stderr


def test_the_c_exchange_comes_up_on_every_rank_or_the_run_fails_on_every_rank():
    """ONE multi-GPU contract (bench.HipPlatform.exchange, the code the 8-GPU run executes, here with the double's communicator
    behind it): rank 0's id reaches the others, every rank probes its communicator, the ranks agree -- and if ANY rank could
    not bring mi_exchange up, EVERY rank leaves with exit code 3 and RCCL's reason on stderr: no substitute transport, no
    rank left waiting in a collective, no JSON line."""
    ok = run(2, {"DOUBLE_EXCHANGE_C": "1"}, args=["--streams", "8192"])
    assert ok.returncode == 0, ok.stderr[-3000:]
    d = the_line(ok)
    assert "mi_exchange_allreduce_i32 (C ABI, RCCL over xGMI" in ok.detail["config"]["parallelism"]
    assert "mi_exchange_allreduce_i32 (RCCL)" in d["config"]["parallelism"]
    assert d["config"]["split_conferences"]["mix_bit_exact_vs_single_gpu"] is True
    bad = run(2, {"DOUBLE_EXCHANGE_C": "1", "DOUBLE_EXCHANGE_FAIL_RANK": "1"}, args=["--streams", "8192"])
    assert bad.returncode != 0
    assert not [ln for ln in bad.stdout.splitlines() if ln.startswith("{")]
    assert bad.stderr.count("could not be set up") >= 2 and "unhandled system error" in bad.stderr   # both ranks said so, with the reason
    assert "torch.distributed's RCCL communicator instead" not in bad.stderr


def test_a_late_tick_that_was_the_submitting_threads_stall_is_told_from_one_the_device_took_long_over():
    """bench.TickTimes keeps, beside every tick's HIP-event time, the host's own time from before the first event's record to the
    return of the launch call; host_stalls_only(): every late tick of the series is covered (>= 90 % of its excess over the
    median) by that -- the acceptance loop then tries the count once more instead of stepping down (a stall of the box would
    make a tick of one leg late); series_stats() carries the evidence into the line."""
    import numpy as np
    sys.path.insert(0, ROOT)
    import bench
    v = bench.TickTimes(3000)
    v[:] = 9.1
    v.submit[:] = 0.02
    assert not bench.host_stalls_only(v)                      # nothing late
    v[93], v.submit[93] = 21.3, 12.25                         # the thread was held up for 12 ms between the event and the launch
    assert bench.host_stalls_only(v)
    st = bench.series_stats(v)
    assert st["late"] == 1 and st["slowest"][0] == [93, 21.3] and st["slowest_host_submit_ms"][0] == 12.25 and st["host_submit_max_ms"] == 12.25
    v[700], v.submit[700] = 10.4, 0.02                        # ... and one the DEVICE took long over: the series is a verdict again
    assert not bench.host_stalls_only(v)
    w = bench.TickTimes(100)
    w[:] = 9.0
    w.submit[:] = 0.02
    w[5], w.submit[5] = 11.0, 0.6                             # a small hiccup that explains a third of the lateness does not excuse it
    assert not bench.host_stalls_only(w)
    assert not bench.host_stalls_only(np.full(10, 11.0))      # a plain array (no host times): never excused
    assert float(np.median(v)) == 9.1 and float(v.max()) == 21.3 and isinstance(v[1:5], bench.TickTimes)


def test_a_one_off_stall_in_the_sweeps_first_point_does_not_lower_the_proposal():
    """some boxes hold the device for 35-45 ms once, at the first timed tick of the sweep's first point: a point whose
    series holds ONE tick more than 5 ms over its median is measured once more -- the proposal, and with it `value`, is what it
    is without the stall; the point says what it saw"""
    r = run(1, {"DOUBLE_STALL_AT_TIMER": "3:38"})
    assert r.returncode == 0, r.stderr[-2000:]
    d = the_line(r)
    first = r.detail["config"]["capacity_sweep"][0]
    assert first["streams"] == 16384 and first["fits"] and first["first_series_held_a_stall"]["at"] == 0
    assert first["first_series_held_a_stall"]["tick_ms_worst"] > 38.0
    assert d["value"] == 22528


def test_the_plugin_probes_search_goes_up_by_steps_and_down_by_bisection():
    """bench.search_counts (plugin_path_probe's search): from the first count up a step at a time while it fits, three more at most;
    a first count that does not fit has the counts below it bisected -- every count measured by the same function"""
    sys.path.insert(0, ROOT)
    import bench
    counts = list(range(8192, 98304 + 1, 8192))
    for limit, want, most in ((60000, 57344, 3), (49152, 49152, 2), (30000, 24576, 4), (9000, 8192, 4), (4000, None, 4), (200000, 73728, 4)):
        seen = []

        def measure(c):
            seen.append(c)
            return {"fits": c <= limit, "legs": c}
        got = bench.search_counts(measure, counts, counts.index(49152))
        assert (got and got["legs"]) == want, (limit, got, seen)
        assert seen[0] == 49152 and len(seen) <= most and len(set(seen)) == len(seen), (limit, seen)
