"""Test double for bench.py's device side (TEST INFRASTRUCTURE): runs bench.main() with the HIP platform, the kernel
library and the chain rig replaced by CPU stand-ins, so the multi-rank control flow -- process-group bring-up, capacity
agreement (MIN over ranks), step count agreement, barriers, the per-tick partial-sum exchange, the split-mix check, the
one JSON line from rank 0 and the exit codes -- runs over gloo on a box without a GPU.  Nothing here computes audio the
way the kernels do except the conference mix, which is the mixer's definition (audiomixer.c:33-44,:301-344: int32 sum,
own contribution removed, saturation to +-32767) in plain torch; time is VIRTUAL: a tick of n legs costs
n / LEGS_PER_MS milliseconds on the double's clock (times DOUBLE_SLOW_RANK<r> for rank r), so capacities are exact.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P \
      tests/bench_cpu_double.py --gpus 2 --steps 16 --warmup 8 --sweep-lo 16384 --sweep-hi 65536"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LEGS_PER_MS = 2400.0  # 10 ms at 24 000 legs: the sweep (steps of 2048) settles on 22 528


class Clock:
    now_ms = 0.0


class Graph:
    def __init__(self, ops):
        self.ops = ops

    def launch(self):
        for op in self.ops:
            op()

    def close(self):
        pass


class Context:
    stream = 0

    def __init__(self, device):
        self.recording = None
        self.t0 = 0.0

    def props(self):
        return {"name": "cpu double", "cu_count": 0}

    def sync(self):
        pass

    timer_stops = 0

    def timer_start(self):
        self.t0 = Clock.now_ms

    def timer_stop(self):
        # DOUBLE_STALL_AT_TIMER="k:ms": the k-th timed interval of the process comes out `ms` longer (a one-off stall of the device:
        # k = 3 is the first tick of the sweep's first point)
        Context.timer_stops += 1
        at, _, extra = os.environ.get("DOUBLE_STALL_AT_TIMER", "0:0").partition(":")
        if Context.timer_stops == int(at):
            Clock.now_ms += float(extra)
        return Clock.now_ms - self.t0

    def capture_begin(self):
        self.recording = []

    def capture_end(self):
        ops, self.recording = self.recording, None
        return Graph(ops)


def mix_minus_own(torch, rows, total):
    """rows [conf][members][n] int16, total [conf][n] int32 -> each member hears the sum of the others, saturated"""
    return (total[:, None, :] - rows.to(torch.int32)).clamp(-32767, 32767).to(torch.int16)


class MixerBatch:
    def __init__(self, ctx, nconf, members, nsamples):
        pass

    def process(self, rows, out):
        import torch
        out.copy_(mix_minus_own(torch, rows, rows.to(torch.int32).sum(1)))

    def close(self):
        pass


class Rig:
    """stands in for bench.ChainRig: same attributes and calls, a virtual cost per tick, a real split-conference mix"""
    MEMBERS, RING = 32, 16

    def __init__(self, ms, torch, ctx, nstreams, world=1, rank=0, nsplit=0, stagger=True):
        self.torch, self.ctx, self.nsplit = torch, ctx, nsplit
        self.mloc = self.MEMBERS // world if nsplit else 0
        self.nconf = max(1, (nstreams - nsplit * self.mloc) // self.MEMBERS)
        self.n = self.nconf * self.MEMBERS + nsplit * self.mloc
        self.cost_ms = self.n / LEGS_PER_MS * float(os.environ.get(f"DOUBLE_SLOW_RANK{rank}", "1"))
        self.rank = rank
        if nsplit:
            g = torch.Generator().manual_seed(1234 + rank)
            self.feed = [torch.randint(-12000, 12000, (nsplit, self.mloc, 480), generator=g, dtype=torch.int16) for _ in range(self.RING)]
            self.split_in = torch.zeros((nsplit, self.mloc, 480), dtype=torch.int16)
            self.split_out = torch.zeros_like(self.split_in)
            self.d_sum = torch.zeros((nsplit, 480), dtype=torch.int32)

    def _tick(self, t):
        Clock.now_ms += self.cost_ms
        if self.nsplit:
            self.split_in.copy_(self.feed[t % self.RING])
            self.d_sum.copy_(self.split_in.to(self.torch.int32).sum(1))
            if os.environ.get("DOUBLE_BREAK_RANK") == str(self.rank):
                self.d_sum[0, 0] += 1  # a rank contributing a wrong partial sum must be caught by the split-mix check

    def tick(self, t, parts=None):
        self._tick(t)

    def _finalize(self):
        if self.nsplit:
            self.split_out.copy_(mix_minus_own(self.torch, self.split_in, self.d_sum))

    def finalize(self):
        if self.ctx.recording is not None:
            self.ctx.recording.append(self._finalize)
        else:
            self._finalize()

    def capture(self, ticks):
        if self.nsplit and len(ticks) > 1:
            raise RuntimeError("a tick with a collective in it is captured alone")
        return Graph([(lambda t=t: self._tick(t)) for t in ticks])

    def warm(self, nt=None):
        for t in range(nt or self.RING):
            self._tick(t)

    def overflows(self):
        return 0

    def state_bytes(self):
        return 0

    def close(self):
        pass


class Exchange:
    """stands in for mediastreamer2_amd.Exchange (mi_exchange on RCCL) so that bench.HipPlatform.exchange's own control flow --
    the id from rank 0, the probe, the agreement of the ranks, ONE contract: it works everywhere or the run fails everywhere
    -- runs over gloo (DOUBLE_EXCHANGE_C=1); DOUBLE_EXCHANGE_FAIL_RANK=r makes rank r's communicator fail to start"""

    @staticmethod
    def unique_id(ctx):
        return bytes([7]) * 128

    def __init__(self, ctx, world, rank, idb):
        if os.environ.get("DOUBLE_EXCHANGE_FAIL_RANK") == str(rank):
            raise RuntimeError("ncclCommInitRank: unhandled system error (the double's)")
        assert idb == bytes([7]) * 128

    def __call__(self, d_sum):
        import torch.distributed as dist
        dist.all_reduce(d_sum)
        Clock.now_ms += float(os.environ.get("DOUBLE_EXCHANGE_MS", "0.02"))

    def close(self):
        pass


class KernelLibraryDouble:
    Context = Context
    MixerBatch = MixerBatch
    Exchange = Exchange


class CpuDouble(bench.HipPlatform):
    device = "cpu"
    backend = "gloo"

    def available(self, torch):
        return True

    def select(self, torch, local):
        pass

    def sync(self, torch):
        pass

    def release(self, torch):
        pass

    def load(self):
        return KernelLibraryDouble

    def converged(self, ms, torch, ctx, rank):
        return None  # no canceller to converge: the double's ticks cost the same in any state

    def exchange(self, ctx, local, dist_, rank, world, backend):
        import torch.distributed as dist
        if os.environ.get("DOUBLE_EXCHANGE_C"):  # the product's own bring-up code, with the double's communicator behind it
            return bench.HipPlatform.exchange(self, ctx, local, dist_, rank, world, "nccl")

        def exchange(d_sum):
            dist.all_reduce(d_sum)
            Clock.now_ms += float(os.environ.get("DOUBLE_EXCHANGE_MS", "0.02"))  # the exchange's share of the virtual tick

        return exchange


if __name__ == "__main__":
    bench.PLATFORM = CpuDouble()
    bench.ChainRig = Rig
    bench.main()
