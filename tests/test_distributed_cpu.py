"""N>1 path on CPU: world_size-2 `gloo` run of the one exchange step on the hot path, the
conference mixer's int32 all-reduce, plus the static stream sharding.  The GPU
kernels are replaced by the oracle here (no GPU in this container); what is under test is the
placement logic and the collective of mediastreamer2_amd/sharding.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    import oracle
    from mediastreamer2_amd import sharding
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nconf, mm, ns = 6, 32, 480
    rng = np.random.default_rng(123)
    x = rng.normal(0, 2500, (nconf, mm, ns)).round().clip(-32767, 32767).astype(np.int16)
    place = sharding.place_conferences(nconf, mm, world, rank, split_conf=[1, 4])
    res = {}
    # whole conferences: mixed locally, no communication
    for c in place.local_whole:
        res[c] = oracle.mixer_tick(x[c])[0]
    # split conferences: partial int32 sums of the local members, ONE all-reduce, local finalize
    lo, hi = place.member_lo, place.member_hi
    part = np.stack([oracle.mixer_tick(x[c, lo:hi])[1] for c in place.split])
    t = torch.from_numpy(part.astype(np.int32))
    sharding.allreduce_partial_sums(t)
    tot = t.numpy().astype(np.int64)
    for i, c in enumerate(place.split):
        res[c] = (lo, hi, np.clip(tot[i][None] - x[c, lo:hi].astype(np.int64), -32767, 32767).astype(np.int16))
    q.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_mixer_allreduce_gloo_world2():
    import torch.multiprocessing as mp
    import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    nconf, mm, ns = 6, 32, 480
    rng = np.random.default_rng(123)
    x = rng.normal(0, 2500, (nconf, mm, ns)).round().clip(-32767, 32767).astype(np.int16)
    seen = set()
    for rank, res in got:
        for c, v in res.items():
            full = oracle.mixer_tick(x[c])[0]
            if isinstance(v, tuple):
                lo, hi, out = v
                np.testing.assert_array_equal(out, full[lo:hi])
                seen.add((c, lo, hi))
            else:
                np.testing.assert_array_equal(v, full)
                seen.add((c, 0, mm))
    covered = {c: sorted((lo, hi) for cc, lo, hi in seen if cc == c) for c in range(nconf)}
    for c in range(nconf):
        spans = covered[c]
        assert spans[0][0] == 0 and spans[-1][1] == mm
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))


def test_shard_range_partitions_exactly():
    sys.path.insert(0, ROOT)
    from mediastreamer2_amd.sharding import shard_range, place_conferences
    for n in (0, 1, 7, 4096, 32768):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, w, r) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
    # BASELINE config 4: 1024 conferences x 32 members over 8 GPUs -> 128 whole conferences each
    per = [place_conferences(1024, 32, 8, r) for r in range(8)]
    assert all(len(p.local_whole) == 128 and not p.split for p in per)
    assert sorted(c for p in per for c in p.local_whole) == list(range(1024))
