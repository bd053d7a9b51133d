"""mi_exchange: the conference mixer's cross-GPU step through the C ABI, straight on RCCL.  The GPU test box has one
device, so the communicator has one rank there (RCCL refuses two ranks on one device): init, the all-reduce enqueued on
the context's stream between partial_sum and finalize, destroy.  With two or more devices the same runs with a rank per
device from threads of this one process, and two contexts on different devices are driven from one thread."""
import threading

import numpy as np
import pytest

import mediastreamer2_amd as ms

pytestmark = pytest.mark.gpu


def _signal(nconf, members, ns, seed=3):
    rng = np.random.default_rng(seed)
    return rng.integers(-16000, 16000, (nconf, members, ns), dtype=np.int16)


def test_exchange_with_one_rank_sits_between_partial_sum_and_finalize(ctx, oracle):
    torch = pytest.importorskip("torch")
    nconf, mm, ns = 16, 32, 480
    x = _signal(nconf, mm, ns)
    uid = ms.Exchange.unique_id(ctx)
    assert len(uid) == 128 and any(uid)
    ex = ms.Exchange(ctx, 1, 0, uid)
    mx = ms.MixerBatch(ctx, nconf, mm, ns)
    d_in = torch.from_numpy(x).cuda()
    d_sum = torch.zeros((nconf, ns), dtype=torch.int32, device="cuda")
    d_out = torch.zeros_like(d_in)
    torch.cuda.synchronize()
    for _ in range(3):  # all three on the context's stream: no event, no host wait in between
        mx.partial_sum(d_in, d_sum)
        ex(d_sum)
        mx.finalize(d_in, d_sum, d_out)
    ctx.sync()
    np.testing.assert_array_equal(d_sum.cpu().numpy(), x.astype(np.int64).sum(1))
    for c in range(nconf):
        ref, _ = oracle.mixer_tick(x[c])
        np.testing.assert_array_equal(d_out.cpu().numpy()[c], ref)
    with pytest.raises(ms.MiError):
        ms.Exchange(ctx, 2, 5, uid)  # rank outside the communicator
    ex.close()
    mx.close()


def test_two_contexts_on_two_devices_from_one_thread_and_a_rank_per_device():
    torch = pytest.importorskip("torch")
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two visible GPUs")
    assert ms.Context(0).L.mi_device_count() >= 2
    nconf, mm, ns, world = 8, 32, 480, 2
    x = _signal(nconf, mm, ns, seed=9)
    ctxs = [ms.Context(d) for d in range(world)]
    # one thread, two devices: every entry point makes its context's device current
    mixers = [ms.MixerBatch(ctxs[d], nconf, mm // world, ns) for d in range(world)]
    ins = [torch.from_numpy(np.ascontiguousarray(x[:, d * 16:(d + 1) * 16])).to(f"cuda:{d}") for d in range(world)]
    sums = [torch.zeros((nconf, ns), dtype=torch.int32, device=f"cuda:{d}") for d in range(world)]
    outs = [torch.zeros_like(i) for i in ins]
    for d in range(world):
        torch.cuda.synchronize(d)
        mixers[d].partial_sum(ins[d], sums[d])
    for d in range(world):
        ctxs[d].sync()
    np.testing.assert_array_equal(sums[0].cpu().numpy() + sums[1].cpu().numpy(), x.astype(np.int64).sum(1))
    # a rank per device, joined from two threads of this process (ncclCommInitRank blocks until both are in)
    uid = ms.Exchange.unique_id(ctxs[0])
    exs, errs = [None, None], []

    def join(d):
        try:
            exs[d] = ms.Exchange(ctxs[d], world, d, uid)
        except Exception as e:  # pragma: no cover
            errs.append(e)
    th = [threading.Thread(target=join, args=(d,)) for d in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    for d in range(world):
        exs[d](sums[d])
        mixers[d].finalize(ins[d], sums[d], outs[d])
    for d in range(world):
        ctxs[d].sync()
    whole = ms.MixerBatch(ctxs[0], nconf, mm, ns)
    ref = whole.process(torch.from_numpy(x).to("cuda:0"))
    ctxs[0].sync()
    for d in range(world):
        np.testing.assert_array_equal(outs[d].cpu().numpy(), ref.cpu().numpy()[:, d * 16:(d + 1) * 16])
    for o in exs + mixers + [whole]:
        o.close()
