"""GPU parity: mi_mixer_* vs the oracle's restatement of audiomixer.c.
Pure integer arithmetic: results must be BIT-EXACT."""
import numpy as np
import pytest

import mediastreamer2_amd as ms
from conftest import synth_pcm

pytestmark = pytest.mark.gpu
L, A, O = ms.MI_MIX_LINKED, ms.MI_MIX_ACTIVE, ms.MI_MIX_OUTPUT


def _oracle_conf(oracle, x, flags, gain, has, conf_mode):
    linked = (flags & L) != 0
    has_data = (has != 0) & linked
    out, s = oracle.mixer_tick(x, has_data.astype(np.uint8), gain, ((flags & A) != 0).astype(np.uint8),
                               ((flags & O) != 0).astype(np.uint8), conf_mode)
    return out, s


@pytest.mark.parametrize("nconf,mm,ns", [(3, 32, 480), (5, 4, 160), (2, 50, 480), (4, 9, 80), (1, 17, 320),
                                         (3, 5, 441), (2, 12, 110)])  # 44.1 / 11.025 kHz ticks: not multiples of 4
@pytest.mark.parametrize("conf_mode", [1, 0])
def test_mixer_bit_exact(ctx, oracle, nconf, mm, ns, conf_mode):
    rng = np.random.default_rng(nconf * 1000 + mm)
    mx = ms.MixerBatch(ctx, nconf, mm, ns)
    x = np.stack([[synth_pcm(c * mm + m, ns, sigma=9000.0) for m in range(mm)] for c in range(nconf)])
    x[0, 0, :8] = [-32768, 32767, -32768, 32767, 0, 1, -1, -32767]
    flags = np.full((nconf, mm), L | A | O, np.uint8)
    flags[rng.random((nconf, mm)) < 0.15] &= ~np.uint8(A)      # inactive (A4)
    flags[rng.random((nconf, mm)) < 0.15] &= ~np.uint8(O)      # output disabled
    flags[rng.random((nconf, mm)) < 0.10] = 0                  # unlinked pin
    gain = np.ones((nconf, mm), np.float32)
    sel = rng.random((nconf, mm)) < 0.3
    gain[sel] = rng.uniform(0.0, 2.5, sel.sum()).astype(np.float32)  # A3
    has = (rng.random((nconf, mm)) > 0.1).astype(np.uint8)     # short reads (A5)
    mx.set_controls(flags, gain)
    sentinel = np.int16(12345)
    out = np.full(x.shape if conf_mode else (nconf, ns), sentinel, np.int16)
    out = mx.process(x, has, conf_mode, out=out)
    for c in range(nconf):
        ref, _ = _oracle_conf(oracle, x[c], flags[c], gain[c], has[c], conf_mode)
        if conf_mode:
            for m in range(mm):
                if flags[c, m] & O:
                    np.testing.assert_array_equal(out[c, m], ref[m], err_msg=f"conf {c} member {m}")
                else:  # disabled output rows are never written
                    assert (out[c, m] == sentinel).all()
        else:
            np.testing.assert_array_equal(out[c], ref)
    mx.close()


def test_mixer_saturation_is_symmetric(ctx, oracle):
    """A1: clamp is +-32767, even for a lone -32768 input."""
    mx = ms.MixerBatch(ctx, 1, 4, 160)
    x = np.zeros((1, 4, 160), np.int16)
    x[0, 0] = -32768
    x[0, 1] = -32768
    x[0, 2] = 32767
    x[0, 3] = 32767
    out = mx.process(x, None, 1)
    ref, _ = oracle.mixer_tick(x[0])
    np.testing.assert_array_equal(out[0], ref)
    assert out.min() >= -32767
    mx.close()


def test_mixer_split_form_equals_fused(ctx, oracle):
    """partial int32 sums of two member shards, added, then finalize ==
    the fused single-GPU kernel == the oracle (integer add is associative)."""
    torch = pytest.importorskip("torch")
    nconf, mm, ns = 6, 32, 480
    x = np.stack([[synth_pcm(c * mm + m, ns, sigma=8000.0) for m in range(mm)] for c in range(nconf)])
    fused = ms.MixerBatch(ctx, nconf, mm, ns)
    want = fused.process(x, None, 1)
    half = mm // 2
    shards = [np.ascontiguousarray(x[:, :half]), np.ascontiguousarray(x[:, half:])]
    sums, mixers, dins = [], [], []
    for sh in shards:
        m = ms.MixerBatch(ctx, nconf, half, ns)
        d = torch.from_numpy(sh).cuda()
        s = torch.zeros((nconf, ns), dtype=torch.int32, device="cuda")
        m.partial_sum(d, s)
        ctx.sync()
        sums.append(s)
        mixers.append(m)
        dins.append(d)
    total = (sums[0] + sums[1]).contiguous()  # stands in for the RCCL all-reduce
    torch.cuda.synchronize()
    for k in range(2):
        o = torch.zeros((nconf, half, ns), dtype=torch.int16, device="cuda")
        mixers[k].finalize(dins[k], total, o)
        ctx.sync()
        got = o.cpu().numpy()
        np.testing.assert_array_equal(got, want[:, k * half:(k + 1) * half])
    for c in range(nconf):
        ref, s = oracle.mixer_tick(x[c])
        np.testing.assert_array_equal(want[c], ref)
        np.testing.assert_array_equal(total[c].cpu().numpy(), s)


def test_mixer_full_size_config4_shard(ctx, oracle):
    """BASELINE config 4, one GPU's shard: 128 conferences x 32 members x 480 samples,
    device-resident; size-independent property: sum over members of (out + own) == members*sum
    where nothing saturates, plus oracle spot checks."""
    torch = pytest.importorskip("torch")
    nconf, mm, ns = 128, 32, 480
    rng = np.random.default_rng(7)
    x = rng.normal(0, 600, (nconf, mm, ns)).round().clip(-32767, 32767).astype(np.int16)
    mx = ms.MixerBatch(ctx, nconf, mm, ns)
    d = torch.from_numpy(x).cuda()
    o = mx.process(d, None, 1)
    ctx.sync()
    torch.cuda.synchronize()
    out = o.cpu().numpy().astype(np.int64)
    tot = x.astype(np.int64).sum(axis=1, keepdims=True)
    assert np.abs(tot).max() < 32767  # no saturation with sigma 600 * sqrt(32)
    np.testing.assert_array_equal(out + x, np.broadcast_to(tot, out.shape))
    for c in (0, 77, 127):
        ref, _ = oracle.mixer_tick(x[c])
        np.testing.assert_array_equal(out[c], ref)
    mx.close()
