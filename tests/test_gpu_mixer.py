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
        torch.cuda.synchronize()  # torch fills `d` and `s` on ITS stream; the kernels below run on the context's stream
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


@pytest.mark.parametrize("mm,ns,rate", [(32, 480, 48000), (50, 160, 16000), (3, 80, 8000)])
def test_volume_and_conference_mix_in_one_launch_equal_the_two_launches(ctx, mm, ns, rate):
    """mi_mixer_process_volume_fifo (every pin's chunk popped from its FIFO, metered and levelled, the conference mixed from
    the levelled chunks, nothing in between written to HBM) == mi_volume_process_fifo + mi_mixer_process: mixed samples,
    the meters' whole state incl. the one-second maximum, FIFO levels -- with AGC, noise gate, DC removal, an echo-limiter
    pair (when the conferences cover the whole batch), muted / listen-only / unplumbed pins, pin gains, legs that run dry and ring heads off the 16-byte grid; the
    conferences start at stream `first` of a larger volume batch whose other streams are served by the ranged volume call."""
    import mediastreamer2_amd as ms
    torch = pytest.importorskip("torch")
    nconf, first, extra = (5, 8, 3) if mm != 3 else (5, 0, 0)   # the small case: conferences = the whole batch, with peers
    n = first + nconf * mm + extra
    rng = np.random.default_rng(mm)
    L, A, O = ms.MI_MIX_LINKED, ms.MI_MIX_ACTIVE, ms.MI_MIX_OUTPUT
    flags = np.full((nconf, mm), L | A | O, np.uint8)
    flags[0, 1] = L | O          # muted
    flags[1, 0] = L | A          # no return audio
    flags[2, 2] = 0              # unplumbed
    gain = np.ones((nconf, mm), np.float32)
    gain[3, 1] = 0.5
    gain[4, 2] = 1.7

    def rig():
        v = ms.VolumeBatch(ctx, n, rate)
        ps = []
        for s in range(n):
            p = v.default_params()
            p.agc_enabled = int(s % 3 == 0)
            p.noise_gate_enabled = int(s % 5 == 1)
            p.remove_dc = int(s % 7 == 2)
            if s % 4 == 1:
                p.static_gain = 0.5
            ps.append(p)
        if extra == 0:  # an echo-limiter pair inside conference 0 (peers need the batch processed whole)
            ps[first + 1].peer = first + 2
            ps[first + 2].peer = first + 1
        v.set_params(ps)
        st = v.get_state()
        for s in range(n):
            st[s].gain = st[s].target_gain = ps[s].static_gain
        v.set_state(st)
        m = ms.MixerBatch(ctx, nconf, mm, ns)
        m.set_controls(flags, gain)
        f = ms.FifoBatch(ctx, n, 4 * ns + 64)
        return v, m, f

    (v1, m1, f1), (v2, m2, f2) = rig(), rig()
    z = lambda *sh, dt=torch.int16: torch.zeros(sh, dtype=dt, device="cuda")
    pre, junk = torch.from_numpy(rng.integers(-9000, 9000, (n, 8), dtype=np.int16)).cuda(), z(n, 8)
    odd = torch.from_numpy((np.arange(n) % 3 == 1).astype(np.uint8)).cuda()
    torch.cuda.synchronize()
    for f in (f1, f2):  # a third of the rings get a head that is no multiple of 8
        f.push(pre, nsamples=8)
        f.pop(3, junk, gate=odd)
    t1, o1, o2 = z(n, ns), z(nconf, mm, ns), z(nconf, mm, ns)
    t2 = z(n, ns)
    lv1, lv2 = z(n, dt=torch.int32), z(n, dt=torch.int32)
    for t in range(30):
        blk = rng.integers(-20000, 20000, (n, ns), dtype=np.int16)
        blk[:, ::7] += 900  # some DC
        cnt = rng.choice([0, ns], n, p=[0.15, 0.85]).astype(np.int32)  # now and then a leg delivers nothing: it runs dry
        d, c = torch.from_numpy(blk).cuda(), torch.from_numpy(cnt).cuda()
        torch.cuda.synchronize()
        f1.push(d, nsamples=ns, count=c)
        f2.push(d, nsamples=ns, count=c)
        v1.process_fifo(f1, t1)
        m1.process(t1[first:first + nconf * mm].view(nconf, mm, ns), out=o1)
        m2.process_volume_fifo(v2, f2, o2, first_stream=first)
        if extra:
            v2.process_fifo(f2, t2, first=0, count=first)                    # the streams around the conferences
            v2.process_fifo(f2, t2, first=first + nconf * mm, count=extra)
        f1.levels(lv1)
        f2.levels(lv2)
        ctx.sync()
        np.testing.assert_array_equal(o1.cpu().numpy(), o2.cpu().numpy(), err_msg=f"tick {t}")
        np.testing.assert_array_equal(lv1.cpu().numpy(), lv2.cpu().numpy())
        a, b = t1.cpu().numpy(), t2.cpu().numpy()
        np.testing.assert_array_equal(a[:first], b[:first])
        np.testing.assert_array_equal(a[first + nconf * mm:], b[first + nconf * mm:])
        s1, s2 = v1.get_state(), v2.get_state()
        for s in range(n):
            assert bytes(s1[s]) == bytes(s2[s]), f"tick {t} stream {s}"
        np.testing.assert_array_equal(v1.get_max().view(np.uint32), v2.get_max().view(np.uint32))
    assert o1.cpu().numpy().any() and not o2.cpu().numpy()[2, 2].any() and not o2.cpu().numpy()[1, 0].any()
    for o in (v1, m1, f1, v2, m2, f2):
        o.close()


def test_pinned_copies_as_kernels_move_every_byte(ctx):
    """mi_copy_h2d_pinned / mi_copy_d2h_pinned (the plugin's copies: kernels of this library on the context's stream instead of
    the runtime's copy path) at every alignment class the facades produce: 16-byte, 4-byte and odd byte counts and offsets."""
    import ctypes as C
    torch = pytest.importorskip("torch")
    L = ctx.L
    n = 1 << 16
    hp = L.mi_host_alloc(ctx.h, n + 64)
    hq = L.mi_host_alloc(ctx.h, n + 64)
    assert hp and hq
    src = (C.c_uint8 * (n + 64)).from_address(hp)
    dst = (C.c_uint8 * (n + 64)).from_address(hq)
    pattern = np.random.default_rng(5).integers(0, 256, n + 64, dtype=np.uint8)
    np.frombuffer(src, np.uint8)[:] = pattern
    dev = torch.zeros(n + 64, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    for off_h, off_d, nbytes in ((0, 0, n), (16, 32, 4096), (4, 8, 1000), (1, 3, 999), (2, 0, 2), (0, 0, 16), (7, 5, 1)):
        dev.zero_()
        np.frombuffer(dst, np.uint8)[:] = 0
        torch.cuda.synchronize()
        assert L.mi_copy_h2d_pinned(ctx.h, dev.data_ptr() + off_d, hp + off_h, nbytes) == 0
        assert L.mi_copy_d2h_pinned(ctx.h, hq + off_h, dev.data_ptr() + off_d, nbytes) == 0
        ctx.sync()
        got = np.frombuffer(dst, np.uint8)
        np.testing.assert_array_equal(got[off_h:off_h + nbytes], pattern[off_h:off_h + nbytes], err_msg=str((off_h, off_d, nbytes)))
        assert not got[:off_h].any() and not got[off_h + nbytes:].any(), "bytes outside the range were written"
        d = dev.cpu().numpy()
        assert not d[:off_d].any() and not d[off_d + nbytes:].any()
    L.mi_host_free(ctx.h, hp)
    L.mi_host_free(ctx.h, hq)
