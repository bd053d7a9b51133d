"""oracle/audiodiff.py -- TEST INFRASTRUCTURE ONLY (see ms2_oracle.h).

numpy restatement of the reference's similarity / energy metrics, src/utils/audiodiff.c, the functions the
AEC tester grades recordings with (tester/mediastreamer2_aec3_tester.c:596-760).  Mono only.
All correlations are exact integers (< 2^53) carried in float64.
"""
import wave

import numpy as np


def read_wav(path):
    """PCM16 WAV -> (rate, nchannels, int16 array [nsamples, nchannels] squeezed for mono)."""
    with wave.open(path, "rb") as w:
        assert w.getsampwidth() == 2
        rate, nch, n = w.getframerate(), w.getnchannels(), w.getnframes()
        x = np.frombuffer(w.readframes(n), dtype="<i2").astype(np.int16)
    return rate, nch, (x.reshape(-1, nch) if nch > 1 else x)


def write_wav(path, rate, x):
    x = np.ascontiguousarray(x, dtype="<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1 if x.ndim == 1 else x.shape[1])
        w.setsampwidth(2)
        w.setframerate(rate)
        w.writeframes(x.tobytes())


def audio_energy(x):
    """ms_audio_energy audiodiff.c:653-682: sum of (s/32768)^2."""
    s = np.asarray(x, np.float64) / 32768.0
    return float(np.sum(s * s))


def cross_correlation(s1, s2_padded, nshifts):
    """compute_cross_correlation audiodiff.c:184-216 (step 1).  s2_padded holds len(s1)+nshifts samples at least.
    Returns (xcorr[nshifts] float32, argmax of |numerator|)."""
    s1 = np.asarray(s1, np.float64)
    s2 = np.asarray(s2_padded, np.float64)
    n1 = len(s1)
    need = n1 + nshifts - 1
    if len(s2) < need:
        s2 = np.concatenate([s2, np.zeros(need - len(s2))])
    num = np.correlate(s2[:need], s1, mode="valid")           # num[i] = <s1, s2[i:i+n1]>
    sq = np.concatenate([[0.0], np.cumsum(s2[:need] ** 2)])
    norm2 = sq[n1:n1 + nshifts] - sq[:nshifts]                # energy of s2[i:i+n1]
    norm1 = float(np.sum(s1 * s1))
    den = np.sqrt(norm1 * norm2)
    xc = np.where(den > 0, num / np.where(den > 0, den, 1.0), 1.0).astype(np.float32)
    return xc, int(np.argmax(np.abs(num)))   # first maximum, like the strict '>' of :205


def diff_one_chunk(s1, s2_padded, max_shift):
    """_ms_audio_diff_one_chunk audiodiff.c:218-287, mono branch: (position, similarity)."""
    xc, idx = cross_correlation(s1, s2_padded, 2 * max_shift)
    return idx - max_shift, float(xc[idx])


def silence_mask_and_energy(s1, s2):
    """ms_audio_compute_energy_in_silence audiodiff.c:349-407 (window sizes tuned for 16 kHz there)."""
    n = len(s1)
    a = np.abs(np.asarray(s1, np.int32)).astype(np.float64) / 32768.0

    def moving_mean(v, half):
        c = np.concatenate([[0.0], np.cumsum(v)])
        lo = np.maximum(0, np.arange(n) - half)
        hi = np.minimum(n, np.arange(n) + half + 1)
        return (c[hi] - c[lo]) / (hi - lo)

    mask = (moving_mean(a, 200) < 0.001).astype(np.float64)
    mask = (moving_mean(mask, 1400) >= 0.5)
    s = np.asarray(s2[:n], np.float64) / 32768.0
    return mask, float(np.sum((s * s)[mask]))


def similarity_in_speech(s1, s2, mask):
    """ms_audio_compute_similarity_in_speech audiodiff.c:413-440."""
    keep = ~mask
    n_speech = int(keep.sum())
    max_shift = int(n_speech / 100.0)
    a = np.asarray(s1)[: len(mask)][keep]
    b = np.concatenate([np.zeros(max_shift), np.asarray(s2, np.float64)[: len(mask)][keep], np.zeros(max_shift)])
    return diff_one_chunk(a, b, max_shift)


def compare_silence_and_speech(ref, out, rate, start_short_ms, stop_short_ms, start_ms, max_shift_percent):
    """ms_audio_compare_silence_and_speech audiodiff.c:442-576 on two mono int16 arrays of one rate.
    Returns (similarity_in_speech, energy_in_silence, alignment_position)."""
    tested = stop_short_ms - start_short_ms
    assert tested < len(ref) / rate * 1000 and tested < len(out) / rate * 1000
    max_shift = tested * rate // 1000 * min(max(1, max_shift_percent), 100) // 100
    start = int(start_short_ms / 1000.0 * rate)
    size = int(tested / 1000.0 * rate)
    seg_out = np.asarray(out[start:start + size])                   # fi2, no padding
    seg_ref = np.concatenate([np.zeros(max_shift), np.asarray(ref[start:start + size], np.float64), np.zeros(max_shift)])
    maxpos, _ = diff_one_chunk(seg_out, seg_ref, max_shift)          # (fi2->buffer, fi1->buffer, ...) :523
    pad1, pad2 = (-maxpos, 0) if maxpos < 0 else (0, maxpos)
    s0 = int(start_ms / 1000.0 * rate)
    n1, n2 = len(ref) - s0, len(out) - s0
    # file_info_read_short(fi, zero_pad, start, size): zero_pad zeros, then the samples
    r = np.concatenate([np.zeros(pad1, np.int16), np.asarray(ref[s0:], np.int16)])
    o = np.concatenate([np.zeros(pad2, np.int16), np.asarray(out[s0:], np.int16)])
    n = min(n1, n2)
    mask, energy = silence_mask_and_energy(r[:n], o[:n])
    _, sim = similarity_in_speech(r[:n], o[:n], mask)
    return sim, energy, maxpos
