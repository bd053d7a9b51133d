"""The nine scenarios of the reference's echo-canceller tester (tester/mediastreamer2_aec3_tester.c:601-812), as data:
which recordings play where, the delay between the far end and the microphone side, the analysis windows and the
thresholds the tester asserts.  The recordings are the tester's own (tests/golden/aec_wav/, copied as fixtures).

The tester drives MSWebRTCAEC (an out-of-tree plugin); here the same scenes grade MSSpeexEC, the in-tree canceller
this repository replaces.  Two things follow from that and are stated wherever a number is reported:
  * the speex canceller conditions its microphone input with a DC notch (filter_dc_notch16, radius .982 at 16 kHz /
    .992 at 48 kHz), which this LF-heavy material feels: similarity is reported BOTH against the raw near-end file
    (the tester's metric as is) and against the near-end file passed through that notch;
  * MS_ECHO_CANCELLER_GET_DELAY returns the configured value for MSSpeexEC (speexec.c:340-344), so the tester's
    estimated-delay assertions have no counterpart; a delay beyond the canceller's tail (470 ms against speexec.c:82's
    250 ms) is handed to the filter with MS_ECHO_CANCELLER_SET_DELAY, as a calibrated linphone configuration does.
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
WAV = os.path.join(HERE, "golden", "aec_wav")
FILE_RATE = 16000  # every recording of the tester is 16 kHz mono

# name: near, far, echo, noise, delay_ms, ec rate, (start_short, stop_short, start) ms, expected final delay for the
# alignment search, similarity threshold, energy threshold (aec3_tester.c, line of the set_audio_analysis_param call)
SCENARIOS = {
    "simple_talk": dict(near="nearend_simple_talk", far="farend_simple_talk", echo="echo_simple_talk", noise=None,
                        delay=100, rate=16000, win=(12500, 14500, 11000), sim=0.99, energy=1.0, line=721),
    "double_talk": dict(near="nearend_double_talk", far="farend_double_talk", echo="echo_double_talk", noise=None,
                        delay=100, rate=16000, win=(11500, 13500, 9500), sim=0.83, energy=1.0, line=688),
    "simple_talk_white_noise": dict(near="nearend_simple_talk", far="farend_simple_talk", echo="echo_simple_talk",
                                    noise="white_noise", delay=100, rate=16000, win=(12500, 14500, 11000), sim=0.98,
                                    energy=4.0, line=738),
    "double_talk_white_noise": dict(near="nearend_double_talk", far="farend_double_talk", echo="echo_double_talk",
                                    noise="white_noise", delay=100, rate=16000, win=(11500, 13500, 9500), sim=0.90,
                                    energy=3.0, line=705),
    "near_end_single_talk": dict(near="nearend_double_talk", far=None, echo=None, noise=None, delay=0, rate=16000,
                                 win=(2000, 4000, 0), sim=0.99, energy=1.0, line=615),
    "far_end_single_talk": dict(near=None, far="farend_double_talk", echo="echo_double_talk", noise=None, delay=100,
                                rate=16000, win=None, sim=None, energy=3.0, line=649),  # energy of the WHOLE output
    "simple_talk_48000Hz": dict(near="nearend_simple_talk", far="farend_simple_talk", echo="echo_simple_talk",
                                noise=None, delay=100, rate=48000, win=(12500, 14500, 11000), sim=0.98, energy=1.0,
                                line=755),
    "simple_talk_with_delay_change": dict(near="nearend_simple_talk", far="farend_simple_talk",
                                          echo="echo_delay_change", noise=None, delay=100, final_delay=150, rate=16000,
                                          win=(12500, 14500, 11000), sim=0.99, energy=1.0, line=806),
}
for _d in (0, 40, 80, 200, 470):  # simple_talks_with_several_delays, aec3_tester.c:760-794
    SCENARIOS[f"simple_talk_delay_{_d}ms"] = dict(near="nearend_simple_talk", far="farend_simple_talk",
                                                  echo="echo_simple_talk", noise=None, delay=_d, rate=16000,
                                                  win=(12500, 14500, 11000), sim=0.99,
                                                  energy=3.3 if _d > 400 else 1.0, line=788,
                                                  set_delay=_d - 40 if _d > 250 else 0)

TAIL_MS = 250  # speexec.c:82 default, the tester sets none


def wav(name):
    from oracle import audiodiff as ad
    rate, nch, x = ad.read_wav(os.path.join(WAV, name + ".wav"))
    assert rate == FILE_RATE and nch == 1
    return x


def sat_mix(*xs):
    """MSAudioMixer, non-conference mode: int32 sum, symmetric saturation (audiomixer.c:33-44,:210-217)."""
    n = max(len(x) for x in xs)
    acc = np.zeros(n, np.int32)
    for x in xs:
        acc[:len(x)] += x
    return np.clip(acc, -32767, 32767).astype(np.int16)


def scene(name):
    """What the tester's graph delivers at the FILE rate, sample-aligned with the far-end player's start:
    (near-end file or None, far-end track, microphone track = near + echo (+ looped noise), each started delay_ms after
    the far end (aec3_tester.c:487-505)).  Length = the longest track, as the tester waits for the longest player."""
    sc = SCENARIOS[name]
    d = sc["delay"] * FILE_RATE // 1000
    near = wav(sc["near"]) if sc["near"] else None
    far = wav(sc["far"]) if sc["far"] else None
    echo = wav(sc["echo"]) if sc["echo"] else None
    n = max([len(x) + (0 if x is far else d) for x in (near, far, echo) if x is not None])
    n = (n + 159) // 160 * 160
    pad = lambda x, lead: np.concatenate([np.zeros(lead, np.int16), x, np.zeros(n - lead - len(x), np.int16)])
    tracks = []
    if near is not None:
        tracks.append(pad(near, d))
    if echo is not None:
        tracks.append(pad(echo, d))
    if sc["noise"]:
        nz = wav(sc["noise"])
        tracks.append(np.tile(nz, n // len(nz) + 1)[:n])  # MS_FILE_PLAYER_LOOP 0: started with the far end
    mic = sat_mix(*tracks, np.zeros(n, np.int16))
    ref = pad(far, 0) if far is not None else np.zeros(n, np.int16)
    return near, ref, mic


def notch(x, rate=16000):
    """filter_dc_notch16 of the canceller's input stage as a transfer function (speex mdf.c), at the EC's rate radius."""
    from scipy.signal import lfilter
    radius = .9 if rate < 12000 else (.982 if rate < 24000 else .992)
    den2 = radius * radius + .7 * (1 - radius) * (1 - radius)
    y = lfilter([radius, -2 * radius, radius], [1, -2 * radius, den2], x.astype(np.float64))
    return np.clip(np.round(y), -32768, 32767).astype(np.int16)


def grade(name, near_ref, out):
    """ms_audio_compare_silence_and_speech with the tester's windows and max shift (aec3_tester.c:112-121)."""
    from oracle import audiodiff as ad
    sc = SCENARIOS[name]
    a, b, c = sc["win"]
    dly = sc.get("final_delay", sc["delay"])
    msp = 1 if dly == 0 else int(dly * 1.5 / (b - a) * 100)
    n = min(len(near_ref), len(out))
    return ad.compare_silence_and_speech(near_ref[:n] if len(near_ref) >= len(out) else near_ref, out, FILE_RATE, a, b, c, msp)


def report(name, near, out, ec_rate=16000):
    """(similarity vs the raw near-end file, similarity vs the notch-conditioned one, energy in silence / of the file)"""
    from oracle import audiodiff as ad
    sc = SCENARIOS[name]
    if sc["win"] is None:
        return None, None, ad.audio_energy(out)
    sim_raw, energy, _ = grade(name, near, out)
    sim_notch, _, _ = grade(name, notch(near, ec_rate), out)
    return sim_raw, sim_notch, energy
