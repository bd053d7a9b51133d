"""Cuts the fixture pair tests/golden/resample_wav/voice_{16000,48000}_6s.wav out of the reference's
tester/sounds/test_silence_voice_{16000,48000}.wav (the SAME recording shipped at two rates): seconds 3..9 of
each, samples untouched.  Run in the build container (the reference tree is not on the GPU box)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import audiodiff as ad  # noqa: E402

SRC = "/root/reference/tester/sounds"
os.makedirs(os.path.join(HERE, "resample_wav"), exist_ok=True)
for rate in (16000, 48000):
    r, ch, x = ad.read_wav(os.path.join(SRC, f"test_silence_voice_{rate}.wav"))
    assert (r, ch) == (rate, 1)
    cut = x[3 * rate: 9 * rate + (600 if rate == 48000 else 0)]  # slack for the resampler's delay
    ad.write_wav(os.path.join(HERE, "resample_wav", f"voice_{rate}_6s.wav"), rate, np.ascontiguousarray(cut))
    print(rate, len(cut))

# the other three rates the reference ships the recording at: seconds 3..6 (+ slack), for the pairs that exercise
# every resampler kernel family (x6 / x2 up, 3/2 and 2/3, 160/147 interpolated, /2 /3 /6 down)
for rate in (8000, 32000, 44100):
    r, ch, x = ad.read_wav(os.path.join(SRC, f"test_silence_voice_{rate}.wav"))
    assert (r, ch) == (rate, 1)
    cut = x[3 * rate: 6 * rate + rate // 80]
    ad.write_wav(os.path.join(HERE, "resample_wav", f"voice_{rate}_3s.wav"), rate, np.ascontiguousarray(cut))
    print(rate, len(cut))
