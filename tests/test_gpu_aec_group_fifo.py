"""The small-frame FIFO entries' opt-in path (MSMI355X_AEC_GROUP_FIFO=1, csrc/aec.hip aec_fifos_group: the entry as its own
definition, the frames cancelled by aec_group_kernel) stays bit-equal to what the FIFO tests pin: the tests that exercise
mi_aec_process_fifos[_resampled] run again in a process with the switch on (it is read once per process).  Slower than the tick form
for whole ticks (profiles/r05_small_frame_fifo_entry.txt), which is why it is not the default."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fifo_entry_tests_pass_with_the_group_path_selected():
    env = dict(os.environ, MSMI355X_AEC_GROUP_FIFO="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_pipeline.py"), "-q", "-x", "-k",
                        "folded_in or wideband or stage_parity or starts_short", "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout
    env2 = dict(env, PLUGIN_BENCH_SHAPE="")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_plugin_fused.py"), "-q", "-x", "-k", "wideband_8k_16k or no_agc_ptime20_16k or no_resampler_16k",
                        "-p", "no:cacheprovider"], capture_output=True, text=True, timeout=900, env=env2, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:]
