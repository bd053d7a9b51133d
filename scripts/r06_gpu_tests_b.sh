#!/bin/bash
# round 6: every plugin test on the GPU + the AudioStream shape's cost and its first ticks
python -m pytest tests/test_gpu_plugin.py tests/test_gpu_plugin_fused.py tests/test_gpu_plugin_codec.py tests/test_gpu_plugin_server.py tests/test_gpu_plugin_conference.py tests/test_gpu_plugin_video.py -q -x 2>&1 | grep -v "ms2shim-warning" | tail -40 > gpurun_out/r06_tests_b.txt
cat gpurun_out/r06_tests_b.txt
scripts/r06_astream_probe.sh attach1
scripts/r06_astream_probe.sh plain1 32768 ""
