#!/bin/bash
# round 6: the AudioStream endpoint's parity tests on the GPU + the shape's cost
python -m pytest tests/test_gpu_plugin_fused.py tests/test_gpu_plugin_codec.py -q -k "audiostream or default or plc or g711 or astream or no_mixer" 2>&1 | grep -v "ms2shim-warning" | tail -60 > gpurun_out/r06_tests_a.txt
cat gpurun_out/r06_tests_a.txt
