#!/bin/bash
python -m pytest tests/test_gpu_aec.py tests/test_gpu_pipeline.py tests/test_gpu_plugin.py tests/test_gpu_plugin_fused.py -q -x 2>&1 | grep -v "ms2shim-warning" | tail -15 > gpurun_out/r06_tests_c.txt
cat gpurun_out/r06_tests_c.txt
scripts/r06_attach_probe.sh b
