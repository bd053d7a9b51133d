#!/bin/bash
# quick GPU iteration: run the given pytest selection
set -u
mkdir -p gpurun_out
timeout 900 python -m pytest "$@" -q -x 2>&1 | tail -50 | tee gpurun_out/quick.log
