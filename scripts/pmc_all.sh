#!/bin/bash
# Runs on the GPU box: HBM-side byte counters per kernel, FETCH_SIZE and WRITE_SIZE in separate passes
# (TCC slots, MI355X guide), bench-sized workloads, plain launches.  Summarised by scripts/pmc_summarize.py.
set -u
for k in resample mixer volume equalizer aec scaler pixconv g711dec g711enc plc; do
	bash scripts/pmc_run.sh "pmc_${k}_fetch" "FETCH_SIZE" python3 scripts/pmc_probe.py $k > /dev/null
	bash scripts/pmc_run.sh "pmc_${k}_write" "WRITE_SIZE" python3 scripts/pmc_probe.py $k > /dev/null
done
python3 scripts/pmc_summarize.py gpurun_out
