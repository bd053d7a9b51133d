cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o headline -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-extras > gpurun_out/bench_prof_headline.json 2> gpurun_out/prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o all -- python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --pipeline-streams 0 > gpurun_out/bench_prof_all.json 2>> gpurun_out/prof.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -o pipeline -- python3 scripts/pipe_probe.py 65536 > gpurun_out/pipe_prof.json 2>> gpurun_out/prof.err
find gpurun_out/prof -name "*kernel_trace.csv" -delete
for f in $(find gpurun_out/prof -name "*kernel_stats.csv"); do echo "$f"; head -9 "$f" | cut -c1-170; done
python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json
