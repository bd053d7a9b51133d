make -C tests/host plugin_bench >/dev/null 2>&1
for rep in 1 2 3 4 5 6; do for sh in "" "server dec" "astream default"; do
PLUGIN_BENCH_CHURN=50 PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SHAPE="$sh" MSMI355X_CHECK_LEVELS=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 4096 4 200 20 2> /tmp/e.txt | tail -1 > /tmp/o.json
L=$(python3 -c "
import json
d=json.loads(open('/tmp/o.json').read()); print(d['late_events'])")
echo "rep $rep shape '$sh' late_events $L"
if [ "$L" != "0" ]; then grep -a "error" /tmp/e.txt | sed "s/0x[0-9a-f]*/PTR/g" | cut -c1-220 | sort | uniq -c | sort -rn | head -8; grep -a "error" /tmp/e.txt | head -3 | cut -c1-250; fi
done; done
