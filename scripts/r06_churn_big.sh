# PLUGIN_BENCH_CHURN with MSMI355X_CHECK_LEVELS at the bench's scale: 32 768 legs on 16 tickers, paced, 20 re-plumbings a second per ticker
make -C tests/host plugin_bench >/dev/null 2>&1
for rep in 1 2; do for sh in "" "server dec" "astream default"; do
PLUGIN_BENCH_CHURN=20 PLUGIN_BENCH_PACED=1 PLUGIN_BENCH_SHAPE="$sh" MSMI355X_CHECK_LEVELS=1 tests/host/plugin_bench mediastreamer2_amd/libmsmi355xfilters.so 32768 16 400 40 2> /tmp/e.txt | tail -1 > /tmp/o.json
python3 -c "
import json
d=json.loads(open('/tmp/o.json').read())
print('rep $rep shape \'$sh\'', {k: d[k] for k in ('legs','fused_legs','late_events','p50_ms','p99_ms','max_ms','late') if k in d}, d.get('churn'))"
grep -a "error" /tmp/e.txt | sed "s/0x[0-9a-f]*/PTR/g;s/leg [0-9]*/leg N/" | cut -c1-200 | sort | uniq -c | sort -rn | head -4
done; done
