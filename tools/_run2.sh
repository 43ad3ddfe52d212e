cd $GRAFT_REPO_ROOT
python bench.py > gpurun_out/r04_bench_a.json 2> gpurun_out/r04_bench_a.err
tail -c 1500 gpurun_out/r04_bench_a.err
python - <<'PY'
import json
l=json.loads(open('gpurun_out/r04_bench_a.json').read().strip().splitlines()[-1])
print({k:l[k] for k in ('value','value_metric_partition','ms_per_step','success_count')})
print(l['roofline']['frac'], l['roofline']['end_to_end_frac'])
print(json.dumps(l['other_configs'], indent=1)[:3000])
print([ (s['batch_per_gpu'], round(s['ms_per_step'],3)) for s in l['shard_points']])
PY
