#!/bin/bash
# round 4, GPU session 12: the self-validating coefficient cache (keys = the (alpha, beta) the cached coefficients belong to; the step kernels
# re-evaluate in the waves that find a difference) — its test first, the whole GPU suite, the driver's bench command (headline regression <= 0.5 %?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s12; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_step_parity.py -x -q -m gpu -k "validates_itself or invalidated_by" > $out/t_cache.log 2>&1; echo "cache tests rc=$?"; tail -3 $out/t_cache.log; grep -B5 -A30 "Error\|FAILED" $out/t_cache.log | head -80
timeout 2400 python -m pytest tests -x -q -m gpu > $out/gputest.log 2>&1; echo "gpu suite rc=$?"; tail -3 $out/gputest.log; grep -B5 -A25 "Error\|FAILED" $out/gputest.log | head -80
for i in 1 2; do timeout 900 python bench.py --headline-only > $out/bench_headline_$i.json 2> $out/bench.err; python - <<PY
import json
d = json.loads(open('gpurun_out/r04_s12/bench_headline_$i.json').read().strip().splitlines()[-1])
print('headline', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('kernel_avg_ms'))
PY
done
timeout 900 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open('gpurun_out/r04_s12/bench.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('value', 'ms_per_step')}, d['roofline']['frac'])
for k, v in d.get('optional_modes', {}).items():
    if isinstance(v, dict) and 'value' in v: print(k, v['value'], v.get('roofline', {}).get('frac'))
PY
