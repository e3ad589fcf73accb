#!/bin/bash
# round 3, GPU session 8: split ego / opponent layout of the combat kernel (tests), combat loop overhead interleaved vs split,
# de-phasing confirmation at N = 1e6 and 1e7
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s8; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_combat_parity.py tests/test_gpu_full_size.py -x -q -m gpu > $out/pytest.log 2>&1; tail -2 $out/pytest.log
for e in 12500 100000; do
  for lag in 0 1; do
    timeout 300 python bench.py --task combat --engagements $e --steps 200 --warmup 20 --opponent-lag $lag > $out/combat_split_e${e}_lag$lag.json 2> $out/err.log
    timeout 300 python bench.py --task combat --engagements $e --steps 200 --warmup 20 --opponent-lag $lag --interleaved --headline-only > $out/combat_inter_e${e}_lag$lag.json 2>> $out/err.log
  done
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r03_s8/combat_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'ms/step %.4f  plain %.4f  kernel %.4f  overhead %.4f' % (d['ms_per_step'], d['ms_per_step_without_timing_events'], d['roofline']['kernel_avg_ms'], d['loop_overhead_ms']), d.get('expected_scaling') and {k:d['expected_scaling'][k] for k in ('kernel_ms_at_share','loop_ms_at_share_one_gpu','predicted_speedup_8_gpus_lag0','predicted_speedup_8_gpus_lag1')})
    except Exception as e: print(f, 'ERR', e)
PY
timeout 900 python tools/microbench/ab_libs.py --rounds 3 --steps 100 e_head e_7x7000 e_9x7000 > $out/ab_stagger_1e6.log 2>&1; grep -v Warn $out/ab_stagger_1e6.log | tail -4
timeout 600 python tools/microbench/ab_libs.py --rounds 2 --steps 20 --n 10000000 e_head e_7x7000 e_9x7000 > $out/ab_stagger_1e7.log 2>&1; grep -v Warn $out/ab_stagger_1e7.log | tail -4
