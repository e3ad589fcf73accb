#!/bin/bash
# round 3, GPU session 9: aero_1d_tables on the pair variant — parity (every table-mode test), timing vs the MLP numerics and vs the
# round-2 table kernel; full GPU suite; de-phasing confirmation
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s9; mkdir -p $out
timeout 1500 python -m pytest tests/test_gpu_step_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_aero_grid.py -x -q -m gpu -k "aero_1d_tables or tables or pwl or grid" > $out/pytest_tables.log 2>&1; tail -3 $out/pytest_tables.log
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
for lib in a_base e_head; do
  NPF16_LIB=tools/microbench/libs/$lib.so timeout 300 python bench.py --headline-only --steps 100 --warmup 5 --aero-1d-tables 1 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib tables', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['kernel_median_ms'])"
done
timeout 900 python tools/microbench/ab_libs.py --rounds 3 --steps 100 e_head e_7x7000 e_9x7000 > $out/ab_stagger_1e6.log 2>&1; grep -v Warn $out/ab_stagger_1e6.log | tail -4
timeout 600 python tools/microbench/ab_libs.py --rounds 2 --steps 20 --n 10000000 e_head e_7x7000 e_9x7000 > $out/ab_stagger_1e7.log 2>&1; grep -v Warn $out/ab_stagger_1e7.log | tail -4
