#!/bin/bash
# generation quantisation of the pair variant (1 536 resident workgroups of 128 aircraft): whole generations vs a few workgroups more
cd $GRAFT_REPO_ROOT
for n in 983040 1000000 1179648 786432 800000; do python bench.py --headline-only --no-cpu-baseline --steps 100 --warmup 5 --n $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('N=$n', 'wgs', ($n+127)//128, 'gens %.2f'%((($n+127)//128)/1536), 'kernel_ms', round(r['kernel_avg_ms'],4), 'median', round(r['kernel_median_ms'],4), 'ns/aircraft %.4f'%(1e6*r['kernel_median_ms']/$n))"; done
