#!/bin/bash
# round 3, GPU session 3: full GPU suite on the tree with the state / trigonometry pinned before the Overload phase (pair kernel 188 ->
# 159 VGPRs, no scratch), A/B against the round-2 kernel in one session, mid sizes, PMC instruction count
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s3; mkdir -p $out
timeout 1500 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 900 python tools/microbench/ab_libs.py --rounds 3 --steps 100 a_base b_pin > $out/ab_1e6.log 2>&1; tail -4 $out/ab_1e6.log
for pw in 2 3; do NPF16_PAIR_WAVES=$pw timeout 300 python tools/microbench/mid_n.py --variants pair --out $out/mid_pair_pw$pw.json 65536 100000 131072 196608 262144 400000 > $out/mid_pair_pw$pw.log 2>&1; done
timeout 300 python tools/microbench/ab_libs.py --rounds 2 --steps 20 --n 10000000 a_base b_pin > $out/ab_1e7.log 2>&1; tail -3 $out/ab_1e7.log
timeout 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $out/pmc_valu -o p -- python bench.py --steps 3 --warmup 2 --prelude-ms 0 --headline-only > $out/pmc_valu.log 2>&1 < /dev/null
ls $out
