#!/bin/bash
# round 3, GPU session 1: baseline of the mid-size trough per variant, one-generation de-phasing experiment, counter calibration,
# per-workgroup timelines at the mid sizes, the driver's command on the round-2 tree
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s1; mkdir -p $out
L=tools/microbench/libs
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --headline-only > $out/bench_driver_cmd.json 2> $out/bench.err < /dev/null
NPF16_LIB=$L/a_base.so timeout 600 python tools/microbench/mid_n.py --variants latency8,latency,pair,throughput --out $out/mid_base.json > $out/mid_base.log 2>&1
NPF16_LIB=$L/a_base.so NPF16_PAIR_WAVES=3 timeout 300 python tools/microbench/mid_n.py --variants pair --out $out/mid_base_pw3.json > $out/mid_base_pw3.log 2>&1
for lib in st0_3x4000 st0_3x8000 st0_5x3000; do
  NPF16_LIB=$L/$lib.so NPF16_PAIR_WAVES=2 timeout 300 python tools/microbench/mid_n.py --variants pair --out $out/mid_$lib.json 49152 65536 81920 100000 131072 > $out/mid_$lib.log 2>&1
done
for n in 65536 100000 131072; do
  timeout 100 python tools/wg_timeline.py --variant pair --n $n --out $out/wg_timeline_pair_n$n.json > $out/wg_timeline_n$n.log 2>&1
done
bash tools/calibrate_counters.sh 4194304 > $out/calib.log 2>&1
ls -la $out
