#!/bin/bash
# Collect the round's evidence on the GPU box: bench JSON (the driver's command and a long run), rocprofv3 kernel stats, PMC passes
# (separate runs, --kernel-trace only next to --pmc), the per-workgroup timeline.   usage (via gpurun): bash tools/profile_round.sh r02b
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err < /dev/null
timeout 300 python bench.py --steps 200 --warmup 20 --headline-only > $out/bench.json 2> $out/bench.err < /dev/null
timeout 120 python tools/wg_timeline.py --n 1000000 --out $out/wg_timeline_n1e6.json > /dev/null 2> $out/timeline.err < /dev/null
timeout 400 python tools/microbench/mid_n.py --variants auto --out $out/n_sweep.json 3000 10000 30000 49152 65536 81920 98304 100000 131072 196608 262144 > $out/n_sweep.log 2>&1 < /dev/null
timeout 300 python bench.py --task combat --engagements 12500 --steps 200 --warmup 20 > $out/bench_combat_e12500.json 2>> $out/bench.err < /dev/null
timeout 300 python bench.py --task combat --engagements 100000 --steps 100 --warmup 10 > $out/bench_combat_e1e5.json 2>> $out/bench.err < /dev/null
timeout 300 python bench.py --task tracking --steps 20 --warmup 5 --headline-only > $out/bench_tracking.json 2>> $out/bench.err < /dev/null
timeout 300 python bench.py --task control --steps 20 --warmup 5 --headline-only > $out/bench_control.json 2>> $out/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python bench.py --gpus 1 --steps 20 --warmup 5 --headline-only > $out/stats.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_combat -o p -- python tools/microbench/combat_bench.py 100000 > $out/stats_combat.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_actor -o p -- python tools/microbench/actor_bench.py 262144 > $out/stats_actor.log 2>&1 < /dev/null
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_REQ" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -d" " -f1)
  timeout 250 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_$t -o p -- python bench.py --steps 3 --warmup 2 --prelude-ms 0 --headline-only > $out/pmc_$t.log 2>&1 < /dev/null
done
# PlanningEnv (the persistent kernel, n = 8 192 and the reference's training size 1e4), block-fixed-point controller and the fp32 one
for nm in i8 fp32; do
  NUMERICS=$nm timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_planning_$nm -o p -- python tools/microbench/planning_profile.py 8192 40 > $out/stats_planning_$nm.log 2>&1 < /dev/null
done
NUMERICS=i8 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_planning_i8_n1e4 -o p -- python tools/microbench/planning_profile.py 10000 40 > $out/stats_planning_i8_n1e4.log 2>&1 < /dev/null
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_LDS"; do
  t=$(echo $set | cut -d" " -f1)
  for nm in i8 fp32; do
    NUMERICS=$nm timeout 250 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/planning_pmc_${nm}_$t -o p -- python tools/microbench/planning_profile.py 8192 10 > $out/planning_pmc_${nm}_$t.log 2>&1 < /dev/null
  done
done
# the collect loop with the policy's inference step as one launch (N1): which kernels a collect step is made of
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_collect -o p -- python tools/collect_loop.py --only fused --n 3000 --steps 300 > $out/stats_collect.log 2>&1 < /dev/null
ls $out
