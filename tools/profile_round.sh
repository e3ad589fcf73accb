#!/bin/bash
# Collect the round's evidence on the GPU box: bench JSON, rocprofv3 kernel stats, PMC passes (separate runs).
# usage (via gpurun): bash tools/profile_round.sh r01c
tag=${1:-rXX}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/$tag; mkdir -p $out
timeout 300 python bench.py --steps 200 --warmup 20 > $out/bench.json 2> $out/bench.err < /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o p -- python bench.py --steps 50 --warmup 5 --no-cpu-baseline > $out/stats.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_combat -o p -- python tools/microbench/combat_bench.py > $out/stats_combat.log 2>&1 < /dev/null
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_actor -o p -- python tools/microbench/actor_bench.py 262144 > $out/stats_actor.log 2>&1 < /dev/null
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_REQ" "FETCH_SIZE" "WRITE_SIZE"; do
  t=$(echo $set | cut -d" " -f1)
  timeout 250 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_$t -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > $out/pmc_$t.log 2>&1 < /dev/null
done
ls $out
