#!/bin/bash
# round 4, session 19: PMC pass over PlanningEnv's persistent kernel (n = 8 192): matrix-pipe counters
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; out=gpurun_out/r04_s19; mkdir -p $out
rocprofv3 --list-avail 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_BUSY_CYCLES\|SQ_VALU_MFMA_BUSY_CYCLES\|GRBM_GUI_ACTIVE" | sort -u > $out/mfma_counters.txt
cat $out/mfma_counters.txt
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"; do
  t=$(echo $set | cut -d" " -f1)
  timeout 250 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $out/pmc_$t -o p -- python tools/microbench/planning_profile.py 8192 6 0 persistent 8 > $out/pmc_$t.log 2>&1 < /dev/null
  f=$(find $out/pmc_$t -name "p_counter_collection.csv" | head -1)
  echo "== $set"; [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(list)
for r in rows:
    if 'planning_persistent' in r['Kernel_Name']:
        agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    print(f'{k}: launches {len(v)}, mean per launch {sum(v) / len(v):.6g}')
PY
done 2>&1 | tee $out/planning_pmc.log
