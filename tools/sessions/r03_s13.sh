#!/bin/bash
# round 3, GPU session 13: trigonometry rows in the cross-step cache (A/B vs the same tree without them), latency4w variant,
# full GPU suite, PMC instruction count
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r03_s13; mkdir -p $out
timeout 2400 python -m pytest tests -x -q -m gpu > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
timeout 900 python tools/microbench/ab_libs.py --rounds 3 --steps 100 i_notrig i_head > $out/ab_1e6.log 2>&1; grep -v Warn $out/ab_1e6.log | tail -3
timeout 600 python tools/microbench/ab_libs.py --rounds 2 --steps 20 --n 10000000 i_notrig i_head > $out/ab_1e7.log 2>&1; grep -v Warn $out/ab_1e7.log | tail -3
for lib in i_notrig i_head; do NPF16_LIB=tools/microbench/libs/$lib.so timeout 400 python tools/microbench/mid_n.py --variants auto --out $out/mid_$lib.json 3000 10000 49152 57344 65536 98304 131072 262144 > $out/mid_$lib.log 2>&1; grep "N=" $out/mid_$lib.log | sed "s/^/$lib /"; done
timeout 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $out/pmc_valu -o p -- python bench.py --steps 3 --warmup 2 --prelude-ms 0 --headline-only > $out/pmc_valu.log 2>&1 < /dev/null
python - <<'PY'
import csv,glob
cc=glob.glob('gpurun_out/r03_s13/pmc_valu/**/*counter_collection.csv',recursive=True)
acc={}
for r in csv.DictReader(open(cc[0])):
    if 'f16_env_kernel' not in r['Kernel_Name'] or 'true, true' not in r['Kernel_Name']: continue
    a=acc.setdefault(r['Counter_Name'],{}); a[r['Dispatch_Id']]=a.get(r['Dispatch_Id'],0)+float(r['Counter_Value']); g=int(r['Grid_Size'])
for k,a in acc.items():
    v=list(a.values()); print(k, sum(v)/len(v)/(g/64))
PY
