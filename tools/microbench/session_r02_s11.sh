mkdir -p gpurun_out/s11
timeout 1200 python3 -m pytest tests -m gpu -x -q > gpurun_out/s11/pytest.log 2>&1
tail -3 gpurun_out/s11/pytest.log
python3 tools/microbench/ab_libs.py --rounds 2 > gpurun_out/s11/ab.log 2>&1
tail -4 gpurun_out/s11/ab.log
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
timeout 250 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d gpurun_out/s11/pmc -o p -- python bench.py --steps 3 --warmup 2 --prelude-ms 0 --headline-only > gpurun_out/s11/pmc.log 2>&1
python3 - <<'PY'
import csv, glob
f=glob.glob('gpurun_out/s11/pmc/**/*counter_collection.csv', recursive=True)[0]
acc={}
for r in csv.DictReader(open(f)):
    if 'f16_env_kernel' in r['Kernel_Name'] and 'true, true' in r['Kernel_Name']:
        acc.setdefault(r['Counter_Name'],{}).setdefault(r['Dispatch_Id'],0.0)
        acc[r['Counter_Name']][r['Dispatch_Id']]+=float(r['Counter_Value'])
        waves=int(r['Grid_Size'])/64
for k,v in acc.items(): print(k, sum(v.values())/len(v)/waves)
PY
