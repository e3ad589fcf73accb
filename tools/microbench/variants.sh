cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 80 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'steps/s', '%.3e'%d['value'])"; }
for n in 786432 917504 983040 1000000 1048576 1179648 1200000 1966080 2000000; do run "n=$n" "--n $n"; done
