cd $GRAFT_REPO_ROOT
for n in 1000000 200000; do for pw in 2 3 2 3; do NPF16_PAIR_WAVES=$pw python bench.py --headline-only --no-cpu-baseline --solver rk4 --steps 50 --warmup 5 --n $n 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rk4 pw=$pw N=$n', round(d['roofline']['kernel_avg_ms'],4), round(d['roofline']['kernel_median_ms'],4))"; done; done
