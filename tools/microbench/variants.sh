cd $GRAFT_REPO_ROOT
run() { timeout 120 python bench.py --steps 80 --warmup 5 --no-cpu-baseline $2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'value', '%.3e'%d['value'])"; }
for rep in 1 2; do
NPF16_EXTRA_FLAGS="-DNPF16_STAGGER_CYCLES=0" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
run "no stagger"
NPF16_EXTRA_FLAGS="-DNPF16_STAGGER_CYCLES=10000" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
run "slot stagger 10000"
NPF16_EXTRA_FLAGS="-DNPF16_STAGGER_CYCLES=10000 -DNPF16_STAGGER_BY_BLOCK" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
run "block%3 stagger 10000"
NPF16_EXTRA_FLAGS="-DNPF16_STAGGER_CYCLES=20000 -DNPF16_STAGGER_BY_BLOCK" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
run "block%3 stagger 20000"
done
python -c "from neuralplane_amd import build; build.build_hip(force=True)"
