cd $GRAFT_REPO_ROOT
for v in "-DNPF16_BLOCK=256 -DNPF16_MINWAVES=3" "-DNPF16_BLOCK=192 -DNPF16_MINWAVES=4" "-DNPF16_BLOCK=64 -DNPF16_MINWAVES=4" "-DNPF16_BLOCK=128 -DNPF16_MINWAVES=4" "-DNPF16_BLOCK=128 -DNPF16_MINWAVES=3" "-DNPF16_BLOCK=64 -DNPF16_MINWAVES=3"; do
  NPF16_EXTRA_FLAGS="$v" python -c "from neuralplane_amd import build; build.build_hip(force=True)" 2>&1 | tail -2
  timeout 120 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', 'kernel_ms', round(d['roofline']['kernel_avg_ms'],4), 'value', '%.3e'%d['value'])"
done
