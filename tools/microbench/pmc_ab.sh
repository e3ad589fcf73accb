cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
pm() { tag=$1; shift; timeout 250 rocprofv3 --pmc SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_DCACHE_MISSES_DUPLICATE SQC_DCACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_TC_STALL SQC_DCACHE_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/pmcD_$tag -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/pmcD_$tag.log 2>&1 < /dev/null; echo "$tag rc=$?"; }
NPF16_EXTRA_FLAGS="-DNPF16_DBG_OLD_SCHEME -DNPF16_DBG_NOSTORE" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
NPF16_NO_CACHE=1 pm old
NPF16_EXTRA_FLAGS="-DNPF16_DBG_NOSTORE -DNPF16_DBG_NOLOAD" python -c "from neuralplane_amd import build; build.build_hip(force=True)"
pm new
