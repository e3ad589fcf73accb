#!/bin/bash
# round 4, GPU session 16: (a) A/B of the cache check: full cold path (k_check3), a one-instruction cold path (k_trap), no check (k_nocheck);
# (b) the whole GPU suite on the library built with NPF16_PIN_MASK=7 (the trigonometry pins in every variant: does the parked fault reproduce?)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_s16; mkdir -p $out
python tools/microbench/ab_libs.py --rounds 3 --steps 100 k_check3 k_trap k_nocheck 2>&1 | tee $out/ab_cache_check4.log | tail -5
NPF16_LIB=tools/microbench/libs/k_pin7.so timeout 1500 python -m pytest tests -q -m gpu -x --deselect tests/test_gpu_step_parity.py::test_cross_step_cache_validates_itself_against_invisible_state_edits > $out/gputest_pin7.log 2>&1; echo "pin7 suite rc=$?"; tail -3 $out/gputest_pin7.log; grep -B5 -A25 "Error\|FAILED\|fault" $out/gputest_pin7.log | head -60
