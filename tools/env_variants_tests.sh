#!/bin/bash
# The decoder / scoring parity tests under every diagnostic switch (the switches select other kernels or other plans; results
# must not depend on them).  GPU box: bash tools/env_variants_tests.sh  → gpurun_out/variants/<name>.log
mkdir -p gpurun_out/variants
T="tests/test_gpu_lazy.py tests/test_gpu_parity.py tests/test_gpu_headline_config.py tests/test_gpu_general.py tests/test_gpu_edge_cases.py tests/test_gpu_config2_fullsize.py"
for v in "MFA_VIT_LEAN=0" "MFA_VIT_LAG=0" "MFA_PLAN_GROUPS=1" "MFA_GMM_PRESPLIT=0" "MFA_LAZY_LOOKAHEAD=16" "MFA_LAZY_LOOKAHEAD=63" "MFA_GMM_F16=0" "MFA_GMM_BF16=0" "MFA_PLAN_SPAN=8"; do
  name=$(echo $v | tr '=' '_')
  env $v timeout -k 10 600 python -m pytest $T -m gpu -q > gpurun_out/variants/$name.log 2>&1
  echo "$v: $(tail -1 gpurun_out/variants/$name.log)"
done
# Expected to fail under a switch, because they assert the DEFAULT configuration itself: test_gpu_config2_fullsize (groups == 8)
# under MFA_PLAN_GROUPS=1; test_speculative_lookahead_failures_fall_back_to_the_proven_band[6|24] (speculation fills fewer cells
# than the proven band) under MFA_LAZY_LOOKAHEAD=16 and =63.  Round 3: everything else green under all eight switches.
