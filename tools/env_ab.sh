#!/bin/bash
# same-box alternating A/B of ONE library under an environment switch (3 rounds):   tools/env_ab.sh UVC_GELU_GRAD_BF16=1 [bench.py arguments]
# prints <variant> <img/s> <ms per step> <sum of stand-alone kernel ms>
V=$1; shift
for r in 1 2 3; do
  for v in default "$V"; do
    if [ "$v" = default ]; then E=""; else E="$v"; fi
    env $E python bench.py --no_cpu_baseline --steps 60 "$@" 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());print('$v',d['value'],d['ms_per_step'],d['kernel_ms_per_step_standalone_sum'])"
  done
done
