#!/bin/bash
# flexible-load units inside the lean step launch (owning wave): parity tests + A/B against the two launches
set -u
O=gpurun_out/r04_run11; mkdir -p $O
bash scripts/gpurun/r04_run12.sh 2>/dev/null | tail -2
timeout 900 python -m pytest tests/test_gpu_flex.py -x -q 2>&1 | tail -4
for tun in "flex_fused=2" "flex_fused=0" "flex_fused=0,vec=2" "flex_fused=0,vec=1"; do
  echo "== CL_TUNING=$tun"; CL_TUNING=$tun timeout 300 python scripts/ev_step_bench.py 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print({k:v for k,v in d.items() if k.startswith('graph')})"
done
