#!/bin/bash
env | grep -i "^ROCP\|rocprof" | head
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1])['roofline']; print(r['traffic'], r['traffic_source'], r.get('traffic_live_error'))"
export TMPDIR=/tmp; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/nest -o run -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-streaming 2>/dev/null | python -c "
import json,sys; r=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])['roofline']; print('under rocprof:', r['traffic_source'], '|', r.get('traffic_live_error'))"
