#!/bin/bash
# Round 6, seventh GPU call: smoke() (free-running now), the env / EV / observe tests with the tightened gates, recorded.
set -u
OUT=gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
rm -f $OUT/parity.jsonl
CL_PARITY_REPORT=$OUT/parity.jsonl CL_PARITY_MEASURE=1 timeout 1200 python -m pytest tests/test_env_gpu.py tests/test_gpu_flex.py tests/test_gpu_observe.py -m gpu -q > $OUT/tests.log 2>&1
echo "tests rc=$?"; grep -E "^(FAILED|ERROR)|passed|failed" $OUT/tests.log | tail -20
python - <<'PY'
import json
for l in open('gpurun_out/r06g/parity.jsonl'):
    r = json.loads(l); w = r['worst']; k = max(w, key=w.get)
    print('%-90s %-50s worst %s = %.3f' % (r['test'].split('::')[-1][:90], r['label'][:50], k, w[k]))
PY
