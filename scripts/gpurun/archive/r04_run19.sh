#!/bin/bash
# another box: the default bench line and the C4 lines (box-to-box spread of the round)
set -u
O=gpurun_out/r04_run19; mkdir -p $O
python bench.py > $O/bench_line.json 2>/dev/null
for c in C4 C4-lean T9; do python bench.py --config $c > $O/bench_$c.json 2>/dev/null; done
python -c "
import json
for f in ('bench_line','bench_C4','bench_C4-lean','bench_T9'):
    d=json.load(open('$O/'+f+'.json')); r=d['roofline']; print(f, '%.2f us frac %.3f' % (r['launch_us'], r['frac']), r.get('hbm_streaming',{}).get('frac'))"
