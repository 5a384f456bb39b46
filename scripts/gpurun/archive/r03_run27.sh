#!/bin/bash
set -u
mkdir -p gpurun_out/r03_run27
timeout 300 python scripts/kpi_cost_probe.py > gpurun_out/r03_run27/kpi_vec2_probe.log 2>&1; grep -v amdgpu.ids gpurun_out/r03_run27/kpi_vec2_probe.log
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from golden_util import golden
from citylearn_amd.engine import StepEngine
for name,E in (('g2020_cz1',4096),('g2023_p2',516),('s_2023_p3',1028)):
    g=golden(name); spec=g.spec(); tab=spec.episode_tables(0)
    low,high=spec.action_limits(); lo,hi=torch.from_numpy(low).cuda()[:,None],torch.from_numpy(high).cuda()[:,None]
    a1=StepEngine(tab,E,kpi=True,tuning=dict(nw=3)); a2=StepEngine(tab,E,kpi=True,tuning=dict(vec=2,nw=3))
    a1.trace_kernels(); a2.trace_kernels()
    gen=torch.Generator(device='cuda').manual_seed(1)
    for t in range(40):
        a=(lo+torch.rand((a1.n_act_cols,E),device='cuda',generator=gen)*(hi-lo)).contiguous()
        a1.step(a,t); a2.step(a,t)
    print(name, a1.last_kernels, a2.last_kernels, 'state',torch.equal(a1.state,a2.state),'out_env',torch.equal(a1.out_env,a2.out_env),'out_bldg',torch.equal(a1.out_bldg[:2],a2.out_bldg[:2]),'kpi_bldg',torch.equal(a1.kpi_bldg,a2.kpi_bldg),'kpi_env',torch.equal(a1.kpi_env,a2.kpi_env), float((a1.kpi_bldg-a2.kpi_bldg).abs().max()))
PY
