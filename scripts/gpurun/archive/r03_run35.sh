#!/bin/bash
# flexible loads inside the step launch: identity against the two launches, timing
set -u
mkdir -p gpurun_out/r03_run35
python - <<'PY'
import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import torch
from golden_util import golden
from citylearn_amd.engine import StepEngine
for name in ('g2022_evs', 'g_cc_demo', 'g_evs_central'):
    g=golden(name); spec=g.spec(); tab=spec.episode_tables(0)
    for reward in ('MARL','Electric_Vehicles_Reward_Function','RewardFunction'):
        E=65536 if name=='g2022_evs' else 16384
        a1=StepEngine(tab,E,reward=reward,central_agent=spec.central_agent); a2=StepEngine(tab,E,reward=reward,central_agent=spec.central_agent,tuning=dict(lean_variant=8))
        a1.trace_kernels(); a2.trace_kernels()
        gen=torch.Generator(device='cuda').manual_seed(1)
        for t in range(30):
            a=(torch.rand((a1.n_act_cols,E),device='cuda',generator=gen)*2-1).contiguous()
            a[torch.rand(a.shape,device='cuda',generator=gen)<0.2]=0.0
            a1.step(a,t); a2.step(a,t)
        print(name,reward,a1.last_kernels,'|',a2.last_kernels,'state',torch.equal(a1.state,a2.state),'ev',torch.equal(a1.ev_state,a2.ev_state),'out',torch.equal(a1.out_bldg[:2],a2.out_bldg[:2]),'env',torch.equal(a1.out_env,a2.out_env),'flex_out',torch.equal(a1.flex_out,a2.flex_out))
PY
timeout 300 python scripts/ev_step_bench.py > gpurun_out/r03_run35/ev_step_bench.log 2>&1; tail -1 gpurun_out/r03_run35/ev_step_bench.log
