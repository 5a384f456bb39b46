import ctypes, sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo/scripts')
import torch
from golden_util import golden
from citylearn_amd import _lib
from citylearn_amd.engine import StepEngine
from copy_floor import timed
lib=_lib.load(); lib.cl_debug_set_lean.argtypes=[ctypes.c_int,ctypes.c_int]
tab=golden('g2022_all').spec().episode_tables(0)
for E in (65536,):
    eng=StepEngine(tab,E); a=torch.rand((eng.n_act_cols,E),device='cuda')*2-1
    for vec in (1,2,4):
        for nw in (9,10,12,16):
            lib.cl_debug_set_vec(vec); lib.cl_debug_set_lean(0,nw)
            try:
                us=timed(lambda: eng.step(a,5))
                print(f'E={E} vec={vec} nw={nw}: {us:.2f} us', flush=True)
            except Exception as ex: print('fail',vec,nw,ex)
lib.cl_debug_set_vec(0); lib.cl_debug_set_lean(0,0)
