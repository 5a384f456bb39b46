"""Issue-rate microbenchmark behind the LSTM kernel's design notes (GPU box): ns per MFMA / per 4 v_exp for dependent
MFMA chains, with and without interleaved transcendental VALU work, at 1 and 2 waves per SIMD."""
import ctypes, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd import _lib
lib = _lib.load_tune()
out = torch.zeros(256, device='cuda')
names = {0: 'bf16 32x32x16, 1 chain', 1: 'bf16, 2 chains', 2: 'bf16, 4 chains', 3: 'bf16 2 chains + 4 v_exp after each MFMA',
         4: 'the v_exp work alone (32 per iteration)', 5: 'f32 32x32x2, 2 chains', 6: 'f32 2 chains + 4 v_exp after each MFMA',
         7: 'bf16 2 chains (8 MFMAs) then the 32 v_exp as a block'}
iters = 20000
for wps in (1, 2):
    for mode in range(8):
        def run():
            assert lib.cl_tune_mfma_bench(mode, wps, iters, out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ns_iter = e0.elapsed_time(e1) * 1e6 / iters
        print(f'{wps} wave(s)/SIMD  {names[mode]:58s}: {ns_iter:7.1f} ns per iteration (8 MFMAs and/or 32 v_exp) = {ns_iter / 8:6.1f} ns per MFMA slot', flush=True)
