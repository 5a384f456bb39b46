"""Is the headline launch (17 buildings x 65 536 envs, one 9-wave workgroup per CU, wave w = buildings w and w + 9) bound by the ARITHMETIC of
its most loaded SIMD?  Waves go to the four SIMDs of a CU round-robin: SIMD 0 gets waves 0, 4, 8 = five building tiles, the others four.  If
that is the critical path, a district of 16 buildings (8 waves, four tiles per SIMD) should be ~20 % faster than 17, not the 6 % its bytes
say, and 12 / 13 buildings (3 vs 4 tiles on SIMD 0) likewise.  us per step by building count, fp32 map and float64 chain, hipGraph replay."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from citylearn_amd import load_district
from citylearn_amd.data import sample_schema
from citylearn_amd.engine import StepEngine
from citylearn_amd.synthetic import tile_district
sys.path.insert(0, str(ROOT / 'scripts'))
from f64_cost import measure   # noqa: E402  (prints its own table first when imported as a script -- so guard there)

if __name__ == '__main__':
    E = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    spec0 = load_district(sample_schema('citylearn_challenge_2022_phase_all_720h'))
    for B in (8, 9, 12, 13, 16, 17, 18, 20, 24, 25, 32):
        spec = tile_district(spec0, B, jitter=0.0)
        tab = spec.episode_tables(0)
        low, high = spec.action_limits()
        lo, hi = torch.from_numpy(low).cuda(), torch.from_numpy(high).cuda()
        acts = [lo[:, None] + torch.rand((len(low), E), device='cuda') * (hi - lo)[:, None] for _ in range(2)]
        row = []
        for f64 in (False, 'chain'):
            best = 1e9
            for rnd in range(2):
                eng = StepEngine(tab, E, f64_maps=f64)
                eng.trace_kernels()
                best = min(best, measure(eng, acts, steps=200, reps=5))
                k = eng.last_kernels
                del eng
            row.append((best, k))
        print(f'B={B:3d} E={E}: fp32 {row[0][0]:.3f} us ({row[0][0] * 1e3 / (B * E / 1024):.3f} ns/kunit)  chain {row[1][0]:.3f} us '
              f'({row[1][0] * 1e3 / (B * E / 1024):.3f} ns/kunit)   {row[0][1]} | {row[1][1]}', flush=True)
