"""Where do the ~22 us between the wall clock and the kernel time of a K = 20 timed region go (driver flags: --steps 20 --warmup 5)?
The headline workload, K steps as one hipGraph replay between synchronize() calls: with / without event records inside the region, and
(run this script twice) with HSA_ENABLE_INTERRUPT=0 (signal waits poll instead of sleeping on an interrupt)."""
import os
import statistics
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import bench

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = bench.build_workload('headline', 65536, 'cuda:0', 0, 1, {}, False, False, 8760)
stream = torch.cuda.Stream()
runner = bench.Runner(wl.step_fn, wl.period, stream, True, getattr(wl, 'flush', None))
with torch.cuda.stream(stream):
    runner.run(0, 50); stream.synchronize()
    runner.prepare(5, K)
    wl.reset()
    for events in (True, False, True, False):
        walls = []
        for _ in range(200):
            torch.cuda.synchronize()
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0 = time.perf_counter()
            if events: ev0.record(stream)
            runner.advance(5, K)
            if events: ev1.record(stream)
            stream.synchronize()
            torch.cuda.synchronize()
            walls.append(time.perf_counter() - t0)
        walls.sort()
        print(f'K={K} HSA_ENABLE_INTERRUPT={os.environ.get("HSA_ENABLE_INTERRUPT", "(unset)")} events={events}: median {statistics.median(walls) / K * 1e6:.3f} us/step, '
              f'p10 {walls[20] / K * 1e6:.3f}, min {walls[0] / K * 1e6:.3f}  (region {statistics.median(walls) * 1e6:.1f} us)')
