"""Do forked hipGraph branches overlap on this stack?  A ~10 us kernel per iteration (`main`, 40 MB copy) followed by a small dependent
kernel (`tail`, launch-latency sized) -- serially in one stream, or with the tail on a forked branch that is joined only after the NEXT
iteration's main kernel (what a pipelined cl_finish_kernel would do)."""
import torch
x = torch.rand(10 * 1024 * 1024, device='cuda'); y = torch.empty_like(x)
small = [torch.zeros(65536, device='cuda') for _ in range(2)]
S = torch.cuda.Stream(); A = torch.cuda.Stream()
N = 200


def build(forked):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=S):
        prev = None
        for i in range(N):
            y.copy_(x)                                   # main(i)
            if not forked:
                small[i & 1].add_(y[:65536])             # tail(i) right behind it
                continue
            ev = torch.cuda.Event(); ev.record(S)
            A.wait_event(ev)
            with torch.cuda.stream(A):
                small[i & 1].add_(y[:65536])             # tail(i) on the branch (reads what main(i) wrote; main(i+1) rewrites the same values)
                done = torch.cuda.Event(); done.record(A)
            if prev is not None:
                S.wait_event(prev)                       # join tail(i-1) only now, after main(i) was enqueued
            prev = done
        if forked:
            S.wait_event(prev)
    return g


for forked in (False, True, False, True):
    g = build(forked)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f'forked={forked}: {e0.elapsed_time(e1) / (5 * N) * 1e3:.2f} us per iteration', flush=True)
