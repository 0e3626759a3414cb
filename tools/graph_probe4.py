"""GPU probe: what a second branch costs a replayed HIP graph (diagnostics): chains of small fills, recorded as one chain or as two."""
import time
import torch

dev = torch.device("cuda:0")
a = torch.zeros(1 << 20, device=dev)
b = torch.zeros(1 << 20, device=dev)

def record(n_main, n_side):
    side = torch.cuda.Stream()
    cap = torch.cuda.Stream()
    cap.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap):
        if n_side:
            side.wait_stream(cap)
            with torch.cuda.stream(side):
                for _ in range(n_side):
                    b.add_(1.0)
        for _ in range(n_main):
            a.add_(1.0)
        if n_side:
            cap.wait_stream(side)
        a.add_(1.0)   # the join's consumer
    return g

for n_main, n_side in ((8, 0), (4, 4), (8, 8), (16, 0), (1, 1), (2, 0)):
    g = record(n_main, n_side)
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    K = 500
    t0 = time.perf_counter()
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"{n_main} launches on the main chain, {n_side} on a branch (+1 after the join): {1e6 * (t1 - t0) / K:6.1f} us per replay")
