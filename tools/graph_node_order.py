"""Are memset / memcpy NODES of a captured hipGraph ordered like the kernels around them?  (Round 5: a hipMemsetAsync between two
kernels of the celerite reverse pass was not, once in a few thousand replays -- exo_math.hpp, zero_fill_async.)

A captured step:  K1: x[:] = c (c counts the replays, on the device)  ->  NODE  ->  K2: bad += #(elements of the node's target that
are not what the node should have left).  NODE is one of
    memset   hipMemsetAsync(x, 0)            K2 expects 0 everywhere          (stale = this replay's c)
    memcpy   y.copy_(x)  (hipMemcpyAsync)    K2 expects y == c everywhere     (stale = the previous replay's c)
    kernel   x.zero_() / an elementwise copy: the control
replayed back to back, or with what a sampler does between replays (eager launches, a device-to-host read every eighth replay).
python tools/graph_node_order.py [replays] [doubles ...]"""
import ctypes
import sys

import torch

dev = torch.device("cuda:0")
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
n_rep = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
sizes = [int(a) for a in sys.argv[2:]] or [768, 131072]


def run(kind, n, busy):
    x = torch.zeros(n, dtype=torch.float64, device=dev)
    y = torch.zeros(n, dtype=torch.float64, device=dev)
    c = torch.zeros((), dtype=torch.float64, device=dev)
    bad = torch.zeros((), dtype=torch.int64, device=dev)
    filler = torch.zeros(1 << 16, dtype=torch.float64, device=dev)
    filler2 = torch.zeros(1 << 16, dtype=torch.float64, device=dev)

    def step():
        c.add_(1.0)
        x.copy_(c.expand(n))                                   # K1 (a broadcast: an elementwise kernel)
        if kind == "memset":
            s = torch.cuda.current_stream(dev).cuda_stream
            assert hip.hipMemsetAsync(ctypes.c_void_p(x.data_ptr()), 0, n * 8, ctypes.c_void_p(s)) == 0
            bad.add_((x != 0).sum())
        elif kind == "kernel_zero":
            x.zero_()
            bad.add_((x != 0).sum())
        elif kind == "memcpy":
            y.copy_(x)                                         # same dtype, both contiguous: a device-to-device copy
            bad.add_((y != c).sum())
        elif kind == "kernel_copy":
            torch.add(x, 0.0, out=y)
            bad.add_((y != c).sum())
        filler.mul_(1.0)                                       # something for the next replay's K1 to queue behind

    side = torch.cuda.Stream(dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream(dev).wait_stream(side)
    bad.zero_()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    bad.zero_()
    first = None
    for i in range(n_rep):
        g.replay()
        if busy and i % 8 == 7:          # what a sampler does between replays: eager launches, device-to-device copies (the
            filler.add_(0.0)             # runtime's own blit kernels) and a device-to-host read
            for _ in range(12):
                filler2.copy_(filler)
            if int(bad) and first is None:
                first = i
    torch.cuda.synchronize()
    return int(bad), first


for busy in (False, True):
    for n in sizes:
        for kind in ("memset", "kernel_zero", "memcpy", "kernel_copy"):
            bad, first = run(kind, n, busy)
            print("%-12s %8d doubles, %s between replays: %d wrong elements in %d replays%s"
                  % (kind, n, "eager launches, copies, host reads" if busy else "nothing", bad, n_rep, "" if first is None else " (first seen at replay %d)" % first), flush=True)
