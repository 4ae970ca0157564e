"""CPU: rounds of resident waves of the celerite chunk kernels at the benchmarked shapes -- waves of a launch over the waves the
chip holds at the kernel's occupancy (tools/kres.py: registers -> waves per SIMD; 1024 SIMDs).  A launch that is 1.33 rounds
costs two: the J <= 2 element kernel at three waves per SIMD did (round 4), at four it is one round.
usage: python tools/rounds.py"""
import math
import os
import re
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = subprocess.run([sys.executable, os.path.join(R, "tools", "kres.py"), "exo_celerite.hip"], capture_output=True, text=True).stdout
occ = {}
for line in out.splitlines():
    m = re.search(r"(\S+)\s+VGPR\s+(\d+)\s+AGPR\s+(\d+)\s+scratch\s+(\d+)\s+occ\s+(\d+)", line)
    if m:
        occ[m.group(1)] = (int(m.group(2)) + int(m.group(3)), int(m.group(4)), int(m.group(5)))


def find(sub):
    return [(k, v) for k, v in occ.items() if sub in k]


def plan(n_draw, n):      # exo_celerite_core.hpp, chunk_plan (one-lane path)
    C = min(512, max(4, (64 * 4096) // n_draw))
    C = min(C, n // 32)
    L = -(-n // C)
    L = -(-L // 4) * 4
    return -(-n // L)


SHAPES = (("C3 (J = 2, 1024 draws, 150 000 cadences)", 1024, 150_000, ("celerite_elem_mixed_kernel", "celerite_chunk1_fwd_mixed_kernel", "celerite_chunk1_vjp_mixed_kernel")),
          ("J = 4 at the C3 shape", 1024, 150_000, ("elem_kernelILi4ELi0", "chunk1_fwd_kernelILi4ELi0", "chunk1_vjp_kernelILi4ELi0")),
          ("C5 (J = 6, 128 chains, 65 000 cadences)", 128, 65_000, ("elem_kernelILi6", "chunk1_fwd_kernelILi6", "chunk1_vjp_kernelILi6")),
          ("C5 on one GPU (1024 chains)", 1024, 65_000, ("elem_kernelILi6", "chunk1_fwd_kernelILi6", "chunk1_vjp_kernelILi6")))
for name, D, N, kernels in SHAPES:
    C = plan(D, N)
    waves = -(-D // 64) * C
    print(f"{name}: {C} chunks, {waves} waves per launch")
    for sub in kernels:
        for k, (regs, scratch, o) in find(sub)[:1]:
            cap = 1024 * o
            print(f"   {k[:58]:58s} {regs:4d} registers, {scratch:3d} B scratch, {o} waves/SIMD: {waves / cap:5.2f} rounds -> {math.ceil(waves / cap)}")
