#!/usr/bin/env python3
"""LDS cycles of k_fast_cells' quick-test reads for a cell of ng 4-pixel groups per row, for every tile pitch (dwords): the eleven
ds_read_b32 of a step, bank = dword address mod 32, conflicts counted inside each 32-lane half (cdna_hip_programming.md section 2).
Row-major item order (lane -> (row, group) = divmod(item, ng)), as the ring's pixel order needs it."""
import sys


def cost(P, ng, dh, offs):
    tot = ideal = 0
    n = ng * dh
    for b in range(0, n, 64):
        for half in (0, 32):
            for dy, dx in offs:
                banks = {}
                for l in range(32):
                    ip = b + half + l
                    if ip >= n:
                        continue
                    row, gi = divmod(ip, ng)
                    a = (row + 3 + dy) * P + gi + 1 + dx
                    banks.setdefault(a % 32, set()).add(a)
                if banks:
                    tot += max(len(v) for v in banks.values())
                    ideal += 1
    return tot, ideal


OFFS = [(-3, 0), (3, 0), (0, -1), (0, 0), (0, 1), (-2, -1), (-2, 0), (-2, 1), (2, -1), (2, 0), (2, 1)]
dh = int(sys.argv[1]) if len(sys.argv) > 1 else 35
for ng in (9, 10):
    for P in range(ng + 2, 24):
        t, i = cost(P, ng, dh, OFFS)
        print("groups per row %2d  pitch %2d dwords: %4d LDS cycles for %4d reads (%.2f x)" % (ng, P, t, i, t / i))
