#!/usr/bin/env python3
"""Schedule model of per-ray early termination with re-packing (diagnostic): level-synchronous (one launch per segment, what
gpnerf_render_fused does) against a barrier-free ray FIFO inside one launch.
Input: per-ray stop indices (tools/et_potential.py -> gpurun_out/stop_rays.npy) and the measured step time of a wave against
the number of waves active on its CU (GPNERF_WAVE_CAP sweep of the headline frame)."""
import sys
import heapq
import numpy as np

STEP_US = {0: 0.0, 1: 29.6, 2: 30.6, 3: 31.5, 4: 31.5, 5: 41.8, 6: 46.3, 7: 53.5, 8: 55.2}   # MI355X, fp32 form, round 2
N_CU, WAVES = 256, 8


def run_level(n_tiles, steps, launch_us=15.0):
    """one launch: n_tiles items of `steps` steps over N_CU x WAVES wave slots, dealt evenly; returns its duration in us"""
    if n_tiles == 0:
        return 5.0
    per_cu = np.full(N_CU, n_tiles // N_CU)
    per_cu[: n_tiles % N_CU] += 1
    worst = 0.0
    for c in np.unique(per_cu):
        # c tiles on a CU: full rounds of 8 waves, then a partial round
        t, left = 0.0, int(c)
        while left > 0:
            w = min(WAVES, left)
            t += steps * STEP_US[w]
            left -= w
        worst = max(worst, t)
    return worst + launch_us


def level_synchronous(stop, seg, S):
    n_seg = (S + seg - 1) // seg
    total, alive = 0.0, len(stop)
    for s in range(n_seg):
        total += run_level((alive + 31) // 32, min(seg, S - s * seg))
        alive = int((stop > (s + 1) * seg).sum())
    return total / 1e3


def fifo(stop, seg, S, dt=1.0, deep_first=False, fresh_first=True):
    """event-free time stepping: waves pull fresh tiles first, then full groups of 32 parked rays (lowest segment first),
    partial groups once nothing below can produce any more"""
    n_seg = (S + seg - 1) // seg
    order = np.arange(len(stop))
    fresh = [order[i:i + 32] for i in range(0, len(stop), 32)][::-1]
    q = [[] for _ in range(n_seg + 1)]                 # parked rays per segment
    inflight = np.zeros(n_seg + 1, dtype=np.int64)     # rays being processed at segment s
    cur = [[None] * WAVES for _ in range(N_CU)]        # (segment, rays, steps left)
    t = 0.0
    n_left = len(stop)
    while n_left > 0:
        for cu in range(N_CU):
            for w in range(WAVES):
                if cur[cu][w] is not None:
                    continue
                item = None
                if fresh and fresh_first:
                    item = (0, fresh.pop())
                else:
                    for s in (range(n_seg - 1, 0, -1) if deep_first else range(1, n_seg)):
                        upstream = sum(len(q[j]) for j in range(1, s)) + inflight[:s].sum() + (32 * len(fresh))
                        if len(q[s]) >= 32 or (q[s] and upstream == 0):
                            item = (s, np.array(q[s][:32])); del q[s][:32]
                            break
                    if item is None and fresh:
                        item = (0, fresh.pop())
                if item is not None:
                    inflight[item[0]] += len(item[1])
                    cur[cu][w] = [item[0], item[1], float(min(seg, S - item[0] * seg))]
        for cu in range(N_CU):
            act = sum(1 for x in cur[cu] if x is not None)
            if not act:
                continue
            prog = dt / STEP_US[act]
            for w in range(WAVES):
                x = cur[cu][w]
                if x is None:
                    continue
                x[2] -= prog
                if x[2] <= 0:
                    s, rays = x[0], x[1]
                    inflight[s] -= len(rays)
                    go = rays[stop[rays] > (s + 1) * seg] if s + 1 < n_seg else rays[:0]
                    q[s + 1].extend(go.tolist())
                    n_left -= len(rays) - len(go)
                    cur[cu][w] = None
        t += dt
    return t / 1e3


if __name__ == "__main__":
    stop = np.load(sys.argv[1]).astype(np.int64)
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    lanes = {seg: float((np.ceil(stop / seg) * seg).mean()) / S for seg in (8, 16, 32)}
    print(f"{len(stop)} rays, mean stop {stop.mean() / S:.3f} of S; lane-steps with segment 8/16/32: {lanes}")
    for seg in (8, 16, 32):
        work = lanes[seg] * (len(stop) / 32) * S * STEP_US[8] / 8 / N_CU / 1e3
        print(f"segment {seg}: saturated {work:.2f} ms, level-synchronous {level_synchronous(stop, seg, S):.2f} ms")
    sub = stop[:: 4]                                   # the FIFO model steps in time: a quarter of the rays on a quarter of the CUs
    N_CU = 64
    print(f"level-synchronous on the quarter-size model: {level_synchronous(sub, 16, S):.2f} ms")
    for deep in (False, True):
        for ff in (True, False):
            print(f"barrier-free FIFO, segment 16, {'deepest' if deep else 'shallowest'} segment first, fresh tiles {'first' if ff else 'last'}: "
                  f"{fifo(sub, 16, S, dt=4.0, deep_first=deep, fresh_first=ff):.2f} ms")
