#!/usr/bin/env python3
"""Schedule model of a frame of one to two rounds of wavefronts (the shape of a real ZJU frame: 50 - 75 k rays) on the fused
kernel's tile queue, fed with the paces tools/wave_times.py measured on the MI355X:
  a SIMD holds two wavefronts and serves the one that was launched first ("older") before the other: paired, the older walks
  a 32-sample step in ~43.9 us and the younger in ~76 us (together one step per 27.8 us in a lone round, 25.9 in steady state);
  a wavefront left alone on its SIMD needs ~46.7 us per step (latency-bound: 55 % of the pair's throughput).
Work units: (P samples of a ray per step) x (32 / P rays): S / P dependent steps, the same arithmetic per step; P > 1 costs a
little more per step (cross-lane composite, per-unit set-up).  The model hands units out from per-class queues and reports
when the last wavefront leaves.   usage: sim_lone_round.py [tiles] [S]"""
import heapq
import sys

PACE_OLD, PACE_YOUNG, PACE_ALONE = 43.9, 76.0, 46.7           # us per step
STEP_COST = {1: 1.00, 2: 1.02, 4: 1.05, 8: 1.12}               # relative cost of a step with P samples of a ray side by side
UNIT_FIXED = 6.0                                                # us per unit: ray set-up, park / resume, queue atomic


def simulate(classes, prefer_old, prefer_young, n_simd=1024, S=64):
    """classes: [(P, n_units)]; prefer_*: class order each kind of wavefront pulls from.  Returns (makespan us, per-class taken)."""
    left = [n for _, n in classes]
    # per SIMD: remaining steps of the older / younger wavefront's current unit
    rem = [[0.0, 0.0] for _ in range(n_simd)]
    t = [0.0] * n_simd
    done = [False] * n_simd
    end = 0.0
    # event-driven per SIMD, but the queues are global: advance the SIMD with the smallest clock
    heap = [(0.0, i) for i in range(n_simd)]
    heapq.heapify(heap)

    def pull(kind):
        for c in (prefer_old if kind == 0 else prefer_young):
            if left[c] > 0:
                left[c] -= 1
                P = classes[c][0]
                return (S / P) * STEP_COST[P] + UNIT_FIXED / PACE_OLD
        return 0.0

    while heap:
        now, i = heapq.heappop(heap)
        for kind in (0, 1):
            if rem[i][kind] <= 1e-9:
                rem[i][kind] = pull(kind)
        a, b = rem[i]
        if a <= 1e-9 and b <= 1e-9:
            end = max(end, now)
            continue
        if a > 1e-9 and b > 1e-9:                 # both resident: run until one of them finishes its unit
            dt = min(a * PACE_OLD, b * PACE_YOUNG)
            rem[i][0] -= dt / PACE_OLD
            rem[i][1] -= dt / PACE_YOUNG
        else:                                     # one wavefront alone on the SIMD
            k = 0 if a > 1e-9 else 1
            dt = rem[i][k] * PACE_ALONE
            rem[i][k] = 0.0
        heapq.heappush(heap, (now + dt, i))
    return end


def main():
    tiles = int(sys.argv[1]) if len(sys.argv) > 1 else 2303
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    slots = 2048
    ideal = tiles * S / 1024 * 25.9
    print(f"{tiles} tiles x {S} samples; at the steady-state pair rate: {ideal:.0f} us")
    whole = min(tiles, slots)
    plans = {
        "today: one round of whole tiles + the rest as P=8 units": ([(1, whole), (8, (tiles - whole) * 8)], [0, 1], [0, 1]),
        "every tile P=4": ([(4, tiles * 4)], [0], [0]),
        "every tile P=8": ([(8, tiles * 8)], [0], [0]),
    }
    for frac_young in (0.25, 0.375, 0.5):
        for tail_p in (4, 8):
            n_old = 1024
            n_y2 = int(1024 * 1)                              # every younger wavefront starts with one P=2 unit (half a tile's steps)
            t_y2 = n_y2 // 2
            rest = tiles - n_old - t_y2
            if rest < 0:
                continue
            plans[f"older: whole tiles, younger: one P=2 unit, rest P={tail_p}"] = ([(1, n_old), (2, n_y2), (tail_p, rest * tail_p)], [0, 2, 1], [1, 2, 0])
    for y_p, tail_p in ((2, 8), (4, 8), (2, 4)):
        for n_old in (1024, 1280, 1536):
            rest_tiles = tiles - n_old
            n_y = 1024 if y_p == 2 else 2048
            t_y = n_y // y_p
            rest = rest_tiles - t_y
            if rest < 0:
                continue
            plans[f"{n_old} whole tiles (older first), younger {n_y} x P={y_p}, rest P={tail_p}"] = ([(1, n_old), (y_p, n_y), (tail_p, rest * tail_p)], [0, 2, 1], [1, 2, 0])
    for name, (classes, po, py) in plans.items():
        t = simulate(classes, po, py, S=S)
        print(f"  {t:7.0f} us  ({t / ideal:.3f} x ideal)  {name}")


if __name__ == "__main__":
    main()
