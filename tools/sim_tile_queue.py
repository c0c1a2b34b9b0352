"""Schedule model of the fused kernel's tile queue under early termination.

Input: the per-tile number of evaluated steps of a frame (bench.py with GPNERF_DUMP_DONE=path) and the measured step time of a
wave against the number of waves active on its CU (tools/gpu_chain_probe.sh, GPNERF_WAVE_CAP sweep of the headline frame).
Output: the frame time of (a) whole-ray work items, (b) FIFO items of L steps -- what the queue could reach with no overhead.
"""
import sys
import numpy as np

STEP_US = {1: 29.6, 2: 30.6, 3: 31.5, 4: 31.5, 5: 41.8, 6: 46.3, 7: 53.5, 8: 55.2}   # MI355X, fp32 form, round 2


def simulate(lengths, item, n_cu=256, waves=8, dt=2.0, gate=True):
    rate = np.array([0.0] + [1.0 / STEP_US[w] for w in range(1, waves + 1)])
    fresh = list(range(len(lengths)))[::-1]            # pop() takes them in order
    fifo = []                                          # (tile) entries, consumed from the front
    head = 0
    done = np.zeros(len(lengths))                      # steps walked per tile
    cur = -np.ones((n_cu, waves), dtype=np.int64)      # tile on each wave
    left_in_item = np.zeros((n_cu, waves))
    unfinished = len(lengths)
    t = 0.0
    while unfinished > 0:
        # idle waves pull work
        idle = np.argwhere(cur < 0)
        if len(idle):
            order = np.lexsort((idle[:, 0], idle[:, 1]))      # low wave index first: spreads over the CUs
            for cu, w in idle[order]:
                if gate and not fresh and unfinished <= w * n_cu:
                    continue
                if fresh:
                    tile = fresh.pop()
                elif head < len(fifo):
                    tile = fifo[head]; head += 1
                else:
                    break
                cur[cu, w] = tile
                left_in_item[cu, w] = min(item, lengths[tile] - done[tile])
        active = (cur >= 0)
        n_act = active.sum(1)
        if n_act.sum() == 0:
            break
        prog = rate[n_act][:, None] * dt * active
        left_in_item -= prog
        fin = active & (left_in_item <= 0)
        for cu, w in np.argwhere(fin):
            tile = cur[cu, w]
            done[tile] = min(lengths[tile], done[tile] + item)
            if done[tile] >= lengths[tile]:
                unfinished -= 1
            else:
                fifo.append(tile)
            cur[cu, w] = -1
        t += dt
    return t / 1e3


if __name__ == "__main__":
    lengths = np.load(sys.argv[1]).astype(np.float64)
    total = lengths.sum()
    print(f"{len(lengths)} tiles, {total:.0f} steps, mean {lengths.mean():.1f}, max {lengths.max():.0f}; "
          f"saturated (8 waves/CU): {total * STEP_US[8] / 2048 / 1e3:.2f} ms")
    print("whole-ray items:", round(simulate(lengths, 1 << 20, gate=False), 2), "ms")
    for item in (64, 32, 16, 8):
        print(f"FIFO items of {item}:", round(simulate(lengths, item), 2), "ms (gated)", round(simulate(lengths, item, gate=False), 2), "ms (ungated)")
