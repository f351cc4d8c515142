#!/usr/bin/env python3
"""Statistics of the walk's per-ray entry lists on the bench scene (CPU, numpy; approximate DDA in float64 -- counts only):
cells per ray, entries per ray, and per wave of 64 consecutive rays what phase 2 iterates over with 16-slot lists.
    python scripts/entry_stats.py [--res 128] [--waves 400]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--res", type=int, default=128)
ap.add_argument("--waves", type=int, default=400)
ap.add_argument("--slots", type=int, default=16)
args = ap.parse_args()
res = args.res
grid = bench.make_grid(res, "shell10")[0]
o, d = bench.make_rays(1024 * 1024, "image")
rng = np.random.default_rng(1)
wave_ids = rng.choice(16384, args.waves, replace=False)
idx = (wave_ids[:, None] * 64 + np.arange(64)[None]).reshape(-1)
o = o[idx].astype(np.float64); d = d[idx].astype(np.float64)
inv = 1.0 / d
t1 = (-1 - o) * inv; t2 = (1 - o) * inv
tmin = np.maximum(np.minimum(t1, t2).max(-1), 0.0); tmax = np.maximum(t1, t2).min(-1)
hit = tmax > tmin
n = len(idx)
cells = np.zeros(n, int); entries = np.zeros(n, int)
# per ray: list of entry closing positions as cell counts (for the per-round simulation we only need the order of flips)
flips_at = [[] for _ in range(n)]
vox = 2.0 / res
p = o + d * (tmin[:, None] + 1e-6)
cur = np.clip(((p + 1) / 2 * res).astype(int), 0, res - 1)
pe = o + d * (tmax[:, None] - 1e-6)
fin = np.clip(((pe + 1) / 2 * res).astype(int), 0, res - 1)
step = np.sign(d).astype(int)
nxt = (cur + (step > 0)) * vox - 1
tdist = np.where(d != 0, (nxt - o) * inv, np.inf)
delta = np.where(d != 0, vox * np.abs(inv), np.inf)
alive = hit.copy()
prev = np.full(n, -1)
ar = np.arange(n)
while alive.any():
    occ = grid[cur[:, 0], cur[:, 1], cur[:, 2]].astype(int)
    flip = alive & (occ != prev) & (prev >= 0)
    for i in np.nonzero(flip)[0]:
        flips_at[i].append(cells[i])
    prev = np.where(alive, occ, prev)
    cells += alive
    ax = np.argmin(tdist, -1)
    done = cur[ar, ax] == fin[ar, ax]
    cur[ar, ax] += np.where(alive, step[ar, ax], 0)
    tdist[ar, ax] += delta[ar, ax]
    alive &= ~done
    alive &= ((cur >= 0) & (cur < res)).all(-1)
    cur = np.clip(cur, 0, res - 1)
entries = np.array([len(f) + 1 if c > 0 else 0 for f, c in zip(flips_at, cells)])
print(f"rays {n}: hit {hit.mean():.3f}  cells/ray mean {cells.mean():.1f} (hit rays {cells[hit].mean():.1f}) max {cells.max()}   entries/ray mean {entries.mean():.2f} (hit {entries[hit].mean():.2f}) max {entries.max()}")
# phase-2 rounds with S-slot lists: a span start takes two slots, then the entries; a lane stops when its list is full; the wave
# runs phase 2 when every lane has stopped (list full or ray finished)
S = args.slots
rows_total = 0; ent_total = 0; rounds_total = 0; dense_total = 0; cellrows = 0; celltot = 0
for w in range(args.waves):
    e = entries[w * 64:(w + 1) * 64].copy()
    c = cells[w * 64:(w + 1) * 64]
    cellrows += c.max(); celltot += c.sum()
    first = True
    rem = e.copy()
    while (rem > 0).any():
        cap = S - (2 if first else 0)     # slots for entries in this round (the open entry carried over occupies slot 0: ignore)
        take = np.minimum(rem, cap - (0 if first else 1))
        slots = take + (2 if first else 0)
        slots = np.where(rem > 0, slots, 0)
        rows_total += slots.max(); ent_total += slots.sum(); rounds_total += 1
        dense_total += -(-slots.sum() // 64)
        rem -= take
        first = False
W = args.waves
print(f"per wave: rounds {rounds_total / W:.2f}  lock-step rows {rows_total / W:.1f}  slots used {ent_total / W:.0f}  lane utilisation {ent_total / (64 * rows_total):.2f}  dense iterations {dense_total / W:.1f}")
print(f"per wave: cell-loop trips (max over lanes) {cellrows / W:.1f}  lane utilisation of the cell loop {celltot / (64 * cellrows):.2f}")
