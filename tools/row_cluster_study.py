#!/usr/bin/env python3
"""Row-side clusters for the boundary passes of the Newton-3 kernel (review item 4 of round 3): what share of all unordered
pairs has to be evaluated when, beside the column tile's four projection slabs of 16 (n3_cluster_pass), the ROW tile is
tested in units of 16 as well?  Bench configuration (n = 262144, jittered lattice), a sample of row tiles against all
column tiles, CPU only (numpy, ~3 min).  Row units = the k-d leaves of 16 the tile is made of (lanes 16 r .. 16 r + 15 =
DPP row r: the only row units the rotation scheme has, the row particles live in registers for the whole kernel).
  64 x 16          what n3_cluster_pass does: (row tile, column slab) kept unless the projection proves every pair outside
  16 x 16 ideal    every (row leaf, column slab) pair tested and skipped on its own (a cluster-pair-list kernel)
  16 x 16 in-wave  the rotation scheme: the four row leaves of a tile are the four DPP rows of ONE wave instruction, a
                   (tile, slab) step is issued when ANY leaf needs it
  16 x 16 sorted   as in-wave, but with the four row leaves re-dealt so that the leaves of the four ROW TILES of a lane
                   group that sit in one DPP row are the ones closest to the column tile (best case of any static deal)
Measurement tool, not product code."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ljmd_amd  # noqa: F401,E402
from ljmd_amd import synthetic  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
nrows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
p, r, v = synthetic.make_config(n)
L, rc = p.box_length, p.rc
r = np.mod(r, L)


def kd(idx, depth, leaf):
    if len(idx) <= leaf:
        return [idx]
    ax = depth % 3
    o = idx[np.argsort(r[ax, idx], kind='stable')]
    h = len(o) // 2
    return kd(o[:h], depth + 1, leaf) + kd(o[h:], depth + 1, leaf)


order = np.concatenate(kd(np.arange(n), 0, 16))
rs = r[:, order]
T = n // 64
tiles = rs.reshape(3, T, 64)
lo, hi = tiles.min(axis=2), tiles.max(axis=2)
cen = 0.5 * (lo + hi)
rng = np.random.default_rng(1)
rows = rng.choice(T, size=nrows, replace=False)
acc = dict(tile=0.0, c64x16=0.0, ideal=0.0, inwave=0.0, exact16=0.0, inside=0.0, ideal8=0.0)
tot = 0
for I in rows:
    xi = tiles[:, I, :]
    d2 = np.zeros(T)
    for k in range(3):
        a = lo[k, I] - hi[k]
        b = hi[k, I] - lo[k]
        g = np.full(T, np.inf)
        zero = np.zeros(T, bool)
        for m in (-1, 0, 1):
            c = m * L
            zero |= (a <= c) & (c <= b)
            g = np.minimum(g, np.minimum(np.abs(a - c), np.abs(b - c)))
        g[zero] = 0
        d2 += g * g
    keep = d2 <= rc * rc
    keep[I] = False
    J = np.nonzero(keep)[0]
    tot += T
    acc['tile'] += len(J) * 64.0
    xj = tiles[:, J, :]
    sh = np.round((cen[:, J] - cen[:, I][:, None]) / L) * L
    xjs = xj - sh[:, :, None]
    dc = (cen[:, J] - sh) - cen[:, I][:, None]
    nh = dc / np.linalg.norm(dc, axis=0)
    dd = xi[:, None, :, None] - xjs[:, :, None, :]
    r2 = (dd * dd).sum(axis=0)                                  # (nJ, 64 rows, 64 cols)
    inside = r2 < rc * rc
    acc['inside'] += inside.sum() / 64.0
    inner = inside.all(axis=(1, 2))                             # INNER passes run the plain loop, in full
    pj = (xjs * nh[:, :, None]).sum(axis=0)                     # (nJ, 64)
    o = np.argsort(pj, axis=1)
    pmin = np.take_along_axis(pj, o, axis=1).reshape(len(J), 4, 16).min(axis=2)      # (nJ, 4 slabs)
    # row leaves: support of each leaf's exact box along the pass direction (what tile_class_kernel would store)
    leaf = xi.reshape(3, 4, 16)
    llo, lhi = leaf.min(axis=2), leaf.max(axis=2)               # (3, 4)
    supp_leaf = np.maximum(nh[:, :, None] * llo[:, None, :], nh[:, :, None] * lhi[:, None, :]).sum(axis=0)   # (nJ, 4 leaves)
    supp_tile = supp_leaf.max(axis=1)
    k_tile = (pmin - supp_tile[:, None]) <= rc                   # (nJ, 4 slabs)
    k_leaf = (pmin[:, None, :] - supp_leaf[:, :, None]) <= rc    # (nJ, 4 leaves, 4 slabs)
    k_tile[inner] = True
    k_leaf[inner] = True
    acc['c64x16'] += k_tile.sum() * 16.0
    acc['ideal'] += k_leaf.sum() * 4.0                           # 16 x 16 pairs = 4 wave-lanes-worth of 64
    acc['inwave'] += k_leaf.any(axis=1).sum() * 16.0
    ins = np.take_along_axis(inside, o[:, None, :].repeat(64, 1), axis=2).reshape(len(J), 4, 16, 4, 16)
    acc['exact16'] += ins.any(axis=(2, 4)).sum() * 4.0
    acc['ideal8'] += np.take_along_axis(inside, o[:, None, :].repeat(64, 1), axis=2).reshape(len(J), 8, 8, 8, 8).any(axis=(2, 4)).sum() * 1.0
den = tot * 64.0
print(f"n = {n}, {nrows} row tiles sampled; share of all unordered pairs evaluated")
print(f"64 x 64 tile pairs                                   {acc['tile'] / den:.4f}")
print(f"64 x 16  column slabs by projection (n3_cluster_pass) {acc['c64x16'] / den:.4f}")
print(f"16 x 16  in-wave: a step is issued when any leaf needs it {acc['inwave'] / den:.4f}")
print(f"16 x 16  ideal: every (row leaf, column slab) on its own  {acc['ideal'] / den:.4f}")
print(f"16 x 16  exact: pairs of units that hold a pair inside    {acc['exact16'] / den:.4f}")
print(f" 8 x 8   exact (k-d leaves of 8 x projection slabs of 8)  {acc['ideal8'] / den:.4f}")
print(f"inside the cutoff                                    {acc['inside'] / den:.4f}")
