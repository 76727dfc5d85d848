#!/usr/bin/env python3
"""Host-side emulation (NumPy, no GPU) of which wave-steps of atrous_lds_kernel take the uniform-normal tap path on bench.py's headline scene at 3840x2160 —
the launch geometry of svgf_atrous_lds.h (128-column tiles, row residues, bands from cut_bands, the 6-row ring window, the workgroup's reference normal) —
under the rule the kernel implements (A) and under three finer rules nobody built:
  A  every counting texel of the ring window (6 decimated rows x 128 + 4S columns) carries the workgroup's reference normal (the kernel; the device counters of
     svgf_path_stats_enable read 0.8575 / 0.8478 / 0.8314 / 0.8001 / 0.6681 at steps 1-16: this emulation reproduces them to the last digit)
  B  the same per 64-column half of the tile (the columns one wave's taps reach)
  C  all counting texels of the window carry ONE normal, whichever (no fixed reference: a band that crosses into another planar region recovers)
  D  B and C together
Output: profiles/r06_small_experiments.txt block 3.   python3 tools/uniform_share_emul.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from svgf_amd import synth


def emulate(W=3840, H=2160, steps=(1, 2, 4, 8, 16), scene="planar", verbose=False):
    """-> {step: {rule: share of the wave-steps that hold a surface pixel and would take the uniform-normal path}}."""
    sc = synth.make_scene(W, H, 0, scene=scene)
    n = sc["normal"].astype(np.uint64)
    key = n[..., 0] | (n[..., 1] << 16) | (n[..., 2] << 32)
    depth = sc["motion"][..., 2]
    counting = (depth != 0) | (key != 0)
    surface = depth != 0
    BIG = np.uint64(1 << 60)
    kmin = np.where(counting, key, BIG); kmax = np.where(counting, key, np.uint64(0))
    TX = 128
    xt = (W + TX - 1) // TX
    def colwin(lo_fn, hi_fn):
        mn = np.empty((H, xt), np.uint64); mx = np.empty((H, xt), np.uint64)
        for t in range(xt):
            a, b = max(0, lo_fn(t)), min(W, hi_fn(t))
            mn[:, t] = kmin[:, a:b].min(1); mx[:, t] = kmax[:, a:b].max(1)
        return mn, mx
    def band_rows(S):
        per_cu = 5 if S <= 8 else 4
        njmax = (H + S - 1) // S
        nb = max(1, per_cu * 256 * 4 // (xt * S))
        band = max(8, (njmax + nb - 1) // nb); band = (band + 1) // 2 * 2
        return band
    res = {}
    for S in steps:
        band = band_rows(S)
        wins = {"wg": colwin(lambda t: t*TX - 2*S, lambda t: t*TX + TX + 2*S),
                "h0": colwin(lambda t: t*TX - 2*S, lambda t: t*TX + 64 + 2*S),
                "h1": colwin(lambda t: t*TX + 64 - 2*S, lambda t: t*TX + TX + 2*S)}
        # per half: has a surface centre in (row, tile, half)
        surf = [np.stack([surface[:, t*TX + 64*c: t*TX + 64*c + 64].any(1) for t in range(xt)], 1) for c in (0, 1)]
        tot = 0; cnt = {"A": 0, "B": 0, "C": 0, "D": 0}
        for r in range(S):
            rows = np.arange(r, H, S)
            nj = len(rows)
            def rowwin(mn, mx):   # window j-2..j+3 for step pairs (j even within band)
                pad_mn = np.full((nj + 6, xt), BIG, np.uint64); pad_mx = np.zeros((nj + 6, xt), np.uint64)
                pad_mn[2:2+nj] = mn[rows]; pad_mx[2:2+nj] = mx[rows]
                return pad_mn, pad_mx
            P = {k: rowwin(*v) for k, v in wins.items()}
            for j0 in range(0, nj, band):
                j1 = min(nj, j0 + band)
                refrow = rows[j0]
                ref = np.array([key[refrow, t*TX] if True else 0 for t in range(xt)], np.uint64)
                refcount = np.array([counting[refrow, t*TX] for t in range(xt)])
                for j in range(j0, j1, 2):
                    def uni(name, fixed_ref):
                        mn = P[name][0][j:j+6].min(0); mx = P[name][1][j:j+6].max(0)   # rows j-2..j+3 (padded index j..j+5)
                        empty = mx == 0
                        same = mn == mx
                        if fixed_ref:
                            return empty | (same & (mn == ref))
                        return empty | same
                    uA = uni("wg", True); uC = uni("wg", False)
                    uB = [uni("h0", True), uni("h1", True)]; uD = [uni("h0", False), uni("h1", False)]
                    for rr in (j, j + 1):
                        if rr >= j1: continue
                        for c in (0, 1):
                            s = surf[c][rows[rr]]
                            tot += s.sum()
                            cnt["A"] += (s & uA).sum(); cnt["C"] += (s & uC).sum()
                            cnt["B"] += (s & uB[c]).sum(); cnt["D"] += (s & uD[c]).sum()
        res[S] = {k: round(float(v) / float(tot), 4) for k, v in cnt.items()}
        if verbose:
            print(S, band, int(tot), res[S], flush=True)
    return res


if __name__ == "__main__":
    emulate(verbose=True)
