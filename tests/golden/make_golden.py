"""Generates the golden fixtures in tests/golden/ from the CPU oracle (oracle/lfi_oracle.c), refusing to write
anything the independent numpy restatement (oracle/lfi_oracle_np.py) disagrees with.

PARITY UNPINNED: the reference ships no golden vectors and cannot be run in this pipeline, so these fixtures pin the
oracle's behaviour (and through it the HIP kernels'), not the reference binary's.  Parameters follow the reference's
README example (-t 0,0,1,1 -a 1.783 -f 0.23, default -s 3) and scripts/focusMapCompare.sh (-s 7, -t 0.071…0.93).

Run from the repository root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import GOLDEN_DIR, SEED, SMALL_CASES  # noqa: E402
from oracle import lfi_oracle_c as oc  # noqa: E402
from oracle import lfi_oracle_np as on  # noqa: E402


def main():
    oc.build()
    for name, cols, rows, W, H, V, traj, focus, aspect, effect in SMALL_CASES:
        n = cols * rows
        se = oc.interpret_trajectory(traj, cols, rows)
        assert (se == on.interpret_trajectory(traj, cols, rows)).all()
        w = oc.weight_matrix_f16(se, cols, rows, V, effect)
        assert (w == on.weight_matrix_f16(se, cols, rows, V, effect)).all(), "weights: C vs numpy"
        off, foc = oc.offsets(se, cols, rows, W, H, aspect, focus)
        off_n, foc_n = on.offsets(se, cols, rows, W, H, aspect, focus)
        assert (off == off_n).all() and (foc == foc_n).all()
        ids = oc.focus_map_ids(se, cols, rows)
        assert (ids == on.focus_map_ids(se, cols, rows)).all()
        radius = oc.block_radius(W, H)
        lf = oc.synthetic_lf(n, W, H, SEED)
        assert (lf == on.synthetic_lf(n, W, H, SEED)).all()
        std = oc.blend_std(lf, foc, off, w)
        assert (std == on.blend_std(lf, foc, off, w)).all(), "STD: C vs numpy"
        m16 = oc.blend_ten(lf, foc, off, w, model=oc.TEN_M16)
        assert (m16 == on.blend_ten(lf, foc, off, w, model=on.TEN_M16)).all(), "TEN M16: C vs numpy"
        exact = oc.blend_ten(lf, foc, off, w, model=oc.TEN_EXACT)
        assert (exact == on.blend_ten(lf, foc, off, w, model=on.TEN_EXACT)).all(), "TEN exact: C vs numpy"
        # all-focus leg: focus map from the same grid, range from focusMapCompare.sh's table
        rng = 0.18
        map0 = oc.focus_estimate(lf, off, ids, focus, rng, radius)
        assert (map0 == on.focus_estimate(lf, off, ids, focus, rng, radius)).all(), "map0: C vs numpy"
        map1 = oc.focus_filter(map0, radius)
        assert (map1 == on.focus_filter(map0, radius)).all(), "map1: C vs numpy"
        af_std = oc.blend_std(lf, foc, off, w, all_focus=True, map_plane=map1, focus=focus, rng=rng)
        assert (af_std == on.blend_std(lf, foc, off, w, all_focus=True, map_plane=map1, focus=focus, rng=rng)).all()
        af_m16 = oc.blend_ten(lf, foc, off, w, model=oc.TEN_M16, all_focus=True, map_plane=map1, focus=focus, rng=rng)
        assert (af_m16 == on.blend_ten(lf, foc, off, w, model=on.TEN_M16, all_focus=True, map_plane=map1, focus=focus,
                                       rng=rng)).all()
        # Tensors::process<true> reads the UNFILTERED map 0 (reference src/kernels.cu:430), Standard::process map 1 (:326): the
        # library's default reproduces that; af_ten_m16 (map 1) is what LFI_FLAG_UNIFIED_FOCUS_MAP renders
        af_m16_map0 = oc.blend_ten(lf, foc, off, w, model=oc.TEN_M16, all_focus=True, map_plane=map0, focus=focus, rng=rng)
        assert (af_m16_map0 == on.blend_ten(lf, foc, off, w, model=on.TEN_M16, all_focus=True, map_plane=map0, focus=focus,
                                            rng=rng)).all()
        path = os.path.join(GOLDEN_DIR, name + ".npz")
        np.savez_compressed(path, lf=lf, weights=w, offsets=off, focused=foc, ids=ids, radius=radius,
                            std=std, ten_m16=m16, ten_exact=exact, range=np.float32(rng), map0=map0, map1=map1,
                            af_std=af_std, af_ten_m16=af_m16, af_ten_m16_map0=af_m16_map0)
        print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
