"""r6 host logic without a device: the item table of the persistent inverse-transform launch (wg3_item_table, k_idct_wg3.hip) through the
library's CPU test hook jxl_debug_wg3_item_table -- every item exactly once, a workgroup's list ends in holes and nowhere else, special 8x8
items behind the METHOD_DCT ones and 64x64 blocks last, the lists balanced by cost, and the plain form for grids the balancing does not
take. (What the kernels do with the table is the -m gpu suite's business: tests/test_idct_items_gpu.py.)"""
import ctypes as C

import numpy as np
import pytest

from jxlatte_amd import _lib, abi

NAME2TYPE = {t[0]: i for i, t in enumerate(abi.TRANSFORM_TYPES)}
SPECIAL = {1, 2, 3, 12, 13, 14, 15, 16, 17}
COST = {0: 55, 4: 76, 5: 100, 6: 67, 7: 67, 8: 83, 9: 83, 10: 90, 11: 90, 18: 330}  # wg3_item_cost (relative; specials 70)


def _table(types, counts, frame_bw, grid):
    lib = _lib.load()
    f = lib.jxl_debug_wg3_item_table
    f.restype = C.c_int
    f.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_int]
    t = np.asarray(types, np.int32)
    n = np.asarray(counts, np.int32)
    cap = int(sum(counts)) + 4 * max(grid, 8) * 64
    out = np.zeros(cap * 8, np.int32)
    got = f(t.ctypes.data_as(C.POINTER(C.c_int32)), n.ctypes.data_as(C.POINTER(C.c_int32)), len(types), frame_bw, grid,
            out.ctypes.data_as(C.POINTER(C.c_int32)), cap)
    assert got >= 0
    return out[:got * 8].reshape(got, 8)


def _blocks_per_item(t):
    ph, pw = abi.tt_pixel_size(t)
    return 1 if t == 18 else 2048 // (ph * pw)


def _expected_items(types, counts):
    exp, first = set(), 0
    for t, n in zip(types, counts):
        nb = _blocks_per_item(t)
        for o in range(0, n, nb):
            exp.add((t, first + o, min(nb, n - o)))
        first += n
    return exp


def _default_mix_4k():
    """block counts of a 3840 x 2160 frame with about the default mix's area shares"""
    cells = (3840 // 8) * (2160 // 8)
    share = {"DCT8": 0.40, "DCT16": 0.15, "DCT32": 0.10, "DCT16_8": 0.05, "DCT8_16": 0.05, "DCT32_8": 0.02, "DCT8_32": 0.02, "DCT32_16": 0.02,
             "DCT16_32": 0.02, "DCT64": 0.05, "DCT4": 0.015, "DCT4_8": 0.015, "DCT8_4": 0.015, "DCT2": 0.015, "HORNUSS": 0.012, "AFV0": 0.012,
             "AFV1": 0.012, "AFV2": 0.012, "AFV3": 0.012}
    types, counts = [], []
    for name, s in share.items():
        t = NAME2TYPE[name]
        ph, pw = abi.tt_pixel_size(t)
        types.append(t)
        counts.append(max(1, int(cells * s / ((ph // 8) * (pw // 8)))))
    return types, counts


@pytest.mark.parametrize("grid", [768, 512, 64, 8])
def test_every_item_once_and_holes_only_at_the_ends(grid):
    types, counts = _default_mix_4k()
    tab = _table(types, counts, 480, grid)
    assert len(tab) % grid == 0
    items = [(int(r[0]), int(r[1]), int(r[2])) for r in tab if r[0] != -2]
    assert len(items) == len(set(items))
    assert set(items) == _expected_items(types, counts)
    rounds = len(tab) // grid
    load = np.zeros(grid)
    for w in range(grid):
        kinds = []
        seen_hole = False
        for k in range(rounds):
            r = tab[k * grid + w]
            if r[0] == -2:
                seen_hole = True
                continue
            assert not seen_hole, "an item behind a hole in workgroup %d's list" % w
            t = int(r[0])
            kinds.append(2 if t == 18 else 1 if t in SPECIAL else 0)
            load[w] += COST.get(t, 70)
        assert kinds == sorted(kinds), "workgroup %d: METHOD_DCT items, then special 8x8 items, then 64x64 blocks" % w
    if grid >= 64:
        assert load.max() <= 1.2 * load.mean() and load.min() >= 0.8 * load.mean(), (load.min(), load.mean(), load.max())


def test_geometry_words_and_the_64x64_flag():
    tab = _table([NAME2TYPE["DCT8"], NAME2TYPE["AFV0"], NAME2TYPE["DCT64"]], [400, 100, 16], 128, 64)
    for r in tab:
        if r[0] == -2:
            continue
        geo, t = int(r[3]) & 0xffffffff, int(r[0])
        flip = (geo >> 15) & 1
        assert ((geo >> 14) & 1) == (1 if t == 18 else 0)          # Item64 fetches its own coefficients
        assert flip == (0 if t in SPECIAL else 1)                  # TransformType.flip(): square AND METHOD_DCT (TransformType.java:129-131)
        assert (geo >> 16) & 0xff == abi.tt_param_index(t)


@pytest.mark.parametrize("grid", [0, 5, 40])
def test_grids_the_balancing_does_not_take(grid):
    """no grid, fewer than 8 workgroups or not a multiple of the number of queues: a plain list, every item once, no holes"""
    types, counts = [NAME2TYPE["DCT16"], NAME2TYPE["DCT2"], NAME2TYPE["DCT64"]], [500, 300, 7]
    tab = _table(types, counts, 256, grid)
    if grid % 8:
        assert not (tab[:, 0] == -2).any()
    items = {(int(r[0]), int(r[1]), int(r[2])) for r in tab if r[0] != -2}
    assert items == _expected_items(types, counts)
