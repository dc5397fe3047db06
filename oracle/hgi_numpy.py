"""Second, independent CPU restatement of the HGI hot path in numpy.

TEST INFRASTRUCTURE ONLY (same rules as hgi_oracle.c).  PARITY UNPINNED by the
reference's own tests; this file exists so that two structurally different
readings of the reference (scalar traversal in C, whole-level array algebra
here) have to agree bit for bit.

Structure: per level, all pixels of the level are produced at once from the
stride-`step` corner lattice.  Citations are to
/root/reference.
"""
import numpy as np

LEFTTOP, CROSSED = 0, 1


def linear_lut(level):
    """src/quantizator.rs:41-63."""
    error = {0: 0, 1: 10, 2: 20, 3: 30}[int(level)]
    scale = 2 * error + 1
    i = np.arange(256, dtype=np.int64)
    return (((i + error) // scale) * scale).astype(np.uint8), error


def _corner_lattice(img, step):
    """Values at (i*step, j*step) for i <= ceil(W/step), j <= ceil(H/step); OOB -> 0
    (src/interpolator.rs:75-82)."""
    h, w = img.shape
    ny, nx = -(-h // step) + 1, -(-w // step) + 1
    lat = np.zeros((ny, nx), np.int64)
    sub = img[::step, ::step]
    lat[: sub.shape[0], : sub.shape[1]] = sub
    return lat


def _cell_prediction(img, step, interp):
    """One prediction per step-cell (src/interpolator.rs:41-55, 57-90; 15-28)."""
    lat = _corner_lattice(img, step)
    lt = lat[:-1, :-1]          # (x0, y0)          left_top   :85
    rt = lat[1:, :-1]           # (x0, y0+step)     right_top  :86
    lb = lat[:-1, 1:]           # (x0+step, y0)     left_bot   :87
    rb = lat[1:, 1:]            # (x0+step, y0+step) right_bot :88
    if interp == LEFTTOP:
        return lt.astype(np.uint8)
    avg = lambda a, b: (a + b + 1) >> 1
    left, right = avg(lt, lb), avg(rb, rt)
    top, bot = avg(rt, lt), avg(rb, lb)
    return ((left + right + top + bot) >> 2).astype(np.uint8)


def _level_views(arr, sub):
    """The three new-pixel families of a level as strided views (src/utils.rs:12-41):
    (x0+sub, y0), (x0, y0+sub), (x0+sub, y0+sub)."""
    step = 2 * sub
    return (arr[0::step, sub::step], arr[sub::step, 0::step], arr[sub::step, sub::step])


def encode(img, levels, lut, interp=CROSSED, want_rec=False):
    """src/encoder.rs:39-71."""
    img = np.ascontiguousarray(img, np.uint8)
    lut = np.asarray(lut, np.uint8)
    rec = img.copy()
    grid = np.zeros_like(img)
    nfb = 0
    if img.size == 0:
        return (grid, rec, 0) if want_rec else grid
    b = 1 << levels
    grid[::b, ::b] = img[::b, ::b]                                   # :26-37
    for level in range(levels):
        step = 1 << (levels - level)
        sub = step >> 1
        pred = _cell_prediction(rec, step, interp)
        for rv, gv in zip(_level_views(rec, sub), _level_views(grid, sub)):
            p = pred[: rv.shape[0], : rv.shape[1]]
            a = rv.copy()
            d = a - p                                                # :53 wrapping (uint8)
            q = lut[d]                                               # :54
            ovf = (p.astype(np.int64) + q) > 255                     # :56
            exp = (p.astype(np.int64) + d) > 255                     # :57
            fb = ovf != exp
            q = np.where(fb, d, q)                                   # :58-60
            nfb += int(fb.sum())
            gv[...] = q                                              # :62
            rv[...] = p + q                                          # :63-64 wrapping
    return (grid, rec, nfb) if want_rec else grid


def decode(grid, levels, interp=CROSSED):
    """src/decoder.rs:18-46."""
    grid = np.ascontiguousarray(grid, np.uint8)
    out = np.zeros_like(grid)
    if grid.size == 0:
        return out
    b = 1 << levels
    out[::b, ::b] = grid[::b, ::b]                                   # :22-28
    for level in range(levels):
        step = 1 << (levels - level)
        sub = step >> 1
        pred = _cell_prediction(out, step, interp)
        for ov, gv in zip(_level_views(out, sub), _level_views(grid, sub)):
            ov[...] = pred[: ov.shape[0], : ov.shape[1]] + gv        # :39-40 wrapping
    return out
