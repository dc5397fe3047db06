"""The cone (rustyhgi_amd/csrc/hgi_fused_impl.h, cone_*): pyramids of six to eight levels run as ONE launch at four fused
levels in which every tile rebuilds the levels above it from the few lattice points its own pixels depend on.  This is the
CPU statement of that geometry -- the boxes `[A & ~(s-1) ...] x [B & ~(s-1) ...]` of cone_n() points per level, the corner
indexing between levels, the out-of-plane rule, base samples from the frame or (deeper pyramids) from the stride-256
lattice's planes -- checked against the oracle for every tile position, every tile height and every depth: what the
kernel's index arithmetic has to reproduce.  No GPU needed; the GPU suite checks the kernels themselves."""
import numpy as np
import pytest

TW = 128


def cone_n(tile_px, t):
    """points per dimension at stride 2^t, worst case over tile positions (cone_n() in hgi_fused_impl.h)"""
    c = tile_px >> 4
    n = c + 2
    for i in range(t):
        n = ((n - 1) // 2 if c % (2 << i) == 0 else n // 2) + 2
    return n


def pred(interp, lt, rt, lb, rb):
    if interp == 0:
        return lt                                                   # src/interpolator.rs:26
    left, right = (lt + lb + 1) >> 1, (rb + rt + 1) >> 1             # :46-47
    top, bot = (rt + lt + 1) >> 1, (rb + lb + 1) >> 1                # :48-49
    return (left + right + top + bot) >> 2                          # :51


def cone_seeds(src, W, H, X0, Y0, TH, up, lut, interp, encode, base=None):
    """seeds (bx, by) -> (rec, q) of the four-level tile at (X0, Y0).  src: image (encode) or grid (decode);
    base: None, or (rec plane, q plane) of the stride-2^(4 + up) lattice when the pyramid is deeper than 4 + up levels."""
    sw, sh = ((W - 1) >> 4) + 1, ((H - 1) >> 4) + 1
    A, B = X0 >> 4, Y0 >> 4
    R = {}
    for t in range(up, -1, -1):
        s = 1 << t
        nx, ny = cone_n(TW, t), cone_n(TH, t)
        ox, oy = A & ~(s - 1), B & ~(s - 1)
        cur = {}
        for i in range(nx * ny):
            ix, iy = i % nx, i // nx
            x, y = ox + ix * s, oy + iy * s
            inside = x < sw and y < sh
            v = int(src[y * 16, x * 16]) if inside else 0
            if t == up:
                rec = q = v
                if base is not None and inside:
                    rec, q = int(base[0][y >> up, x >> up]), int(base[1][y >> up, x >> up])
            else:
                s2 = 2 * s
                jx, jy = ((x & ~(s2 - 1)) - (A & ~(s2 - 1))) >> (t + 1), ((y & ~(s2 - 1)) - (B & ~(s2 - 1))) >> (t + 1)
                nx2, ny2 = cone_n(TW, t + 1), cone_n(TH, t + 1)
                assert 0 <= jx and jx + 1 < nx2 and 0 <= jy and jy + 1 < ny2, "corner outside the coarser level's box"
                c00, c01, c10, c11 = R[t + 1][(jx, jy)], R[t + 1][(jx, jy + 1)], R[t + 1][(jx + 1, jy)], R[t + 1][(jx + 1, jy + 1)]
                if not ((x | y) & s):
                    rec, q = c00                                    # a point of the coarser lattice: handed down
                else:
                    p = pred(interp, c00[0], c01[0], c10[0], c11[0])
                    if encode:                                      # src/encoder.rs:53-60
                        d = (v - p) & 255
                        q = int(lut[d])
                        if ((p + q) > 255) != ((p + d) > 255):
                            q = d
                    else:
                        q = v                                       # src/decoder.rs:40
                    rec = (p + q) & 255
            if not inside:
                rec = q = 0
            cur[(ix, iy)] = (rec, q)
        R[t] = cur
    return R[0]


def test_cone_point_counts_fit_a_wave():
    assert [cone_n(TW, t) for t in range(5)] == [10, 6, 4, 3, 3]
    for th, want in ((64, [6, 4, 3, 3, 3]), (32, [4, 3, 3, 3, 3]), (16, [3, 3, 3, 3, 3])):
        assert [cone_n(th, t) for t in range(5)] == want
        assert cone_n(TW, 0) * cone_n(th, 0) <= 64                                     # the seeds: one lane each
        assert sum(cone_n(TW, t) * cone_n(th, t) for t in range(1, 5)) <= 64           # the levels above them: one lane each
        assert sum(cone_n(TW, t) * cone_n(th, t) for t in range(1, 5)) <= 80           # their byte arrays: two halo-column slots


@pytest.mark.parametrize("TH", [64, 32, 16])
def test_cone_reproduces_the_oracle_at_every_tile(oracle, TH):
    rng = np.random.default_rng(TH)
    checked = 0
    for trial in range(16):
        W, H = int(rng.integers(130, 900)), int(rng.integers(70, 700))
        up = 1 + trial % 4
        extra = int(rng.integers(0, 3)) if trial % 3 == 0 else 0      # pyramids deeper than 4 + up: base from the lattice's planes
        levels = 4 + up + extra
        interp = trial % 2
        img = rng.integers(0, 256, (H, W), dtype=np.uint8)
        if trial % 2:
            img = (np.add.outer(np.arange(H) * 3, np.arange(W) * 5) // 7 + rng.integers(0, 9, (H, W))).astype(np.uint8)
        lut = oracle.linear_lut(int(rng.integers(0, 4)))[0] if trial % 4 else rng.integers(0, 256, 256, dtype=np.uint8)
        grid, rec, _ = oracle.encode(img, levels, lut, interp, want_rec=True)
        base = None
        if extra:
            k = 4 + up
            base = (rec[::1 << k, ::1 << k], grid[::1 << k, ::1 << k])
        for Y0 in range(0, H, TH):
            for X0 in range(0, W, TW):
                for enc in (True, False):
                    seeds = cone_seeds(img if enc else grid, W, H, X0, Y0, TH, up, lut, interp, enc, base)
                    for (bx, by), (r, q) in seeds.items():
                        x, y = X0 + 16 * bx, Y0 + 16 * by
                        if x < W and y < H:
                            assert r == rec[y, x], (W, H, levels, enc, X0, Y0, bx, by)
                            if enc:
                                assert q == grid[y, x], (W, H, levels, X0, Y0, bx, by)
                            checked += 1
                        else:
                            assert r == 0
    assert checked > 20000
