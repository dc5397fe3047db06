#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/ (run in the build container).

Inputs that are DATA of the reference (image assets under /root/reference/res)
are converted once into loader-independent raw form; expected outputs come from
the C oracle (oracle/hgi_oracle.c), cross-checked here against the independent
numpy restatement (oracle/hgi_numpy.py) before anything is written.

  lena_256.u8        raw 256x256 luma of res/LENA.TIF (uncompressed TIFF, loader independent)
  fullhd_luma.png    1920x1080 luma of res/fullhd.jpg decoded by PIL convert('L')
                     -- INPUT PARITY UNPINNED: the reference decodes JPEG with the `image`
                     crate (different IDCT / luma weights); this file *is* config C1's input.
  fullhd_luma709.png the same JPEG decoded by PIL to RGB, luma taken the way image-0.19's to_luma does it (truncated f32
                     0.2126 R + 0.7152 G + 0.0722 B -- the formula the docs pair below confirmed on 625 / 625 lattice
                     points).  Closer to what `hgi test res/fullhd.jpg` feeds the encoder (src/main.rs:42, 74); the JPEG
                     IDCT still differs from the crate's, so C1's input stays UNPINNED -- C1 is run on both planes.
  docs_lena_pair.npz the one input/output pair the reference itself holds: docs/static_files/lena_source.png
                     ("source", README.md:6) and lena_hgi.png ("HGI compressed (low)", README.md:8-9), 400x400, as luma
                     planes.  The source is a grey palette PNG; its luma is taken the way image-0.19's to_luma does it
                     (truncated f32 0.2126 R + 0.7152 G + 0.0722 B), under which the stride-16 base lattice of the
                     two planes is identical.  The output was made by ANOTHER REVISION of the algorithm (it is not
                     bit-reproducible by the current source under any predictor rounding tried), so it pins
                     properties only: tests/test_oracle.py::test_reference_held_docs_pair.
  small_cases.npz    inputs + full expected grids/reconstructions for the tiny cases
  golden.json        sha256 / fallback count / max error for every case (incl. big synthetic)
"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import hgi_numpy as NP  # noqa: E402
from oracle import hgi_oracle as O  # noqa: E402

REF = "/root/reference"
SEED0 = 0x48474930


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def xy(w, h):
    return O.synth(O.SYNTH_XY, 0, 0, w, h)


def lut_for(q):
    return O.noop_lut() if q == "noop" else O.linear_lut(q)[0]


def run_case(img, levels, q, interp, cross_check=True):
    lut = lut_for(q)
    grid, rec, fb = O.encode(img, levels, lut, interp, want_rec=True)
    dec = O.decode(grid, levels, interp)
    assert (dec == rec).all(), "decode(encode(x)) != encoder reconstruction"
    if cross_check:
        g2, r2, fb2 = NP.encode(img, levels, lut, interp, want_rec=True)
        assert (g2 == grid).all() and (r2 == rec).all() and fb2 == fb
        assert (NP.decode(grid, levels, interp) == dec).all()
    _, mse, mx = O.sq_error(img, dec)
    return grid, dec, dict(levels=levels, quant=q, interp=interp, w=int(img.shape[1]),
                           h=int(img.shape[0]), sha_in=sha(img), sha_grid=sha(grid),
                           sha_dec=sha(dec), fallbacks=int(fb), max_abs=mx, int_mse=mse)


def luma709(rgb):
    """image-0.19 `to_luma` on RGB8: truncated f32 0.2126 R + 0.7152 G + 0.0722 B."""
    rgb = rgb.astype(np.float32)
    return np.floor(np.float32(0.2126) * rgb[..., 0] + np.float32(0.7152) * rgb[..., 1] + np.float32(0.0722) * rgb[..., 2]).astype(np.uint8)


def add_luma709(golden):
    from PIL import Image
    plane = luma709(np.array(Image.open(os.path.join(REF, "res/fullhd.jpg")).convert("RGB")))
    assert plane.shape == (1080, 1920)
    Image.fromarray(plane).save(os.path.join(HERE, "fullhd_luma709.png"), optimize=True)
    for q in (0, 1, 2, 3):
        for interp in (O.CROSSED, O.LEFTTOP):
            golden["fullhd_luma709/L4/q%d/i%d" % (q, interp)] = run_case(plane, 4, q, interp)[2]


def main():
    from PIL import Image
    if "--add-luma709" in sys.argv:      # add the BT.709 plane of C1 to the committed golden.json without redoing the big cases
        with open(os.path.join(HERE, "golden.json")) as f:
            golden = json.load(f)
        add_luma709(golden)
        with open(os.path.join(HERE, "golden.json"), "w") as f:
            json.dump(golden, f, indent=1, sort_keys=True)
        print("golden.json now holds %d cases" % len(golden))
        return
    lena = np.array(Image.open(os.path.join(REF, "res/LENA.TIF")))
    assert lena.shape == (256, 256) and lena.dtype == np.uint8
    assert sha(lena).startswith("f6a26c7641342ed5")
    lena.tofile(os.path.join(HERE, "lena_256.u8"))
    fullhd = np.array(Image.open(os.path.join(REF, "res/fullhd.jpg")).convert("L"))
    assert fullhd.shape == (1080, 1920)
    Image.fromarray(fullhd).save(os.path.join(HERE, "fullhd_luma.png"), optimize=True)

    # the reference's own before / after pair (README.md:6-9)
    rgb = np.array(Image.open(os.path.join(REF, "docs/static_files/lena_source.png")).convert("RGB")).astype(np.float32)
    luma = np.floor(np.float32(0.2126) * rgb[..., 0] + np.float32(0.7152) * rgb[..., 1] + np.float32(0.0722) * rgb[..., 2])
    after = np.array(Image.open(os.path.join(REF, "docs/static_files/lena_hgi.png")))
    assert luma.shape == after.shape == (400, 400) and after.dtype == np.uint8
    np.savez_compressed(os.path.join(HERE, "docs_lena_pair.npz"), source_luma=luma.astype(np.uint8), hgi_low=after)

    golden, small = {}, {}
    add_luma709(golden)
    rng = np.random.default_rng(SEED0)
    tiny = {
        "xy_12x8": (xy(12, 8), 3), "xy_8x8": (xy(8, 8), 3), "xy_13x7": (xy(13, 7), 3),
        "xy_30x17": (xy(30, 17), 4), "xy_5x5": (xy(5, 5), 2), "xy_1x1": (xy(1, 1), 3),
        "xy_3x9": (xy(3, 9), 4), "rnd_33x65": (rng.integers(0, 256, (65, 33), dtype=np.uint8), 5),
        "rnd_64x48": (rng.integers(0, 256, (48, 64), dtype=np.uint8), 6),
        "rnd_16x16_l0": (rng.integers(0, 256, (16, 16), dtype=np.uint8), 0),
        "rnd_17x31_l9": (rng.integers(0, 256, (31, 17), dtype=np.uint8), 9),
    }
    for name, (img, levels) in tiny.items():
        small["in/" + name] = img
        for q in (0, 1, 2, 3, "noop"):
            for interp in (O.CROSSED, O.LEFTTOP):
                key = "%s/L%d/q%s/i%d" % (name, levels, q, interp)
                grid, dec, meta = run_case(img, levels, q, interp)
                golden[key] = meta
                small["grid/" + key] = grid
                small["dec/" + key] = dec
    for q in (0, 1, 2, 3):
        for interp in (O.CROSSED, O.LEFTTOP):
            golden["lena_256/L4/q%d/i%d" % (q, interp)] = run_case(lena, 4, q, interp)[2]
            golden["fullhd_luma/L4/q%d/i%d" % (q, interp)] = run_case(fullhd, 4, q, interp)[2]
    # criterion-bench image (benches/bench.rs:15-31) and BASELINE configs, sha only
    for q in (0, 2, "noop"):
        for interp in (O.CROSSED, O.LEFTTOP):
            golden["xy_1920x1080/L4/q%s/i%d" % (q, interp)] = run_case(xy(1920, 1080), 4, q, interp)[2]
    big = {
        "noise2_4096/L6/q0/i1": (O.synth(O.SYNTH_NOISE, SEED0 + 2, 0, 4096, 4096), 6, 0),
        "xy_4096/L6/q0/i1": (xy(4096, 4096), 6, 0),
        "ramp3_f0_4096/L4/q2/i1": (O.synth(O.SYNTH_RAMP, SEED0 + 3, 0, 4096, 4096), 4, 2),
        "ramp3_f511_4096/L4/q2/i1": (O.synth(O.SYNTH_RAMP, SEED0 + 3, 511, 4096, 4096), 4, 2),
        "noise3_f7_4096/L4/q2/i1": (O.synth(O.SYNTH_NOISE, SEED0 + 3, 7, 4096, 4096), 4, 2),
        "ramp4_16384/L8/q3/i1": (O.synth(O.SYNTH_RAMP, SEED0 + 4, 0, 16384, 16384), 8, 3),
    }
    for key, (img, levels, q) in big.items():
        golden[key] = run_case(img, levels, q, O.CROSSED, cross_check=img.size <= 1 << 25)[2]
        print(key, golden[key]["sha_grid"][:16], golden[key]["fallbacks"], flush=True)

    np.savez_compressed(os.path.join(HERE, "small_cases.npz"), **small)
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(golden, f, indent=1, sort_keys=True)
    print("wrote %d cases" % len(golden))


if __name__ == "__main__":
    main()
