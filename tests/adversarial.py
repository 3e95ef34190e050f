"""Adversarial image content for the parity tests (every other input of the suite comes from one
generator, popsift_amd/synth.py -- band-limited noise + Gaussian blobs).  These hit what that generator never
does: saturated 0 / 255 plateaus, step edges, periodic texture, exact DoG ties (the strict 26-neighbour tests of
s_extrema.cu:56-120 must reject every one of them) and long straight edges (the edge test, s_extrema.cu:491).
Deterministic, numpy only; uint8 (h, w) arrays."""
import numpy as np


def _lcg(seed):
    s = np.uint32(seed)
    while True:
        s = np.uint32((int(s) * 1664525 + 1013904223) & 0xFFFFFFFF)
        yield int(s)


def plateaus(w, h, seed=1):
    """Large blobs amplified far beyond the u8 range: wide saturated plateaus at 0 and 255 with smooth rims."""
    g = _lcg(seed)
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.full((h, w), 128.0)
    for _ in range(max(6, w * h // 12000)):
        cx, cy = next(g) % w, next(g) % h
        s = 6.0 + (next(g) % 1000) / 1000.0 * 30.0
        a = (600.0 if next(g) & 1 else -600.0)
        img += a * np.exp(-((x - cx) ** 2 + (y - cy) ** 2) / (2 * s * s))
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)


def checker(w, h, cell=8):
    """Binary checkerboard: step edges in both directions, corners everywhere, DoG ties along the diagonals."""
    y, x = np.mgrid[0:h, 0:w]
    return ((((x // cell) + (y // cell)) & 1) * 255).astype(np.uint8)


def textlike(w, h, seed=2):
    """Binary strokes (thin axis-aligned rectangles of random length) on white: text / line-drawing content."""
    g = _lcg(seed)
    img = np.full((h, w), 255, np.uint8)
    for _ in range(max(40, w * h // 900)):
        x0, y0 = next(g) % w, next(g) % h
        ln, th = 3 + next(g) % 28, 1 + next(g) % 3
        if next(g) & 1:
            img[y0:y0 + th, x0:x0 + ln] = 0
        else:
            img[y0:y0 + ln, x0:x0 + th] = 0
    return img


def grating(w, h, period=7.3, angle_deg=30.0):
    """Sine grating: periodic texture, every crest a line of near-ties."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    a = np.deg2rad(angle_deg)
    return np.rint(127.5 + 127.5 * np.sin(2 * np.pi * (x * np.cos(a) + y * np.sin(a)) / period)).astype(np.uint8)


def ramp(w, h):
    """Pure horizontal ramp: no extremum anywhere; constant rows make every vertical neighbour an exact tie."""
    return np.broadcast_to((np.arange(w, dtype=np.int64) * 255 // max(w - 1, 1)).astype(np.uint8), (h, w)).copy()


def stripes(w, h, width=2):
    """Vertical bars (left half) and horizontal bars (right half): the DoG is constant along a bar, so every pixel
    ties exactly with its neighbours along the bar -- a non-strict extremum test would accept whole lines."""
    y, x = np.mgrid[0:h, 0:w]
    v = ((x // width) & 1) * 255
    hz = ((y // width) & 1) * 255
    return np.where(x < w // 2, v, hz).astype(np.uint8)


def composite(w, h, seed=3):
    """All of the above in a 3 x 2 mosaic (hard boundaries between the tiles included)."""
    tw, th = (w + 2) // 3, (h + 1) // 2
    tiles = [plateaus(tw, th, seed), checker(tw, th), textlike(tw, th, seed + 1), grating(tw, th), ramp(tw, th), stripes(tw, th)]
    rows = [np.concatenate(tiles[0:3], axis=1), np.concatenate(tiles[3:6], axis=1)]
    return np.ascontiguousarray(np.concatenate(rows, axis=0)[:h, :w])


CONTENT = {
    "plateaus": plateaus, "checker": checker, "textlike": textlike, "grating": grating, "ramp": ramp,
    "stripes": stripes, "composite": composite,
}


def make(name, w, h):
    return np.ascontiguousarray(CONTENT[name](w, h))
