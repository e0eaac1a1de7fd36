"""Procedural time-lapse frames with known sub-pixel motion (host side, numpy).

The reference ships no imagery (its loops read private JPEG folders: s1_lucaskanade_tracking.py:259,
s0_1_test_lucaskanade_tracking.py:186), so benchmarks and tests run on a seeded synthetic sequence.
The texture is integer-only value noise (hash lattice + fixed-point smoothstep), which makes the numpy
generator here and the device generator (csrc/icelk_synth.hip, `icelk_synth_frame`) bit-identical:
a frame never has to cross PCIe to be used on the GPU, and a 24-hour sequence (BASELINE.json
configs[3]) is never stored.

Frame t is the texture sampled at (x + ux_t/256, y + uy_t/256); a scene point therefore moves by
-(u_b - u_a)/256 px between frames a and b.
"""
import numpy as np

BIAS_PX = 1 << 16          # keeps fixed-point sample coordinates positive for any shift
FRAC_BITS = 8              # sub-pixel resolution of the motion: 1/256 px
_OCTAVES = ((3, 3), (5, 3), (2, 2))   # (log2 cell size in px, weight); weights sum to 8


def _hash2(ix, iy, seed):
    """uint32 lattice hash; same constants and order as the device generator."""
    with np.errstate(over="ignore"):
        h = (ix * np.uint32(0x9E3779B1)) ^ (iy * np.uint32(0x85EBCA77)) ^ np.uint32((seed * 0xC2B2AE3D) & 0xFFFFFFFF)
        h ^= h >> np.uint32(15)
        h *= np.uint32(0x2C1B3C6D)
        h ^= h >> np.uint32(12)
        h *= np.uint32(0x297A2D39)
        h ^= h >> np.uint32(15)
    return h


def _octave(X, Y, k, seed):
    sh = np.uint32(FRAC_BITS + k)
    cx, cy = X >> sh, Y >> sh
    fx = (X >> np.uint32(k)) & np.uint32(255)
    fy = (Y >> np.uint32(k)) & np.uint32(255)
    sx = (fx * fx * (np.uint32(768) - np.uint32(2) * fx)) >> np.uint32(16)
    sy = (fy * fy * (np.uint32(768) - np.uint32(2) * fy)) >> np.uint32(16)
    one = np.uint32(1)
    s = seed + 7919 * k
    v00 = _hash2(cx, cy, s) >> np.uint32(24)
    v10 = _hash2(cx + one, cy, s) >> np.uint32(24)
    v01 = _hash2(cx, cy + one, s) >> np.uint32(24)
    v11 = _hash2(cx + one, cy + one, s) >> np.uint32(24)
    c256 = np.uint32(256)
    top = v00 * (c256 - sx) + v10 * sx
    bot = v01 * (c256 - sx) + v11 * sx
    return (top * (c256 - sy) + bot * sy) >> np.uint32(16)


AFFINE_BITS = 20           # affine coefficients are in 2^-20 px per px


def frame(width, height, ux=0, uy=0, seed=1234, rows=None, affine=None):
    """One HxW uint8 frame shifted by (ux, uy)/256 px.  `rows=(y0, y1)` renders a band only.  `affine` = (ax, bx, ay, by)
    adds (ax x + bx y, ay x + by y) / 2^20 px to the sample position: a deformation that varies over the frame."""
    y0, y1 = (0, height) if rows is None else rows
    xi = np.arange(width, dtype=np.int64)[None, :]
    yi = np.arange(y0, y1, dtype=np.int64)[:, None]
    ax, bx, ay, by = (0, 0, 0, 0) if affine is None else (int(v) for v in affine)
    sh = AFFINE_BITS - FRAC_BITS
    xs = ((xi + BIAS_PX) << FRAC_BITS) + int(ux) + ((ax * xi + bx * yi) >> sh)
    ys = ((yi + BIAS_PX) << FRAC_BITS) + int(uy) + ((ay * xi + by * yi) >> sh)
    if xs.min() < 0 or ys.min() < 0 or xs.max() >= 1 << 32 or ys.max() >= 1 << 32:
        raise ValueError("shift out of range")
    X = np.broadcast_to(xs, (y1 - y0, width)).astype(np.uint32)
    Y = np.broadcast_to(ys, (y1 - y0, width)).astype(np.uint32)
    acc = np.zeros((y1 - y0, width), np.uint32)
    for k, wgt in _OCTAVES:
        acc += np.uint32(wgt) * _octave(X, Y, k, seed)
    return (acc >> np.uint32(3)).astype(np.uint8)


def shifts(n_frames, seed=1234, max_step_px=3.0):
    """Cumulative integer shifts (1/256 px) of frames 0..n_frames-1; frame 0 is unshifted."""
    rng = np.random.RandomState(seed)
    m = int(round(max_step_px * (1 << FRAC_BITS)))
    steps = rng.randint(-m, m + 1, size=(max(n_frames - 1, 0), 2))
    out = np.zeros((n_frames, 2), np.int64)
    if n_frames > 1:
        out[1:] = np.cumsum(steps, axis=0)
    return out


def affines(n_frames, seed=1234, max_coef=0.005, step=0.0015):
    """Per-frame affine coefficients (ax, bx, ay, by) in 2^-20 px/px: a seeded random walk of steps <= `step`, kept
    within +-`max_coef` (0.5 %: SURVEY.md 8d "smooth shear <= 0.5 %"); frame 0 is undeformed."""
    rng = np.random.RandomState(seed + 17)
    one = 1 << AFFINE_BITS
    m, st = int(round(max_coef * one)), int(round(step * one))
    out = np.zeros((n_frames, 4), np.int64)
    for i in range(1, n_frames):
        out[i] = np.clip(out[i - 1] + rng.randint(-st, st + 1, size=4), -m, m)
    return out


def true_flow(shift_a, shift_b):
    """Displacement (dx, dy) in px of every scene point from frame a to frame b."""
    d = (np.asarray(shift_b, np.float64) - np.asarray(shift_a, np.float64)) / (1 << FRAC_BITS)
    return -d


def sequence(width, height, n_frames, seed=1234, max_step_px=3.0):
    sh = shifts(n_frames, seed, max_step_px)
    return [frame(width, height, int(sx), int(sy), seed) for sx, sy in sh], sh


def rgb_from_gray_seeded(width, height, ux=0, uy=0, seed=1234):
    """HxWx3 uint8 colour frame (three decorrelated textures) for the cvtColor stage."""
    return np.stack([frame(width, height, ux, uy, seed + 101 * c) for c in range(3)], axis=-1)
