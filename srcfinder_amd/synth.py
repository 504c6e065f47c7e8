"""Synthetic AVIRIS-NG-like radiance cubes for parity tests and the benchmark.

The cube is band-interleaved-by-line (BIL) exactly as the reference opens it
(`cmf/robust_mf.py:206-208`: ``img_mm.shape == (lines, bands, samples)``).

Two generators share one recipe:

* :func:`make_cube_numpy` -- host, ``numpy.random.default_rng(seed)``; this is the
  generator the committed golden vectors were made with, so it must never change.
* :func:`make_cube_torch` -- on-device for the full 598 x 20000 x 425 flightline
  (20 GB; values differ from numpy, parity at that size is checked on sampled columns
  and through size-independent properties).

Recipe: ``base_b = 5 exp(-3 b / (B-1)) + 0.2``; five latent factors with per-band
loadings ``N(0,1) * 0.1 * base``; white noise ``0.01 * base``; then injected
pathologies: leading NODATA lines, one all-NODATA column, a NaN pixel and a negative
pixel inside the active window, and sparse "plume" pixels ``x += a * (abscf * base)``.
"""
from __future__ import annotations

import numpy as np

NODATA = -9999.0
NBANDS = 425
CH4_ACTIVE = (351, 422)  # 1-based inclusive, cmf/robust_mf.py:188-189


def base_spectrum(nbands: int = NBANDS) -> np.ndarray:
    b = np.arange(nbands, dtype=np.float64)
    return 5.0 * np.exp(-3.0 * b / max(nbands - 1, 1)) + 0.2


def synth_columns(n: int, p: int, seed: int, scale: float = 1.0) -> np.ndarray:
    """[n, p] float64 matrix of float32-representable spectra (one detector column) -- the input
    generator of the function-level ``looshrinkage`` golden cases (only the seed is stored)."""
    rng = np.random.default_rng(seed)
    b = 5.0 * np.exp(-3.0 * np.arange(p) / max(p - 1, 1)) + 0.2
    lm = rng.standard_normal((5, p)) * 0.1 * b
    x = b + rng.standard_normal((n, 5)) @ lm + rng.standard_normal((n, p)) * 0.01 * b
    return np.float64(np.float32(x * scale))


def make_cube_numpy(lines: int, samples: int, nbands: int = NBANDS, *, seed: int = 1234,
                    abscf_full: np.ndarray | None = None, active=CH4_ACTIVE,
                    nodata_lines: int = 7, nodata_column: int | None = None,
                    plume_frac: float = 1e-3, inject: bool = True) -> np.ndarray:
    """Return a float32 BIL cube ``[lines, nbands, samples]``.

    ``abscf_full`` is the library's third column for all ``nbands`` channels (used only to
    shape the injected plume signal); ``None`` skips plumes.
    """
    rng = np.random.default_rng(seed)
    base = base_spectrum(nbands)
    lmat = rng.standard_normal((5, nbands)) * 0.1 * base
    z = rng.standard_normal((lines, samples, 5))
    cube = base[None, None, :] + z @ lmat                      # [lines, samples, bands]
    cube += rng.standard_normal((lines, samples, nbands)) * 0.01 * base
    a0, a1 = active
    if inject and abscf_full is not None and plume_frac > 0:
        nplume = max(1, int(round(plume_frac * lines * samples)))
        pl = rng.integers(0, lines, nplume)
        ps = rng.integers(0, samples, nplume)
        amp = rng.uniform(0.005, 0.05, nplume)
        sig = (abscf_full * base)[a0 - 1:a1]
        cube[pl, ps, a0 - 1:a1] += amp[:, None] * sig[None, :]
    cube = np.ascontiguousarray(cube.transpose(0, 2, 1)).astype(np.float32)  # -> BIL
    if inject:
        if nodata_lines > 0:
            cube[:min(nodata_lines, lines)] = NODATA
        if nodata_column is None:
            nodata_column = samples // 3
        if 0 <= nodata_column < samples:
            cube[:, :, nodata_column] = NODATA
        if lines > nodata_lines + 12 and samples > 6:
            cube[nodata_lines + 5, a0 - 1 + 3, 5] = np.nan           # NaN inside the window
            cube[nodata_lines + 9, a0 - 1 + 10, 2] = -0.25            # negative inside the window
            cube[nodata_lines + 11, 10, 4] = -1.0                     # negative OUTSIDE the window: row stays valid
    return cube


def make_cube_torch(lines: int, samples: int, nbands: int = NBANDS, *, seed: int = 1234,
                    device="cuda", abscf_full=None, active=CH4_ACTIVE, nodata_lines: int = 7,
                    nodata_column: int | None = None, plume_frac: float = 1e-3,
                    inject: bool = True, line_block: int = 1000):
    """Device-side generator with the same recipe; returns a float32 BIL torch tensor.

    Generated in blocks of ``line_block`` lines so the fp32 temporaries stay small next
    to the 20 GB cube.
    """
    import torch

    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    base = torch.as_tensor(base_spectrum(nbands), dtype=torch.float32, device=dev)
    lmat = torch.randn((5, nbands), generator=g, device=dev) * 0.1 * base
    cube = torch.empty((lines, nbands, samples), dtype=torch.float32, device=dev)
    a0, a1 = active
    sig = None
    if inject and abscf_full is not None and plume_frac > 0:
        sig = torch.as_tensor(np.asarray(abscf_full, dtype=np.float64), dtype=torch.float32, device=dev) * base
    for l0 in range(0, lines, line_block):
        l1 = min(lines, l0 + line_block)
        z = torch.randn((l1 - l0, samples, 5), generator=g, device=dev)
        blk = base[None, None, :] + z @ lmat
        blk += torch.randn((l1 - l0, samples, nbands), generator=g, device=dev) * (0.01 * base)
        if sig is not None:
            m = torch.rand((l1 - l0, samples), generator=g, device=dev) < plume_frac
            amp = 0.005 + 0.045 * torch.rand((l1 - l0, samples), generator=g, device=dev)
            blk[:, :, a0 - 1:a1] += (m * amp)[:, :, None] * sig[None, None, a0 - 1:a1]
        cube[l0:l1] = blk.permute(0, 2, 1)
        del z, blk
    if inject:
        if nodata_lines > 0:
            cube[:min(nodata_lines, lines)] = NODATA
        if nodata_column is None:
            nodata_column = samples // 3
        if 0 <= nodata_column < samples:
            cube[:, :, nodata_column] = NODATA
        if lines > nodata_lines + 12 and samples > 6:
            cube[nodata_lines + 5, a0 - 1 + 3, 5] = float("nan")
            cube[nodata_lines + 9, a0 - 1 + 10, 2] = -0.25
            cube[nodata_lines + 11, 10, 4] = -1.0
    return cube
