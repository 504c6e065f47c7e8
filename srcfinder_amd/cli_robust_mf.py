#!/usr/bin/env python3
"""Command line of the columnwise matched filter with the flag set of ``cmf/robust_mf.py:142-166``:

    python -m srcfinder_amd.cli_robust_mf [-v] [-k K] [--pcadim N] [-r] [-f] [--rgb_bands R,G,B] [-m] [-R] [-M MODEL]
                                          INPUT LIBRARY OUTPUT

INPUT is an ENVI radiance cube (BIL as delivered; BIP/BSQ are re-ordered), LIBRARY the 3-column target table whose
file name selects the active window (:185-194), OUTPUT the float64 BIP product [R, G, B, CH4 ppm*m] (+ ``_bgmeta``
with ``-m``; + ``<input stem>_column_stats.csv``, which the reference intends to write but crashes on).
"""
import argparse
import os
import sys
import time

import numpy as np


def build_parser():
    parser = argparse.ArgumentParser(description="Robust MF")
    parser.add_argument('-v', '--verbose', action='store_true', help='verbose output')
    parser.add_argument('-k', '--kmeans', type=int, default=1, help='number of columnwise modes (k-means clusters)')
    parser.add_argument('--pcadim', type=int, default=6, help='number of PCA dims (for k-means clusters>1)')
    parser.add_argument('-r', '--reject', action='store_true', help='enable multimodal covariance outlier rejection')
    parser.add_argument('-f', '--full', action='store_true',
                        help='regularize multimodal estimates with the full column covariariance')
    parser.add_argument('--rgb_bands', default='60,42,24', help='comma-separated list of RGB channels')
    parser.add_argument('-m', '--metadata', action='store_true', help='save metadata image')
    parser.add_argument('-R', '--reflectance', action='store_true', help='reflectance signature')
    parser.add_argument('-M', '--model', type=str, default='looshrinkage', help='model name (looshrinkage (default)|empirical)')
    parser.add_argument('input', type=str, metavar='INPUT', help='path to input image')
    parser.add_argument('library', type=str, metavar='LIBRARY', help='path to target library file')
    parser.add_argument('output', type=str, metavar='OUTPUT', help='path for output image (mf ch4 ppm)')
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    from . import cmf, envi, ingest

    try:
        gas = cmf.gas_from_library_name(args.library)
    except ValueError:
        print('could not set active range')          # cmf/robust_mf.py:192-194
        return 0
    active = cmf.active_window(gas, args.reflectance)
    rgb_bands = [] if args.rgb_bands == '[]' else [int(v) for v in args.rgb_bands.split(',')]
    print('started processing input file: "%s"' % str(args.input))
    mm, meta = envi.open_memmap(args.input)
    nodata = float(meta.get('data ignore value', -9999))
    libdata = np.float64(np.loadtxt(args.library))
    stime = time.time()
    # file -> HBM: only the active window and the RGB bands leave the memory map (the reference slices the same bands
    # out of its memmap, :206-208, :298, :395-397), line chunks through pinned buffers with asynchronous copies (ingest.py)
    if len(rgb_bands) not in (0, 3):
        raise Exception('invalid value of rgb_bands argument: %s' % (tuple(rgb_bands),))         # :225-226
    cube = ingest.stage_cube(mm, active, rgb_bands, interleave=str(meta.get('interleave', 'bsq')).lower())
    if args.verbose:
        print('staged %d of %d bands: %.2f GB in %.2f s (%.1f GB/s, host fill %.2f s)'
              % (cube.stats['bands_moved'], cube.stats['bands_total'], cube.stats['bytes'] / 1e9, cube.stats['seconds'],
                 cube.stats['GBps'], cube.stats['host_fill_seconds']))
    res = cmf.robust_mf(cube, libdata, gas=gas, reflectance=args.reflectance, kmeans=args.kmeans, pcadim=args.pcadim,
                        reject=args.reject, full=args.full, model=args.model, rgb_bands=rgb_bands, nodata=nodata,
                        metadata=args.metadata)
    nrows, _, ncols = cube.shape
    outmeta = {k: v for k, v in meta.items()
               if k not in ('smoothing factors', 'wavelength', 'wavelength units', 'fwhm')}   # :229-230
    outmeta['lines'] = nrows
    if len(rgb_bands) == 3:
        outmeta['bands'] = 4
        outmeta['band names'] = ['Red Radiance (uW/nm/sr/cm2)', 'Green Radiance (uW/nm/sr/cm2)',
                                 'Blue Radiance (uW/nm/sr/cm2)', 'CH4 Absorption (ppm x m)']   # :218-221
    else:
        outmeta['bands'] = 1
        outmeta['band names'] = ['CH4 Absorption (ppm x m)']
    outmeta['model parameters'] = res.modelparms
    out_mm = envi.create_image(args.output, outmeta, np.float64, 'bip')
    ingest.fetch_product(res.out, out_mm)               # device -> file through pinned chunks
    out_mm.flush()
    if args.metadata:
        alphas = cmf.alpha_grid()
        bgmeta = dict(outmeta)
        bgmeta['bands'] = 2
        bgmeta['num alphas'] = len(alphas)
        bgmeta['alphas'] = '{%s}' % (str(alphas)[1:-1])
        bgmeta['band names'] = '{cluster_id, alpha_index}'                                      # :273-277
        bg_mm = envi.create_image(args.output + '_bgmeta', bgmeta, np.int16, 'bip')
        ingest.fetch_product(res.bgmeta, bg_mm)
        bg_mm.flush()
    colnum, colavg, colstd = res.colstats.cpu().numpy()
    if args.verbose:
        for col in range(ncols):
            print('Column %i mean: %e, std: %e' % (col, colavg[col], colstd[col]))
    colcsv = os.path.splitext(args.input)[0] + '_column_stats.csv'
    print('Saving column stats to', colcsv)
    with open(colcsv, 'w') as f:                         # rows npix/avg/std as the reference intends (:399-403)
        f.write(',' + ','.join(str(c) for c in range(ncols)) + '\n')
        for name, row in (('npix', colnum), ('avg', colavg), ('std', colstd)):
            f.write(name + ',' + ','.join(repr(float(v)) for v in row) + '\n')
    print('done (elapsed time=%ds)' % (time.time() - stime))
    return 0


if __name__ == '__main__':
    sys.exit(main())
