#!/usr/bin/env python3
"""Command line of the CNN saliency scorer with the flag set of ``cnn/cnn_pred_pipeline.py:63-83``:

    python -m srcfinder_amd.cli_cnn_pred FLIGHTLINE [-m MODEL] [-g GPU ...] [-b BATCH] [-o OUTDIR] [--band N] [--weights PT]

FLIGHTLINE is an ENVI raster; ``--band`` (1-based, default 1 like the reference's ``read(1)``) selects the plane that
is scored -- pass 4 for the CMF band of a 4-band matched-filter product.  Weights default to
``<package>/models/<MODEL>.pt``; a missing file exits with status 1 like the reference (:90-95).
"""
import argparse
import os
import os.path as op
import sys
from pathlib import Path

import numpy as np


def build_parser():
    parser = argparse.ArgumentParser(description="Generate a flightline saliency map with a CNN.")
    parser.add_argument('flightline', help="Filepaths to flightline ENVI IMG.", type=str)
    parser.add_argument('--model', '-m', help="Model to use for prediction.", default="COVID_QC",
                        choices=["COVID_QC", "CalCH4_v8", "Permian_QC", "multi_256", "multi_64"])
    parser.add_argument('--gpus', '-g', help="GPU devices for inference. -1 for CPU.", nargs='+', default=[-1], type=int)
    parser.add_argument('--batch', '-b', help="Batch size per device.", default=32, type=int)
    parser.add_argument('--output', '-o', help="Output directory for generated saliency maps.", default=".", type=str)
    parser.add_argument('--band', help="1-based band of the raster to score (reference: 1).", default=1, type=int)
    parser.add_argument('--weights', help="GoogLeNet state_dict (.pt); default <package>/models/<model>.pt", default=None)
    return parser


def main(argv=None):
    args = build_parser().parse_args(argv)
    print("[STEP] MODEL INITIALIZATION")
    weightpath = args.weights or op.join(Path(__file__).parent.resolve(), 'models', f"{args.model}.pt")
    if op.isfile(weightpath):
        print(f"[INFO] Found {weightpath}.")
    else:
        print(f"[INFO] Model not found at {weightpath}, exiting.")
        return 1
    import torch
    if not torch.cuda.is_available():
        print("[ERR] CUDA not found, exiting.")            # there is no CPU path here (reference: -g -1 runs on CPU)
        return 1
    from . import cnn, envi
    if any(g < 0 for g in args.gpus):
        print("[ERR] -g -1 (CPU) is not available: srcfinder_amd has no CPU path, exiting.")
        return 1
    if max(args.gpus) >= torch.cuda.device_count():
        print("[ERR] -g %s: only %d GPU(s) visible, exiting." % (" ".join(map(str, args.gpus)), torch.cuda.device_count()))
        return 1
    sd = torch.load(weightpath, map_location="cpu")
    mm, meta = envi.open_memmap(args.flightline)
    bil = envi.to_bil(mm, meta)
    plane = np.ascontiguousarray(bil[:, args.band - 1, :], dtype=np.float32)
    print("[STEP] MODEL PREDICTION")
    # -g 0 1 2 3: every listed GPU scores its own block of rows with the per-GPU batch size (the reference multiplies
    # the batch by the GPU count for DataParallel to split again, cnn_pred_pipeline.py:170)
    # --batch keeps the reference's flag and default (32 windows per device: a DataLoader batch, cnn_pred_pipeline.py:76, :165).  The
    # saliency map does not depend on the batch size (bit-identical: tests/test_cnn_gpu.py), only the speed does -- 19.6 k windows/s
    # at 32, 103-108 k at 1024 on an MI355X (the per-window rings of the shared trunk are small GEMMs) -- so at least 1024 windows are
    # scored per launch set (two concurrent row halves with 19 GB of workspace each per device)
    exec_batch = max(int(args.batch), 1024)
    sal = cnn.predict_flightline(plane, args.model, weights=sd, batch=exec_batch, gpus=list(args.gpus), to_numpy=True)
    print("[STEP] RESULT EXPORT")
    outpath = op.join(args.output, f"{Path(args.flightline).stem}_saliency.img")
    print("[INFO] Saving to", outpath)
    outmeta = {k: v for k, v in meta.items() if k in ('map info', 'coordinate system string', 'data ignore value')}
    outmeta.update(lines=plane.shape[0], samples=plane.shape[1], bands=1)
    out = envi.create_image(outpath, outmeta, np.float32, 'bsq')
    out[0] = sal
    out.flush()
    print("Done!")
    return 0


if __name__ == "__main__":
    sys.exit(main())
