#!/usr/bin/env python3
"""Command line of the FCN shift-and-stitch saliency map with the flag set of ``cnn/fcn_pred_pipeline.py:95-121``:

    python -m srcfinder_amd.cli_fcn_pred FLIGHTLINE [-n BAND] [-s SCALE] [-m MODEL] [-g GPU ...] [-b BATCH] [-o OUTDIR] [--weights PT]

The reference's approximate fast mode: the trunk runs over the whole flightline once per (top, left) shift and the
32 x 32 maps are interlaced (``srcfinder_amd.cnn.fcn_predict_flightline``).  Weights default to
``<package>/models/<MODEL>.pt``; a missing file exits with status 1 like the reference (:133-141).
"""
import argparse
import os.path as op
import sys
from pathlib import Path

import numpy as np


def build_parser():
    parser = argparse.ArgumentParser(description="Generate a flightline saliency map with a FCN.")
    parser.add_argument('flightline', help="Filepaths to flightline ENVI IMG.", type=str)
    parser.add_argument('--band', '-n', help="Band to read if multiband", default=1, type=int)
    parser.add_argument('--scale', '-s', help="Downscaling factor of the model", default=32, type=int)
    parser.add_argument('--model', '-m', help="Model to use for prediction.", default="COVID_QC",
                        choices=["COVID_QC", "CalCH4_v8", "Permian_QC", "multi_256", "multi_64"])
    parser.add_argument('--gpus', '-g', help="GPU devices for inference. -1 for CPU.", nargs='+', default=[-1], type=int)
    parser.add_argument('--batch', '-b', help="Batch size per device.", default=8, type=int)
    parser.add_argument('--output', '-o', help="Output directory for generated saliency maps.", default=".", type=str)
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
    device = torch.device("cuda:%d" % (args.gpus[0] if args.gpus[0] >= 0 else 0))
    print("[INFO] Converting CNN to FCN.")
    net = cnn.GoogLeNetHIP(torch.load(weightpath, map_location="cpu"), device=device)
    mm, meta = envi.open_memmap(args.flightline)
    bil = envi.to_bil(mm, meta)
    plane = np.ascontiguousarray(bil[:, args.band - 1, :], dtype=np.float32)
    print("[STEP] MODEL PREDICTION")
    sal = cnn.fcn_predict_flightline(plane, args.model, net=net, scale=args.scale, batch=args.batch * max(1, len(args.gpus)),
                                     to_numpy=True)
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
