"""Minimal ENVI header / raw-binary reader and writer.

The reference goes through the ``spectral`` package (``envi.open(hdr, image=file).open_memmap(interleave='source')``
and ``envi.create_image``, cmf/robust_mf.py:206-208, :261-262, :278-279); this module does the same job with
numpy only: parse ``key = value`` / ``key = { ... }`` headers, memory-map the raw file in its own interleave, and
write headers the way the reference's products carry them (cnn/samples/ang20200924t211102_ch4mf_v2y1_img.hdr).
"""
from __future__ import annotations

import os
import re

import numpy as np

ENVI_DTYPES = {1: np.uint8, 2: np.int16, 3: np.int32, 4: np.float32, 5: np.float64, 12: np.uint16, 13: np.uint32,
               14: np.int64, 15: np.uint64}
DTYPE_TO_ENVI = {np.dtype(v).str[1:]: k for k, v in ENVI_DTYPES.items()}


def read_header(path):
    """Parse an ENVI .hdr file -> dict with lower-case keys; brace lists become python lists of strings."""
    text = open(path, "r", errors="replace").read()
    if not text.lstrip().upper().startswith("ENVI"):
        raise ValueError("%s is not an ENVI header" % path)
    meta = {}
    for m in re.finditer(r"^\s*([^=\n]+?)\s*=\s*(\{.*?\}|[^\n]*)", text, flags=re.S | re.M):
        key, val = m.group(1).strip().lower(), m.group(2).strip()
        if val.startswith("{"):
            inner = val[1:-1].strip()
            if key in ("description", "model parameters", "map info", "coordinate system string"):
                meta[key] = inner if key != "model parameters" else "{ %s }" % inner
            else:
                meta[key] = [v.strip() for v in inner.replace("\n", " ").split(",")] if inner else []
        else:
            meta[key] = val
    for k in ("lines", "samples", "bands", "data type", "byte order", "header offset"):
        if k in meta:
            meta[k] = int(meta[k])
    return meta


def _find_header(image_path):
    for cand in (image_path + ".hdr", os.path.splitext(image_path)[0] + ".hdr"):
        if os.path.isfile(cand):
            return cand
    raise FileNotFoundError("no ENVI header for %s" % image_path)


def open_memmap(image_path, mode="r"):
    """(memmap in SOURCE interleave, metadata).  Shapes: bil (lines, bands, samples) -- what the reference's column
    loop indexes (cmf/robust_mf.py:208, :298); bip (lines, samples, bands); bsq (bands, lines, samples)."""
    meta = read_header(_find_header(image_path))
    dt = np.dtype(ENVI_DTYPES[meta["data type"]]).newbyteorder(">" if meta.get("byte order", 0) == 1 else "<")
    L, S, B = meta["lines"], meta["samples"], meta["bands"]
    il = str(meta.get("interleave", "bsq")).lower()
    shape = {"bil": (L, B, S), "bip": (L, S, B), "bsq": (B, L, S)}[il]
    mm = np.memmap(image_path, dtype=dt, mode=mode, offset=meta.get("header offset", 0), shape=shape)
    return mm, meta


def to_bil(mm, meta):
    il = str(meta.get("interleave", "bsq")).lower()
    if il == "bil":
        return mm
    if il == "bip":
        return mm.transpose(0, 2, 1)
    return mm.transpose(1, 0, 2)


def _fmt(v):
    if isinstance(v, (list, tuple)):
        return "{ %s }" % " , ".join(str(x) for x in v)
    return str(v)


def write_header(path, meta):
    order = ["description", "samples", "lines", "bands", "header offset", "file type", "data type", "interleave",
             "byte order"]
    with open(path, "w") as f:
        f.write("ENVI\n")
        done = set()
        for k in order + [k for k in meta if k not in order]:
            if k in meta and k not in done:
                done.add(k)
                v = meta[k]
                if k == "description" and not str(v).lstrip().startswith("{"):
                    v = "{ %s }" % v
                f.write("%s = %s\n" % (k, _fmt(v)))


def create_image(path, meta, dtype, interleave="bip"):
    """Create the raw file + header; returns a writable memmap in the given interleave (the reference's products are
    BIP, cmf/robust_mf.py:228)."""
    m = dict(meta)
    m["data type"] = DTYPE_TO_ENVI[np.dtype(dtype).str[1:]]
    m["interleave"] = interleave
    m["byte order"] = 0
    m["header offset"] = 0          # the data is written at the start of the file, whatever the input's header said
    m.setdefault("file type", "ENVI Standard")
    L, S, B = int(m["lines"]), int(m["samples"]), int(m["bands"])
    shape = {"bil": (L, B, S), "bip": (L, S, B), "bsq": (B, L, S)}[interleave]
    write_header(path + ".hdr", m)
    return np.memmap(path, dtype=np.dtype(dtype), mode="w+", shape=shape)
