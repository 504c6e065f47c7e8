"""Seeded synthetic GoogLeNet weights (the trained ``cnn/models/*.pt`` are not in the reference checkout,
SURVEY.md D9) and the layer table of the eval graph of ``cnn/archs/googlenet1.py``.

The generator is a counter-based integer hash (splitmix64 of tensor index and element index), evaluated with
numpy uint64 arithmetic: bit-reproducible on any machine, no dependence on a library RNG stream.  BatchNorm
running statistics are non-trivial on purpose (default mean 0 / var 1 would hide folding bugs).
"""
from __future__ import annotations

import numpy as np

# (name, cin, cout, k, stride, pad) of every BasicConv2d of the eval graph, in state_dict order
INCEPTION = [  # name, cin, ch1x1, ch3x3red, ch3x3, ch5x5red, ch5x5, pool_proj   (googlenet1.py:66-78)
    ("inception3a", 192, 64, 96, 128, 16, 32, 32),
    ("inception3b", 256, 128, 128, 192, 32, 96, 64),
    ("inception4a", 480, 192, 96, 208, 16, 48, 64),
    ("inception4b", 512, 160, 112, 224, 24, 64, 64),
    ("inception4c", 512, 128, 128, 256, 24, 64, 64),
    ("inception4d", 512, 112, 144, 288, 32, 64, 64),
    ("inception4e", 528, 256, 160, 320, 32, 128, 128),
    ("inception5a", 832, 256, 160, 320, 32, 128, 128),
    ("inception5b", 832, 384, 192, 384, 48, 128, 128),
]


def conv_table():
    t = [("conv1", 1, 64, 7, 2, 3), ("conv2", 64, 64, 1, 1, 0), ("conv3", 64, 192, 3, 1, 1)]
    for name, cin, c1, c3r, c3, c5r, c5, pp in INCEPTION:
        t += [(name + ".branch1", cin, c1, 1, 1, 0),
              (name + ".branch2.0", cin, c3r, 1, 1, 0), (name + ".branch2.1", c3r, c3, 3, 1, 1),
              (name + ".branch3.0", cin, c5r, 1, 1, 0), (name + ".branch3.1", c5r, c5, 3, 1, 1),   # 3x3, googlenet1.py:207-209
              (name + ".branch4.1", cin, pp, 1, 1, 0)]
    return t


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def _uniform(tensor_id: int, n: int, seed: int) -> np.ndarray:
    """n doubles in [0, 1), a pure function of (seed, tensor_id, element index)."""
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.uint64(seed) * np.uint64(0x100000001B3) + np.uint64(tensor_id))
        bits = _splitmix64(idx ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def synthetic_state_dict(seed: int = 2024, num_classes: int = 2):
    """name -> float32 ndarray, the eval-relevant tensors of ``googlenet(num_classes=2)``'s state_dict
    (aux heads and ``num_batches_tracked`` omitted -- the loader must accept their absence or presence)."""
    sd = {}
    tid = 0

    def u(shape, lo, hi):
        nonlocal tid
        tid += 1
        n = int(np.prod(shape))
        return (lo + (hi - lo) * _uniform(tid, n, seed)).reshape(shape).astype(np.float32)

    for name, cin, cout, k, s, p in conv_table():
        a = np.sqrt(6.0 / (cin * k * k))          # He-uniform keeps activations O(1) through 22 layers
        sd[name + ".conv.weight"] = u((cout, cin, k, k), -a, a)
        sd[name + ".bn.weight"] = u((cout,), 0.8, 1.2)
        sd[name + ".bn.bias"] = u((cout,), -0.1, 0.1)
        sd[name + ".bn.running_mean"] = u((cout,), -0.1, 0.1)
        sd[name + ".bn.running_var"] = u((cout,), 0.5, 1.5)
    a = np.sqrt(3.0 / 1024)
    sd["fc.weight"] = u((num_classes, 1024), -a, a)
    sd["fc.bias"] = u((num_classes,), -0.1, 0.1)
    return sd


def synthetic_plane(h: int, w: int, seed: int = 7) -> np.ndarray:
    """A CMF-like float32 plane: background noise, a few bright blobs above the clamp, NODATA pixels."""
    v = _uniform(10_001, h * w, seed).reshape(h, w)
    plane = (v * 600.0 - 100.0)
    yy, xx = np.mgrid[0:h, 0:w]
    for i, (cy, cx, amp, sig) in enumerate([(0.3, 0.4, 5000.0, 2.5), (0.7, 0.2, 1500.0, 4.0), (0.55, 0.8, 900.0, 1.5)]):
        plane += amp * np.exp(-(((yy - cy * h) ** 2 + (xx - cx * w) ** 2) / (2 * sig ** 2)))
    plane = plane.astype(np.float32)
    plane[0, :3] = -9999.0
    plane[h // 2, w // 3] = -9999.0
    return plane
