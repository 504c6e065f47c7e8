"""csrc/cnn_ring.h -- the frame geometry of the tile scorer's trunk sharing -- compiled for the HOST (g++) and checked without a
GPU: the ring enumeration is a bijection onto exactly the positions outside the interior, the frames of
cnn/archs/googlenet1.py:60-68, :110-123 follow from the layers' receptive fields, and interior + ring cover every grid."""
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#define __host__
#define __device__
#include "cnn_ring.h"
#include <cstdio>
int main() {
  // (G, lo, hi) -> count, then for every ring index its (y, x); then for every grid position: ring flag and index (or -1)
  const int frames[][3] = {{64, 1, 1}, {64, 2, 2}, {32, 1, 2}, {32, 2, 3}, {32, 3, 4}, {16, 2, 3}, {8, 0, 0}, {5, 2, 2}};
  for (auto &f : frames) {
    const int G = f[0], lo = f[1], hi = f[2], n = sf_frame_count(G, lo, hi);
    std::printf("F %d %d %d %d\n", G, lo, hi, n);
    for (int j = 0; j < n; ++j) { int y, x; sf_frame_position(G, lo, hi, j, y, x); std::printf("P %d %d %d\n", j, y, x); }
    for (int y = 0; y < G; ++y)
      for (int x = 0; x < G; ++x) std::printf("Q %d %d %d\n", y, x, sf_frame_ring(G, lo, hi, y, x) ? sf_frame_index(G, lo, hi, y, x) : -1);
    // band sharing: the side and the band-interior enumerations
    for (int j = 0; j < sf_side_count(G, lo, hi); ++j) { int y, x; sf_side_position(G, lo, hi, j, y, x); std::printf("S %d %d %d\n", j, y, x); }
    for (int j = 0; j < sf_band_count(G, lo, hi); ++j) { int y, x; sf_band_position(G, lo, hi, j, y, x); std::printf("B %d %d %d\n", j, y, x); }
    std::printf("E\n");
  }
  return 0;
}
"""


def _run():
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, "ring.cpp"), os.path.join(d, "ring")
        open(src, "w").write(SRC)
        subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "srcfinder_amd", "csrc"), src, "-o", exe], check=True)
        return subprocess.run([exe], check=True, capture_output=True, text=True).stdout.splitlines()


def test_ring_enumeration_is_a_bijection_onto_the_frame():
    lines = _run()
    i = 0
    while i < len(lines):
        tag, G, lo, hi, n = lines[i].split()
        assert tag == "F"
        G, lo, hi, n = int(G), int(lo), int(hi), int(n)
        i += 1
        pos = {}
        for _ in range(n):
            _p, j, y, x = lines[i].split()
            pos[int(j)] = (int(y), int(x))
            i += 1
        grid = np.full((G, G), -2)
        for _ in range(G * G):
            _q, y, x, idx = lines[i].split()
            grid[int(y), int(x)] = int(idx)
            i += 1
        ring = np.ones((G, G), bool)
        ring[lo:G - hi, lo:G - hi] = False                          # the interior [lo, G - 1 - hi]^2
        assert n == ring.sum() == G * G - (G - lo - hi) ** 2
        assert np.array_equal(grid >= 0, ring)
        assert sorted(grid[ring].tolist()) == list(range(n))        # every index exactly once
        for j, (y, x) in pos.items():
            assert grid[y, x] == j                                  # position(index(y, x)) == (y, x)
        # band sharing (round 6): side columns + band interior partition the ring; the side is every row's columns x < lo, x >= G - hi
        side, band = [], []
        while lines[i] != "E":
            tag, j, y, x = lines[i].split()
            (side if tag == "S" else band).append((int(y), int(x)))
            i += 1
        i += 1
        assert len(side) == G * (lo + hi) and len(band) == (lo + hi) * (G - lo - hi) and len(side) + len(band) == n
        assert len(set(side)) == len(side) and len(set(band)) == len(band) and not set(side) & set(band)
        assert all(x < lo or x >= G - hi for _y, x in side)
        assert all(lo <= x < G - hi and (y < lo or y >= G - hi) for y, x in band)
        assert all(ring[y, x] for y, x in side + band) if n else True


def test_frames_follow_from_the_receptive_fields():
    """The frames the driver uses (csrc/cnn_driver.hip), derived here from scratch: a position is INTERIOR when the layer's value does
    not depend on the window's zero padding -- i.e. when its receptive field, traced back through the stack, lies inside the 256 x 256
    window.  1-D (the layers are separable in this respect)."""
    def conv(affected, k, s, p, n_out):          # affected[i]: input position i sees the padding (or lies outside the tensor)
        n_in = len(affected)
        out = []
        for o in range(n_out):
            taps = [o * s - p + t for t in range(k)]
            out.append(any(t < 0 or t >= n_in or affected[t] for t in taps))
        return out

    def pool_ceil(affected, k, s, n_out):        # ceil mode: taps beyond the end do not exist (they are not padding)
        n_in = len(affected)
        return [any(affected[t] for t in range(o * s, min(o * s + k, n_in))) for o in range(n_out)]

    def frame(a):
        lo = next(i for i, v in enumerate(a) if not v)
        hi = next(i for i, v in enumerate(reversed(a)) if not v)
        assert not any(a[lo:len(a) - hi])        # the interior is one block
        return lo, hi

    x = [False] * 256
    c1 = conv(x, 7, 2, 3, 128)                   # conv1 7x7 s2 p3 (googlenet1.py:60)
    p1 = pool_ceil(c1, 3, 2, 64)                 # maxpool1 (:61)
    c2 = conv(p1, 1, 1, 0, 64)                   # conv2 1x1 (:62)
    c3 = conv(c2, 3, 1, 1, 64)                   # conv3 3x3 p1 (:63)
    p2 = pool_ceil(c3, 3, 2, 32)                 # maxpool2 (:64)
    a3 = conv(p2, 3, 1, 1, 32)                   # inception3a: its 3 x 3 branches / pool-projection (:184-228)
    b3 = conv(a3, 3, 1, 1, 32)                   # inception3b
    p3 = pool_ceil(b3, 3, 2, 16)                 # maxpool3 (:68)
    assert [i for i, v in enumerate(c1) if v] == [0, 1, 127]
    assert frame(p1) == (1, 1) and frame(c2) == (1, 1) and frame(c3) == (2, 2)
    assert frame(p2) == (1, 2) and frame(a3) == (2, 3) and frame(b3) == (3, 4) and frame(p3) == (2, 3)
