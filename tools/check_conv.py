#!/usr/bin/env python3
"""sf_cnn_conv / sf_cnn_conv_split3 on random operands against a float64 torch-CPU convolution (GPU box):
   python tools/check_conv.py [--knob 17=1]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import torch.nn.functional as F
    from srcfinder_amd import _ffi
    L = _ffi.lib()
    for kv in sys.argv[1:]:
        if "=" in kv:
            k, v = kv.replace("--knob", "").strip().split("=")
            L.sf_debug_set(int(k), int(v))
    g = torch.Generator().manual_seed(7)
    worst = 0.0
    # (N, H, W, Cin, ld_in, Cout, ks, ld_out, ch_off)
    cases = [(2, 8, 8, 32, 32, 64, 1, 64, 0), (3, 5, 7, 16, 16, 40, 3, 48, 8), (1, 9, 6, 24, 24, 64, 3, 64, 0),
             (2, 16, 16, 64, 64, 192, 3, 192, 0), (1, 32, 32, 192, 192, 176, 1, 176, 0), (5, 8, 8, 832, 832, 384, 1, 1024, 100),
             (2, 13, 11, 48, 48, 128, 3, 128, 0), (1, 7, 5, 8, 8, 20, 3, 20, 0), (2, 8, 8, 96, 128, 208, 3, 256, 48),
             (7, 4, 4, 160, 160, 320, 3, 320, 0), (1, 3, 3, 32, 32, 33, 1, 36, 3), (4, 6, 6, 112, 112, 224, 3, 224, 0)]
    for (N, H, W, Cin, ldi, Cout, ks, ldo, off) in cases:
        x = torch.randn((N, H, W, ldi), generator=g)
        x = torch.relu(x)                                   # activations are ReLU outputs in the graph
        w = torch.randn((Cout, ks * ks, Cin), generator=g) / np.sqrt(ks * ks * Cin)
        b = torch.randn((Cout,), generator=g)
        ref = F.conv2d(x[..., :Cin].permute(0, 3, 1, 2).double(), w.view(Cout, ks, ks, Cin).permute(0, 3, 1, 2).double(),
                       b.double(), padding=ks // 2).relu().permute(0, 2, 3, 1)
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
        out = torch.full((N, H, W, ldo), -7.0, device="cuda")
        _ffi.check(L.sf_cnn_conv(_ffi.ptr(xd), N, H, W, Cin, ldi, _ffi.ptr(wd), _ffi.ptr(bd), Cout, ks, _ffi.ptr(out), ldo, off,
                                 _ffi.stream_ptr()), "conv")
        torch.cuda.synchronize()
        got = out.cpu().double()
        err = (got[..., off:off + Cout] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        untouched = bool((got[..., :off] == -7.0).all()) and bool((got[..., off + Cout:] == -7.0).all())
        worst = max(worst, err)
        print("N%d %dx%d Cin %d (ld %d) Cout %d k%d -> ld %d off %d: max err %.2e  untouched %s"
              % (N, H, W, Cin, ldi, Cout, ks, ldo, off, err, untouched))
        assert err < 2e-6 and untouched
    print("conv ok, worst %.2e" % worst)
    # ---- inception branch 4: 3x3 s1 p1 max pool + 1x1 convolution in one launch (k_poolconv) against pool-then-convolve
    #      (sf_debug_set(18, 3): bit-identical) and against float64
    for (N, H, W, Cin, Cout, ldo, off) in [(3, 32, 32, 192, 32, 256, 224), (2, 32, 32, 256, 64, 480, 416), (5, 16, 16, 480, 64, 512, 448),
                                            (3, 16, 16, 528, 128, 832, 704), (9, 8, 8, 832, 128, 1024, 896), (1, 16, 16, 512, 64, 64, 0),
                                            (7, 8, 8, 32, 40, 40, 0)]:
        x = torch.relu(torch.randn((N, H, W, Cin), generator=g))
        w = torch.randn((Cout, 1, Cin), generator=g) / np.sqrt(Cin)
        b = torch.randn((Cout,), generator=g)
        pooled = F.max_pool2d(x.permute(0, 3, 1, 2).double(), 3, 1, 1)
        ref = F.conv2d(pooled, w.view(Cout, 1, 1, Cin).permute(0, 3, 1, 2).double(), b.double()).relu().permute(0, 2, 3, 1)
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
        scratch = torch.empty_like(xd)
        outs = []
        for variant in (0, 3):
            L.sf_debug_set(18, variant)
            out = torch.full((N, H, W, ldo), -7.0, device="cuda")
            _ffi.check(L.sf_cnn_pool_conv(_ffi.ptr(xd), N, H, W, Cin, Cin, _ffi.ptr(wd), _ffi.ptr(bd), Cout, _ffi.ptr(out), ldo, off,
                                          _ffi.ptr(scratch), _ffi.stream_ptr()), "pool_conv")
            torch.cuda.synchronize()
            outs.append(out.cpu())
        L.sf_debug_set(18, 0)
        got = outs[0].double()
        err = (got[..., off:off + Cout] - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        same = bool(torch.equal(outs[0], outs[1]))
        print("pool+conv N%d %dx%d Cin %d Cout %d -> ld %d off %d: max err %.2e  == pool-then-convolve: %s" % (N, H, W, Cin, Cout, ldo, off, err, same))
        assert err < 2e-6 and same
    print("pool_conv ok")


if __name__ == "__main__":
    main()
