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
    # ---- 3x3 convolutions by Winograd F(2x2, 3x3) (cnn_wino.hip) against float64 and against the direct kernel
    for (N, H, Cin, Cout, ldo, off) in [(2, 16, 96, 208, 512, 160), (1, 64, 64, 192, 192, 0), (3, 32, 128, 192, 480, 128), (5, 8, 192, 384, 1024, 384),
                                         (7, 8, 48, 128, 128, 0), (2, 32, 16, 32, 256, 192), (1, 16, 32, 64, 64, 0), (6, 8, 160, 320, 832, 256),
                                         (2, 48, 32, 40, 40, 0), (3, 16, 24, 64, 64, 0)]:
        W = H
        assert L.sf_cnn_wino_ok(H, W, Cin)
        x = torch.relu(torch.randn((N, H, W, Cin), generator=g))
        w = torch.randn((Cout, 9, Cin), generator=g) / np.sqrt(9 * Cin)
        b = torch.randn((Cout,), generator=g)
        ref = F.conv2d(x.permute(0, 3, 1, 2).double(), w.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2).double(), b.double(),
                       padding=1).relu().permute(0, 2, 3, 1)
        xd, wd, bd = x.cuda(), w.cuda(), b.cuda()
        U = torch.empty(L.sf_cnn_wino_weight_floats(Cout, Cin), dtype=torch.float32, device="cuda")
        _ffi.check(L.sf_cnn_wino_weights(_ffi.ptr(wd), Cout, Cin, _ffi.ptr(U), _ffi.stream_ptr()), "wino_weights")
        out = torch.full((N, H, W, ldo), -7.0, device="cuda")
        _ffi.check(L.sf_cnn_conv3x3_wino(_ffi.ptr(xd), N, H, W, Cin, Cin, _ffi.ptr(U), _ffi.ptr(bd), Cout, _ffi.ptr(out), ldo, off,
                                         _ffi.stream_ptr()), "conv3x3_wino")
        direct = torch.full((N, H, W, ldo), -7.0, device="cuda")
        _ffi.check(L.sf_cnn_conv(_ffi.ptr(xd), N, H, W, Cin, Cin, _ffi.ptr(wd), _ffi.ptr(bd), Cout, 3, _ffi.ptr(direct), ldo, off,
                                 _ffi.stream_ptr()), "conv")
        torch.cuda.synchronize()
        got = out.cpu().double()
        scale = max(ref.abs().max().item(), 1e-30)
        err = (got[..., off:off + Cout] - ref).abs().max().item() / scale
        errd = (direct.cpu().double()[..., off:off + Cout] - ref).abs().max().item() / scale
        untouched = bool((got[..., :off] == -7.0).all()) and bool((got[..., off + Cout:] == -7.0).all())
        print("winograd N%d %dx%d Cin %d Cout %d -> ld %d off %d: max err %.2e (direct kernel %.2e)  untouched %s"
              % (N, H, W, Cin, Cout, ldo, off, err, errd, untouched))
        assert err < 5e-6 and untouched
    print("winograd ok")
    if "--exp" in sys.argv:
        import time
        N, H, Cin, Cout = 512, 64, 64, 192
        x = torch.relu(torch.randn((N, H, H, Cin), device="cuda"))
        w = torch.randn((Cout, 9, Cin), device="cuda") / np.sqrt(9 * Cin)
        b = torch.randn((Cout,), device="cuda")
        U = torch.empty(L.sf_cnn_wino_weight_floats(Cout, Cin), dtype=torch.float32, device="cuda")
        L.sf_cnn_wino_weights(_ffi.ptr(w), Cout, Cin, _ffi.ptr(U), _ffi.stream_ptr())
        out = torch.empty((N, H, H, Cout), device="cuda")
        for e in (0, 1, 2, 3, 4, 7, 8, 15):
            L.sf_debug_set(17, 10 + e if e else 0)
            def run():
                L.sf_cnn_conv3x3_wino(_ffi.ptr(x), N, H, H, Cin, Cin, _ffi.ptr(U), _ffi.ptr(b), Cout, _ffi.ptr(out), Cout, 0, _ffi.stream_ptr())
            run(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): run()
            torch.cuda.synchronize()
            print("EXP %2d (1 no staging/loads, 2 no transform, 4 no MFMA, 8 no stores): %.1f us" % (e, (time.perf_counter() - t0) / 5 * 1e6))
        L.sf_debug_set(17, 0)
    if "--time" in sys.argv:
        import time
        for (N, H, Cin, Cout) in [(512, 64, 64, 192), (512, 32, 96, 128), (512, 32, 128, 192), (512, 16, 160, 320), (512, 16, 96, 208), (512, 8, 192, 384),
                                  (512, 32, 32, 96), (512, 16, 32, 128)]:
            x = torch.relu(torch.randn((N, H, H, Cin), device="cuda"))
            w = torch.randn((Cout, 9, Cin), device="cuda") / np.sqrt(9 * Cin)
            b = torch.randn((Cout,), device="cuda")
            U = torch.empty(L.sf_cnn_wino_weight_floats(Cout, Cin), dtype=torch.float32, device="cuda")
            L.sf_cnn_wino_weights(_ffi.ptr(w), Cout, Cin, _ffi.ptr(U), _ffi.stream_ptr())
            out = torch.empty((N, H, H, Cout), device="cuda")
            res = {}
            for name in ("direct", "winograd"):
                def run():
                    if name == "direct":
                        L.sf_cnn_conv(_ffi.ptr(x), N, H, H, Cin, Cin, _ffi.ptr(w), _ffi.ptr(b), Cout, 3, _ffi.ptr(out), Cout, 0, _ffi.stream_ptr())
                    else:
                        L.sf_cnn_conv3x3_wino(_ffi.ptr(x), N, H, H, Cin, Cin, _ffi.ptr(U), _ffi.ptr(b), Cout, _ffi.ptr(out), Cout, 0, _ffi.stream_ptr())
                run(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(5): run()
                torch.cuda.synchronize()
                res[name] = (time.perf_counter() - t0) / 5 * 1e6
            fl = 2.0 * N * H * H * Cin * 9 * Cout
            print("time N%d %dx%d %d->%d: direct %.1f us (%.1f TF/s)  winograd %.1f us (%.1f TF/s nominal)  x%.2f"
                  % (N, H, H, Cin, Cout, res["direct"], fl / res["direct"] / 1e6, res["winograd"], fl / res["winograd"] / 1e6, res["direct"] / res["winograd"]))


if __name__ == "__main__":
    main()
