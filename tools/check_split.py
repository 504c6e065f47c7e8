#!/usr/bin/env python3
"""The split-operand convolution (csrc/cnn_split.hip: fp32 operands as fp16 hi + lo, three fp16 MFMAs, fp32 accumulate) one layer
at a time against float64, beside the direct fp32 kernel and the Winograd kernel: max error and microseconds per launch."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from srcfinder_amd import _ffi
L = _ffi.lib()
raw = C.CDLL(os.path.join(ROOT, "srcfinder_amd", "libsrcfinder_amd.so"))
vp, i32, f32 = C.c_void_p, C.c_int, C.c_float
raw.sf_cnn_split_weights.argtypes = [vp, i32, i32, vp, vp, vp, vp]
raw.sf_cnn_conv_split.argtypes = [vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, i32, i32, f32, vp, i32, f32, i32, i32, vp, vp]
P, st = _ffi.ptr, _ffi.stream_ptr
for kv in sys.argv[1:]:
    k, v = kv.split("="); L.sf_debug_set(int(k), int(v))
XS = float(os.environ.get("SF_CHECK_XSCALE", "1"))     # activation scale (float16's low halves go subnormal below ~0.25)
AS = float(os.environ.get("SF_CHECK_ASCALE", "1"))     # the layer's ascale (a power of two): SF_CHECK_XSCALE=2e-4 SF_CHECK_ASCALE=4096 ...
flag = torch.zeros(1, dtype=torch.int32, device="cuda:0")
dev = torch.device("cuda:0")
g = torch.Generator(device="cpu").manual_seed(5)

def layer(N, H, Cin, Cout, ks, timeit=True):
    x = (torch.relu(torch.randn(N, H, H, Cin, generator=g)) * XS).to(dev)
    w = (torch.randn(Cout, ks, ks, Cin, generator=g) * (0.7 / np.sqrt(ks * ks * Cin))).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    K = ks * ks * Cin
    hi = torch.empty(Cout * K, dtype=torch.float16, device=dev); lo = torch.empty_like(hi)
    sc = torch.empty(Cout, dtype=torch.float32, device=dev)
    assert raw.sf_cnn_split_weights(P(w), Cout, K, P(hi), P(lo), P(sc), st()) == 0
    o_sp = torch.empty(N, H, H, Cout, dtype=torch.float32, device=dev); o_d = torch.empty_like(o_sp)
    run_sp = lambda: raw.sf_cnn_conv_split(P(x), 0, N, H, H, Cin, Cin, P(hi), P(lo), P(sc), P(b), Cout, ks, AS, P(o_sp), 0, 1.0, Cout, 0, P(flag), st())
    run_d = lambda: L.sf_cnn_conv(P(x), N, H, H, Cin, Cin, P(w), P(b), Cout, ks, P(o_d), Cout, 0, st())
    assert run_sp() == 0 and run_d() == 0
    torch.cuda.synchronize()
    n = min(N, 2)
    ref = torch.nn.functional.conv2d(x[:n].double().permute(0, 3, 1, 2), w.double().permute(0, 3, 1, 2), b.double(), padding=ks // 2)
    ref = torch.relu(ref).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    e_sp = float((o_sp[:n].double() - ref).abs().max()) / scale
    e_d = float((o_d[:n].double() - ref).abs().max()) / scale
    line = "N%d %dx%d %d->%d k%d: max err / max|ref|  split %.2e  direct %.2e" % (N, H, H, Cin, Cout, ks, e_sp, e_d)
    if timeit:
        res = {}
        for name, fn in (("split", run_sp), ("direct", run_d)):
            fn(); torch.cuda.synchronize()
            a, bb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(5): fn()
            bb.record(); torch.cuda.synchronize()
            res[name] = a.elapsed_time(bb) / 5 * 1e3
        fl = 2.0 * N * H * H * Cin * Cout * ks * ks
        line += "   us: split %.1f (%.0f TF/s)  direct %.1f (%.0f TF/s)  x%.2f" % (res["split"], fl / res["split"] / 1e6, res["direct"], fl / res["direct"] / 1e6, res["direct"] / res["split"])
    print(line, flush=True)

for cfg in ((3, 16, 24, 40, 3), (2, 32, 64, 96, 1), (5, 8, 160, 320, 3)):
    layer(*cfg, timeit=False)
for cfg in ((512, 64, 64, 192, 3), (512, 32, 128, 192, 3), (512, 16, 160, 320, 3), (512, 8, 192, 384, 3),
            (512, 32, 192, 176, 1), (512, 32, 256, 288, 1), (512, 16, 512, 296, 1), (512, 8, 832, 624, 1), (512, 64, 64, 64, 1)):
    layer(*cfg)
