"""CNN tile scorer on MI355X -- host side of the drop-in boundary for ``cnn/cnn_pred_pipeline.py``.

Mirrors the reference's surface:

* ``ClampCH4(vmin=250, vmax=4000)``                                   cnn_pred_pipeline.py:19-30
* ``FlightlineConvolve(flightline, transform, dim=256, device)`` with ``.x``, ``.inshape``, ``.dim``,
  ``__len__``, ``__getitem__``                                         cnn_pred_pipeline.py:32-58
* ``MODEL_NORM`` per-model mean/std                                    cnn_pred_pipeline.py:126-157
* ``GoogLeNetHIP.load_state_dict(sd)`` accepts the ``state_dict`` of ``googlenet(num_classes=2)``
  (aux heads and ``num_batches_tracked`` ignored)                      googlenet1.py:27-181
* ``predict_flightline(cmf2d, model|(mean,std), weights, batch)`` = the script's batch loop + reshape +
  NODATA rule                                                          cnn_pred_pipeline.py:159-189

Every operator runs in a hand-written HIP kernel behind ``libsrcfinder_amd.so``; there is no torch.nn fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _ffi
from .cnn_weights import INCEPTION, conv_table

MODEL_NORM = {
    "COVID_QC": (110.6390, 183.9152), "CalCH4_v8": (140.6399, 237.5434), "Permian_QC": (100.2635, 158.7060),
    "multi_256": (115.0, 190.0), "multi_64": (115.0, 190.0),
}
BN_EPS = 0.001          # googlenet1.py:270
NODATA = -9999.0
# The convolutions' arithmetic routes (all float32 in / float32 accumulate; include/srcfinder_amd.h: sf_cnn_score_rows).
# "split": operand splitting on the fp16 matrix cores; through the C driver (sf_cnn_score_rows: whole image rows) with the trunk through
# inception3b SHARED between the overlapping windows (csrc/cnn_share.hip); sequenced from Python one batch at a time (forward_tiles)
# every window on its own -- "split_unshared" names that form on both drivers; "split_conv3": the sharing stops behind conv3 (round
# 6's first form).  The three give the same bits.  "winograd" / "direct": the fp32 matrix cores (the split route's rescue path).
ROUTES = {"split": 0, "split_conv3": 5, "split_unshared": 3, "winograd": 4, "direct": 2, "direct_pointer": 1}


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise _ffi.SrcfinderError("no GPU visible: srcfinder_amd has no CPU fallback")
    return torch


class ClampCH4(object):
    """Preprocessing step for the methane layer (same contract as the reference class)."""

    def __init__(self, vmin=250, vmax=4000):
        assert isinstance(vmin, int) and isinstance(vmax, int) and vmax > vmin
        self.vmin = vmin
        self.vmax = vmax

    def __call__(self, T):
        import torch
        return torch.clamp(T, self.vmin, self.vmax)

    def __repr__(self):
        return self.__class__.__name__ + '(vmin={0}, vmax={1})'.format(self.vmin, self.vmax)


class Normalize(object):
    """(x - mean) / std, the one ``torchvision.transforms`` op the script uses (:128-156)."""

    def __init__(self, mean, std):
        self.mean, self.std = float(np.ravel(mean)[0]), float(np.ravel(std)[0])


class Compose(object):
    def __init__(self, transforms):
        self.transforms = list(transforms)


def _parse_transform(transform):
    """(vmin, vmax, mean, std) from a Compose([ClampCH4, Normalize]) / tuple / model name."""
    if isinstance(transform, str):
        m, s = MODEL_NORM[transform]
        return 0.0, 4000.0, m, s
    if isinstance(transform, (tuple, list)) and len(transform) == 2 and not isinstance(transform[0], (ClampCH4, Normalize)):
        return 0.0, 4000.0, float(transform[0]), float(transform[1])
    ts = transform.transforms if isinstance(transform, Compose) else list(transform)
    vmin, vmax, mean, std = None, None, 0.0, 1.0
    for t in ts:
        if isinstance(t, ClampCH4):
            vmin, vmax = float(t.vmin), float(t.vmax)
        elif isinstance(t, Normalize):
            mean, std = t.mean, t.std
        else:
            raise TypeError("unsupported transform %r" % (t,))
    if vmin is None:
        raise TypeError("transform must contain ClampCH4")
    return vmin, vmax, mean, std


class FlightlineConvolve(object):
    """Single flightline for exhaustive CNN convolution (cnn_pred_pipeline.py:32-58).

    ``flightline`` is the 2-D CMF plane (ndarray / tensor; the reference reads band 1 of a raster).
    ``self.x`` is the clamped, normalised, zero-padded plane ``[1, H+dim-1, W+dim-1]`` on the GPU, produced by the
    HIP ``sf_cnn_prepare_plane`` kernel; item ``i`` is the window ``x[:, row:row+dim, col:col+dim]``.
    """

    def __init__(self, flightline, transform, dim=256, device=None):
        torch = _torch()
        self.flightline = flightline
        self.transform = transform
        dev = torch.device("cuda") if device is None else torch.device(device)
        plane = flightline if torch.is_tensor(flightline) else torch.as_tensor(np.ascontiguousarray(flightline, dtype=np.float32))
        plane = plane.to(dev, torch.float32).contiguous()
        if plane.dim() != 2:
            raise TypeError("flightline must be a 2-D plane")
        H, W = plane.shape
        self.plane = plane
        self.inshape = (1, H, W)
        self.dim = dim
        vmin, vmax, mean, std = _parse_transform(transform)
        self.x = torch.empty((1, H + dim - 1, W + dim - 1), dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            _ffi.check(_ffi.lib().sf_cnn_prepare_plane(_ffi.ptr(plane), H, W, vmin, vmax, mean, std, dim, _ffi.ptr(self.x),
                                                       _ffi.stream_ptr()), "sf_cnn_prepare_plane")

    def __len__(self):
        return int(self.inshape[1] * self.inshape[2])

    def __getitem__(self, idx):
        row = idx // self.inshape[2]
        col = idx % self.inshape[2]
        return self.x[:, row:row + self.dim, col:col + self.dim]


def _knob(key):
    """The calling thread's value of a tuning knob of the library (sf_debug_set)."""
    v = C.c_int(0)
    _ffi.lib().sf_debug_get(int(key), C.byref(v))
    return v.value


def _pool_out(n, k, s, p):
    """ceil_mode output size with PyTorch's last-window rule."""
    o = -(-(n + 2 * p - k) // s) + 1
    if (o - 1) * s >= n + p:
        o -= 1
    return o


class GoogLeNetHIP(object):
    """Eval graph of the reference's 1-channel GoogLeNet (googlenet1.py:60-89, :110-163) on HIP kernels."""

    def __init__(self, state_dict=None, device=None, precision="fp32"):
        """precision: "fp32" (the parity path: exact fp32 FMA chains on the fp32 matrix cores) or "fp16" (float16
        activations/weights with fp32 accumulation -- the precision class of cuDNN's TF32 default, ~4x faster,
        own tolerance)."""
        torch = _torch()
        if precision not in ("fp32", "fp16"):
            raise ValueError("precision must be 'fp32' or 'fp16'")
        self.precision = precision
        self.half = precision == "fp16"
        self.adt = torch.float16 if self.half else torch.float32
        self.sfx = "_f16" if self.half else ""
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self.w = {}
        self.wino = {}            # name -> U [16][Cout][Cin]: the 3 x 3 layers' Winograd-domain weights (fp32 path; csrc/cnn_wino.hip)
        self.winograd = True      # False: every 3 x 3 convolution through the direct implicit-GEMM kernel (the tests' cross-check)
        self.split = {}           # name -> (hi, lo, scale): fp16 halves of the folded weights for the split-operand kernels (csrc/cnn_split.hip)
        self._bufs = {}
        # the split-operand route's per-layer activation scales (powers of two; include/srcfinder_amd.h: sf_cnn_calibrate).  1 until
        # calibrate() has seen a plane; predict_flightline calibrates on every plane it scores
        self.ascale = [1.0] * int(_ffi.lib().sf_cnn_num_scales())
        self.route = None         # None: the calling thread's sf_debug_set(17, .) (0 unless a tool set it); else ROUTES / a code
        self._route, self._flag = 0, None      # (state of the forward pass in progress)
        if state_dict is not None:
            self.load_state_dict(state_dict)

    def load_state_dict(self, sd):
        """Fold BN into conv weight/bias (float64 on the host, stored float32), reorder OIHW -> [O][kh*kw][I]."""
        torch = _torch()

        def npy(v):
            return v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)

        for name, cin, cout, k, s, p in conv_table():
            w = npy(sd[name + ".conv.weight"]).astype(np.float64)
            g = npy(sd[name + ".bn.weight"]).astype(np.float64)
            b = npy(sd[name + ".bn.bias"]).astype(np.float64)
            m = npy(sd[name + ".bn.running_mean"]).astype(np.float64)
            v = npy(sd[name + ".bn.running_var"]).astype(np.float64)
            assert w.shape == (cout, cin, k, k), (name, w.shape)
            scale = g / np.sqrt(v + BN_EPS)
            wf = (w * scale[:, None, None, None]).transpose(0, 2, 3, 1).reshape(cout, k * k, cin)
            bf = b - m * scale
            wdt = self.adt if name != "conv1" else torch.float32      # conv1 runs on the fp32 VALU in both modes
            self.w[name] = (torch.as_tensor(np.ascontiguousarray(wf, dtype=np.float32)).to(self.device).to(wdt),
                            torch.as_tensor(bf.astype(np.float32)).to(self.device))
        self.wino = {}
        if not self.half:
            # the 3 x 3 stride-1 convolutions run by Winograd F(2 x 2, 3 x 3) where the geometry allows: U = G g G^T once per upload
            L = _ffi.lib()
            with torch.cuda.device(self.device):
                for name, cin, cout, k, s_, p_ in conv_table():
                    if k == 3 and s_ == 1 and cin % 8 == 0:
                        U = torch.empty(int(L.sf_cnn_wino_weight_floats(cout, cin)), dtype=torch.float32, device=self.device)
                        _ffi.check(L.sf_cnn_wino_weights(_ffi.ptr(self.w[name][0]), cout, cin, _ffi.ptr(U), _ffi.stream_ptr()),
                                   "sf_cnn_wino_weights(%s)" % name)
                        self.wino[name] = U
        for spec in INCEPTION:       # stacked weights of the three 1x1 convs that share the block input
            name = spec[0]
            ws = [self.w[name + s_][0] for s_ in (".branch1", ".branch2.0", ".branch3.0")]
            bs = [self.w[name + s_][1] for s_ in (".branch1", ".branch2.0", ".branch3.0")]
            self.w[name + ".head3"] = (torch.cat(ws, 0).contiguous(), torch.cat(bs, 0).contiguous())
        self.split = {}
        if not self.half:
            # every convolution but conv1 and the pool-projections runs by operand splitting on the fp16 matrix cores (the default;
            # sf_debug_set(17, 4): Winograd / fp32 matrix cores, 2: the direct fp32 kernel): hi | lo halves + per-channel scales
            L = _ffi.lib()
            with torch.cuda.device(self.device):
                for name in self.w:
                    if name == "conv1" or name.endswith(".branch1") or name.endswith(".0"):
                        continue
                    w, _b = self.w[name]
                    cout, K = w.shape[0], w.shape[1] * w.shape[2]
                    hi = torch.empty(cout * K, dtype=torch.float16, device=self.device)
                    lo = torch.empty_like(hi)
                    sc = torch.empty(cout, dtype=torch.float32, device=self.device)
                    _ffi.check(L.sf_cnn_split_weights(_ffi.ptr(w), cout, K, _ffi.ptr(hi), _ffi.ptr(lo), _ffi.ptr(sc), _ffi.stream_ptr()),
                               "sf_cnn_split_weights(%s)" % name)
                    self.split[name] = (hi, lo, sc)
        self.fcw = torch.as_tensor(np.ascontiguousarray(npy(sd["fc.weight"]), dtype=np.float32)).to(self.device)
        self.fcb = torch.as_tensor(np.ascontiguousarray(npy(sd["fc.bias"]), dtype=np.float32)).to(self.device)
        if self.fcw.shape != (2, 1024):
            raise ValueError("fc.weight must be [2, 1024]")
        return self

    def packed_blob(self):
        """The folded float32 weights as the ONE array ``sf_cnn_score_rows`` takes (layout: include/srcfinder_amd.h)."""
        torch = _torch()
        if self.half:
            raise ValueError("the C driver scores in fp32")
        if getattr(self, "_blob", None) is None:
            parts = []
            for name in ("conv1", "conv2", "conv3"):
                parts += [self.w[name][0].reshape(-1), self.w[name][1]]
            for spec in INCEPTION:
                for sfx in (".head3", ".branch2.1", ".branch3.1", ".branch4.1"):
                    w, b = self.w[spec[0] + sfx]
                    parts += [w.reshape(-1), b]
            parts += [self.fcw.reshape(-1), self.fcb]
            self._blob = torch.cat([p.to(torch.float32) for p in parts]).contiguous()
            assert self._blob.numel() == _ffi.lib().sf_cnn_blob_floats()
        return self._blob

    # -- buffers ---------------------------------------------------------------------------------------------
    def _buf(self, key, shape):
        torch = _torch()
        n = int(np.prod(shape))
        b = self._bufs.get(key)
        if b is None or b.numel() < n:
            self._bufs[key] = None
            b = torch.empty(n, dtype=self.adt, device=self.device)
            self._bufs[key] = b
        return b[:n].view(*shape)

    # -- the convolutions' route and the split-operand route's range contracts --------------------------------
    def _route_code(self, route=None):
        """0 operand splitting (fp16 matrix cores, fp32 tolerance class; default), 4 Winograd + fp32 matrix cores, 2 / 1 the direct
        fp32 kernel.  An explicit argument wins, then ``self.route``, then the calling thread's tuning knob 17 (tools)."""
        r = self.route if route is None else route
        if r is None:
            return 0 if self.half else _knob(17)
        r = ROUTES.get(r, r) if isinstance(r, str) else int(r)
        if r not in (0, 1, 2, 3, 4, 5):
            raise ValueError("route must be one of %r or 0 / 5 / 3 / 4 / 2 / 1" % (sorted(ROUTES),))
        return r

    def overflow_slots(self, n):
        """``n`` zeroed device ints for ``forward_tiles(..., overflow=slots[i:i+1])``: one per batch, read once at the end."""
        torch = _torch()
        return torch.zeros(int(n), dtype=torch.int32, device=self.device)

    def calibrate(self, ds, batch=64):
        """Per-layer activation scales of the split-operand route from a fixed sample of the plane's windows
        (``sf_cnn_calibrate``: a function of the plane and the weights alone).  ``ds``: a FlightlineConvolve."""
        torch = _torch()
        if self.half:
            return self.ascale
        L = _ffi.lib()
        H, W = ds.inshape[1], ds.inshape[2]
        with torch.cuda.device(self.device):
            wsb = L.sf_cnn_score_workspace_bytes(int(batch), 0, 0)
            ws = self._buf("c_driver_ws", ((wsb + 3) // 4,))
            sc = (C.c_float * len(self.ascale))()
            _ffi.check(L.sf_cnn_calibrate(_ffi.ptr(ds.x), H, W, _ffi.ptr(self.packed_blob()), int(batch), _ffi.ptr(ws),
                                          C.c_size_t(ws.numel() * 4), sc, _ffi.stream_ptr()), "sf_cnn_calibrate")
        self.ascale = [float(v) for v in sc]
        return self.ascale

    def _begin(self, route, overflow):
        """Set the pass's route and overflow slot; returns True when this call owns the slot (and must check it itself)."""
        torch = _torch()
        self._route = self._route_code(route)
        if self._route in (3, 5):
            self._route = 0        # (the Python-sequenced graph always evaluates every window on its own)
        own = False
        if self._route == 0 and not self.half:
            if overflow is None:
                if getattr(self, "_own_flag", None) is None:
                    self._own_flag = torch.zeros(1, dtype=torch.int32, device=self.device)
                self._own_flag.zero_()
                overflow, own = self._own_flag, True
            same = overflow.is_cuda and (self.device.index is None or overflow.device.index == self.device.index)
            if overflow.dtype != torch.int32 or overflow.numel() < 1 or not same:
                raise ValueError("overflow must be an int32 tensor on the network's device")
        self._flag = overflow
        return own

    # -- operators -------------------------------------------------------------------------------------------
    def _conv(self, x, name, out, ch_off, a_in=1.0, a_out=1.0):
        L = _ffi.lib()
        w, b = self.w[name]
        N, H, W, ldi = x.shape
        cout, taps, cin = w.shape
        k = 3 if taps == 9 else 1
        mode = self._route
        if mode == 0 and name in self.split:
            # tensors only split-operand convolutions read travel in the split format (csrc/cnn_split.hip): conv2's output, and the
            # 3 x 3 reducers' outputs that _inception's split3 call leaves in t2 / t3 -- scaled by their consumer's activation scale
            hi, lo, sc = self.split[name]
            in_split = 1 if (k == 3) else 0
            out_split = 1 if name == "conv2" else 0
            _ffi.check(L.sf_cnn_conv_split(_ffi.ptr(x), in_split, N, H, W, cin, ldi, _ffi.ptr(hi), _ffi.ptr(lo), _ffi.ptr(sc), _ffi.ptr(b),
                                           cout, k, C.c_float(a_in), _ffi.ptr(out), out_split, C.c_float(a_out), out.shape[3], ch_off,
                                           _ffi.ptr(self._flag), _ffi.stream_ptr()),
                       "sf_cnn_conv_split(%s)" % name)
            return
        if k == 3 and self.winograd and name in self.wino and L.sf_cnn_wino_ok(H, W, cin) and mode == 4:
            _ffi.check(L.sf_cnn_conv3x3_wino(_ffi.ptr(x), N, H, W, cin, ldi, _ffi.ptr(self.wino[name]), _ffi.ptr(b), cout,
                                             _ffi.ptr(out), out.shape[3], ch_off, _ffi.stream_ptr()), "sf_cnn_conv3x3_wino(%s)" % name)
            return
        fn = getattr(L, "sf_cnn_conv" + self.sfx)
        _ffi.check(fn(_ffi.ptr(x), N, H, W, cin, ldi, _ffi.ptr(w), _ffi.ptr(b), cout, k, _ffi.ptr(out),
                      out.shape[3], ch_off, _ffi.stream_ptr()), "sf_cnn_conv(%s)" % name)

    def _pool(self, x, key, k, s, p):
        L = _ffi.lib()
        N, H, W, Cc = x.shape
        Ho, Wo = _pool_out(H, k, s, p), _pool_out(W, k, s, p)
        out = self._buf(key, (N, Ho, Wo, Cc))
        fn = getattr(L, "sf_cnn_maxpool" + self.sfx)
        _ffi.check(fn(_ffi.ptr(x), N, H, W, Cc, k, s, p, _ffi.ptr(out), Ho, Wo, _ffi.stream_ptr()), "sf_cnn_maxpool")
        return out

    def _inception(self, x, spec, blk=0):
        name, cin, c1, c3r, c3, c5r, c5, pp = spec
        split = not self.half and self._route == 0
        ax, a2, a3 = (self.ascale[2 + 3 * blk], self.ascale[3 + 3 * blk], self.ascale[4 + 3 * blk]) if split else (1.0, 1.0, 1.0)
        N, H, W, _ = x.shape
        y = self._buf(name + ".y", (N, H, W, c1 + c3 + c5 + pp))
        t2 = self._buf("t2", (N, H, W, c3r))
        t3 = self._buf("t3", (N, H, W, c5r))
        w3, b3 = self.w[name + ".head3"]            # branch1 | branch2[0] | branch3[0] in one GEMM
        if split and (name + ".head3") in self.split:
            hi, lo, sc = self.split[name + ".head3"]
            _ffi.check(_ffi.lib().sf_cnn_conv_split3_split(_ffi.ptr(x), N, H, W, cin, x.shape[3], _ffi.ptr(hi), _ffi.ptr(lo), _ffi.ptr(sc),
                                                           _ffi.ptr(b3), c1, c3r, c5r, C.c_float(ax), _ffi.ptr(y), y.shape[3], 0,
                                                           _ffi.ptr(t2), c3r, 0, _ffi.ptr(t3), c5r, 0, 1, C.c_float(a2), C.c_float(a3),
                                                           _ffi.ptr(self._flag), _ffi.stream_ptr()),
                       "sf_cnn_conv_split3_split(%s)" % name)
        else:
            fn = getattr(_ffi.lib(), "sf_cnn_conv_split3" + self.sfx)
            _ffi.check(fn(_ffi.ptr(x), N, H, W, cin, x.shape[3], _ffi.ptr(w3), _ffi.ptr(b3), c1, c3r, c5r, _ffi.ptr(y),
                          y.shape[3], 0, _ffi.ptr(t2), c3r, 0, _ffi.ptr(t3), c5r, 0, _ffi.stream_ptr()),
                       "sf_cnn_conv_split3(%s)" % name)
        self._conv(t2, name + ".branch2.1", y, c1, a_in=a2)
        self._conv(t3, name + ".branch3.1", y, c1 + c3, a_in=a3)
        if self.sfx or x.shape[3] != cin:           # fp16 path / strided input: pool, then convolve
            pooled = self._pool(x, "pool_s1", 3, 1, 1)
            self._conv(pooled, name + ".branch4.1", y, c1 + c3 + c5, a_in=ax)
        elif split and (name + ".branch4.1") in self.split and \
                _ffi.lib().sf_cnn_pool_conv_split_ok(N, H, W, cin, self.w[name + ".branch4.1"][0].shape[0]):
            hi, lo, sc = self.split[name + ".branch4.1"]
            w4, b4 = self.w[name + ".branch4.1"]
            _ffi.check(_ffi.lib().sf_cnn_pool_conv_split(_ffi.ptr(x), N, H, W, cin, _ffi.ptr(hi), _ffi.ptr(lo), _ffi.ptr(sc), _ffi.ptr(b4),
                                                         w4.shape[0], C.c_float(ax), _ffi.ptr(y), y.shape[3], c1 + c3 + c5,
                                                         _ffi.ptr(self._flag), _ffi.stream_ptr()),
                       "sf_cnn_pool_conv_split(%s)" % name)
        else:                                       # pool + 1x1 convolution in one C call
            w4, b4 = self.w[name + ".branch4.1"]
            scratch = self._buf("pool_s1", (N, H, W, cin))
            _ffi.check(_ffi.lib().sf_cnn_pool_conv(_ffi.ptr(x), N, H, W, cin, cin, _ffi.ptr(w4), _ffi.ptr(b4), w4.shape[0],
                                                   _ffi.ptr(y), y.shape[3], c1 + c3 + c5, _ffi.ptr(scratch), _ffi.stream_ptr()),
                       "sf_cnn_pool_conv(%s)" % name)
        return y

    def _trunk(self, a1, taps=None, pooled=False):
        """maxpool1 .. inception5b on the conv1 activations a1 [N, H, W, 64] (googlenet1.py:61-86); with ``pooled`` a1 is
        already the output of maxpool1 (the fused conv1 + pool kernel of the tile scorer)."""
        x = a1 if pooled else self._pool(a1, "pool1", 3, 2, 0)
        split = not self.half and self._route == 0
        s0, s1 = (self.ascale[0], self.ascale[1]) if split else (1.0, 1.0)
        a3 = self._buf("conv2", x.shape)
        self._conv(x, "conv2", a3, 0, a_in=s0, a_out=s1)
        a4 = self._buf("conv3", tuple(x.shape[:3]) + (192,))
        self._conv(a3, "conv3", a4, 0, a_in=s1)
        x = self._pool(a4, "pool2", 3, 2, 0)
        for blk, spec in enumerate(INCEPTION):
            x = self._inception(x, spec, blk)
            if taps is not None:
                taps[spec[0]] = x.clone()
            if spec[0] == "inception3b":
                x = self._pool(x, "pool3", 3, 2, 0)
            elif spec[0] == "inception4e":
                x = self._pool(x, "pool4", 2, 2, 0)
        return x

    def forward_fcn(self, canvas, out=None, route=None, overflow=None):
        """Fully convolutional pass (fcn_pred_pipeline.py:157-160, :229-231): canvas [N, Hc, Wc] float32 (already
        transformed) -> softmax(final_conv(trunk))[:, 1] as [N, Hc/32, Wc/32] float32.  ``route`` / ``overflow``: see
        :meth:`forward_tiles`."""
        torch = _torch()
        L = _ffi.lib()
        N, Hc, Wc = canvas.shape
        with torch.cuda.device(self.device):
            own = self._begin(route, overflow)
            st = _ffi.stream_ptr()
            Ho, Wo = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
            a1 = self._buf("conv1", (N, Ho, Wo, 64))
            w, b = self.w["conv1"]
            _ffi.check(L.sf_cnn_conv1_image(_ffi.ptr(canvas), N, Hc, Wc, _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(a1),
                                            int(self.half), st), "sf_cnn_conv1_image")
            x = self._trunk(a1)
            n, hq, wq, cc = x.shape
            if out is None:
                out = torch.empty((n, hq, wq), dtype=torch.float32, device=self.device)
            _ffi.check(getattr(L, "sf_cnn_head" + self.sfx)(_ffi.ptr(x), n * hq * wq, 1, cc, _ffi.ptr(self.fcw),
                                                            _ffi.ptr(self.fcb), None, C.c_longlong(0), NODATA,
                                                            _ffi.ptr(out), st), "sf_cnn_head")
            if own and int(self._flag.item()):
                _overflow_warning("a canvas batch")
                return self.forward_fcn(canvas, out=out, route=4)
        return out

    def forward_tiles(self, padded, width, tile0, ntiles, plane=None, out=None, taps=None, route=None, overflow=None):
        """Score tiles tile0..tile0+ntiles-1 of the padded plane; writes ``out[tile0:tile0+ntiles]`` (float32).

        ``route``: ``"split"`` (operand splitting on the fp16 matrix cores -- the float32 tolerance class; default), ``"winograd"``,
        ``"direct"`` -- an argument of THIS call (default: ``self.route``, else the calling thread's tuning knob 17).
        ``overflow`` (split route): a device int32 slot the kernels raise when an activation leaves float16's range.  Without one
        the call uses a slot of its own, reads it (one host synchronisation) and scores the batch AGAIN on the fp32 matrix cores
        when it is up -- never a silently wrong result.  A caller that wants to stay asynchronous passes one slot per batch
        (:meth:`overflow_slots`), reads them once at the end and repeats the raised batches with ``route="winograd"``
        (:func:`score_tiles` does exactly that)."""
        torch = _torch()
        L = _ffi.lib()
        Hp, Wp = padded.shape[-2], padded.shape[-1]
        with torch.cuda.device(self.device):
            own = self._begin(route, overflow)
            st = _ffi.stream_ptr()
            w, b = self.w["conv1"]
            if taps is None and not self.half and getattr(self, "fuse_conv1", True):
                # production: conv1 and maxpool1 in one kernel, the 4 MB-per-tile conv1 activation never reaches HBM
                a1 = self._buf("pool1", (ntiles, 64, 64, 64))
                _ffi.check(L.sf_cnn_conv1_pool(_ffi.ptr(padded), Hp, Wp, width, C.c_longlong(tile0), ntiles, _ffi.ptr(w),
                                               _ffi.ptr(b), _ffi.ptr(a1), st), "sf_cnn_conv1_pool")
                x = self._trunk(a1, None, pooled=True)
            else:
                a1 = self._buf("conv1", (ntiles, 128, 128, 64))
                _ffi.check(getattr(L, "sf_cnn_conv1" + self.sfx)(_ffi.ptr(padded), Hp, Wp, width, C.c_longlong(tile0), ntiles,
                                                                 _ffi.ptr(w), _ffi.ptr(b), _ffi.ptr(a1), st), "sf_cnn_conv1")
                x = self._trunk(a1, taps)
            if taps is not None:
                taps["conv1"] = a1.clone()
            N, H, W, Cc = x.shape
            if out is None:
                out = torch.empty(tile0 + ntiles, dtype=torch.float32, device=self.device)
            _ffi.check(getattr(L, "sf_cnn_head" + self.sfx)(_ffi.ptr(x), ntiles, H * W, Cc, _ffi.ptr(self.fcw),
                                                            _ffi.ptr(self.fcb), _ffi.ptr(plane), C.c_longlong(tile0),
                                                            NODATA, _ffi.ptr(out), st), "sf_cnn_head")
            if own and int(self._flag.item()):
                _overflow_warning("tiles %d..%d" % (tile0, tile0 + ntiles))
                return self.forward_tiles(padded, width, tile0, ntiles, plane=plane, out=out, taps=taps, route=4)
        return out


def _overflow_warning(what):
    import warnings
    warnings.warn("srcfinder_amd.cnn: an activation exceeded the float16 range of the split-operand kernels; %s scored again on "
                  "the fp32 matrix cores" % what)


def _score_rows_one(net, ds, r0, r1, batch, out, code, ws_key="c_driver_ws"):
    """ONE sf_cnn_score_rows call on image rows [r0, r1) on the calling thread's current stream; returns (re-scored, shared) batches."""
    torch = _torch()
    L = _ffi.lib()
    H, W = ds.inshape[1], ds.inshape[2]
    with torch.cuda.device(net.device):
        wsb = L.sf_cnn_score_workspace_bytes(int(batch), H if code in (0, 5) else 0, W if code in (0, 5) else 0)
        ws = net._buf(ws_key, ((wsb + 3) // 4,))
        nres = (C.c_int * 2)(0, 0)
        sc = (C.c_float * len(net.ascale))(*net.ascale)
        _ffi.check(L.sf_cnn_score_rows(_ffi.ptr(ds.x), _ffi.ptr(ds.plane), H, W, int(r0), int(r1), _ffi.ptr(net.packed_blob()),
                                       _ffi.ptr(out), int(batch), code, sc, nres, _ffi.ptr(ws),
                                       C.c_size_t(ws.numel() * 4), _ffi.stream_ptr()), "sf_cnn_score_rows")
    return int(nres[0]), int(nres[1])


# Two halves of a row range in flight on two HIP streams of the device (round 6): a batch is ~60 launches, many of them far smaller than
# the chip (the 8 x 8 layers, the ring kernels' side rows, the copies) -- a second batch fills their tails: 102 -> 108 k windows/s at
# batch 1024, 97 -> 105 k at 512.  Row ranges are independent (own workspace, own phase / strip maps, own overflow slots): the same bits.
LANES = 2


def _score_rows_c(net, ds, r0, r1, batch, out, code, lanes=None):
    """sf_cnn_score_rows on image rows [r0, r1) with the network's current activation scales; returns the batches re-scored.
    ``lanes`` (default LANES): contiguous parts of the range scored concurrently from host threads, one HIP stream and one workspace
    each, when every part has at least four batches."""
    torch = _torch()
    W = ds.inshape[2]
    lanes = LANES if lanes is None else int(lanes)
    lanes = max(1, min(lanes, ((r1 - r0) * W) // (4 * int(batch))))
    net.last_batches = -(-(r1 - r0) * W // int(batch))         # batches the call launches (every lane's last one may be short)
    if lanes == 1 or code not in (0, 3, 5):
        nres = _score_rows_one(net, ds, r0, r1, batch, out, code)
    else:
        import threading
        L = _ffi.lib()
        knobs = {}
        for key in (16, 17, 18):                                # the library's tuning knobs are per calling thread
            v = C.c_int(0)
            _ffi.check(L.sf_debug_get(key, C.byref(v)), "sf_debug_get")
            knobs[key] = v.value
        with torch.cuda.device(net.device):
            cur = torch.cuda.current_stream()
            if getattr(net, "_lane_streams", None) is None or len(net._lane_streams) < lanes:
                net._lane_streams = [torch.cuda.Stream(device=net.device) for _ in range(lanes)]
            net.packed_blob()                                    # (built once, here, not by two threads at a time)
        cuts = [r0 + (r1 - r0) * i // lanes for i in range(lanes + 1)]
        net.last_batches = sum(-(-(cuts[i + 1] - cuts[i]) * W // int(batch)) for i in range(lanes))
        res, errs = [None] * lanes, [None] * lanes

        def work(i):
            try:
                for key, v in knobs.items():
                    L.sf_debug_set(key, v)
                with torch.cuda.device(net.device):
                    s = net._lane_streams[i]
                    s.wait_stream(cur)                           # the plane and the weights were produced on the caller's stream
                    with torch.cuda.stream(s):
                        res[i] = _score_rows_one(net, ds, cuts[i], cuts[i + 1], batch, out, code, ws_key="c_driver_ws_lane%d" % i)
            except Exception as e:                               # surfaced in the caller's thread
                errs[i] = e

        threads = [threading.Thread(target=work, args=(i,)) for i in range(lanes)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        for e in errs:
            if e is not None:
                raise e
        with torch.cuda.device(net.device):
            for s in net._lane_streams[:lanes]:
                cur.wait_stream(s)
        nres = (sum(r[0] for r in res), sum(r[1] for r in res))
    net.last_shared_batches = nres[1]            # (batches of the call that ran on the shared trunk)
    if nres[0]:
        _overflow_warning("%d batch(es) of rows %d..%d" % (nres[0], r0, r1))
    return nres[0]


def score_tiles(net, ds, t_first, t_last, batch, out, route=None, lanes=None):
    """``net.forward_tiles`` over the windows [t_first, t_last) of the FlightlineConvolve ``ds`` in batches, asynchronously: on the
    split-operand route every batch raises its own overflow slot, the slots are read ONCE after the last batch and only the raised
    batches are scored again on the fp32 matrix cores (cnn_pred_pipeline.py:173-181).  Returns the number of batches re-scored."""
    W = ds.inshape[2]
    code = net._route_code(route)
    if not net.half and getattr(net, "c_driver", True) and t_first % W == 0 and t_last % W == 0 and t_last > t_first:
        return _score_rows_c(net, ds, t_first // W, t_last // W, batch, out, code, lanes)       # whole image rows: the C-side driver
    starts = list(range(int(t_first), int(t_last), int(batch)))
    slots = net.overflow_slots(len(starts)) if (code in (0, 3, 5) and not net.half) else None
    for i, t0 in enumerate(starts):
        net.forward_tiles(ds.x, W, t0, min(batch, t_last - t0), plane=ds.plane, out=out, route=code,
                          overflow=None if slots is None else slots[i:i + 1])
    redo = [] if slots is None else [i for i, v in enumerate(slots.cpu().tolist()) if v]
    for i in redo:
        t0 = starts[i]
        net.forward_tiles(ds.x, W, t0, min(batch, t_last - t0), plane=ds.plane, out=out, route=4)
    if redo:
        _overflow_warning("%d of %d batches" % (len(redo), len(starts)))
    return len(redo)


def predict_flightline(cmf2d, model="COVID_QC", weights=None, batch=32, gpus=None, rows=None, net=None, to_numpy=False,
                       precision="fp32", route=None, info=None, scales=None, lanes=None):
    """saliency[H, W] float32 = softmax(GoogLeNet(window))[:, 1] for the 256x256 window centred on every pixel,
    -9999 where ``cmf2d`` is -9999 (cnn_pred_pipeline.py:159-189).

    model   : a name of MODEL_NORM or a (mean, std) pair
    weights : a GoogLeNet ``state_dict`` (name -> tensor/ndarray)
    batch   : windows per launch set.  The map does not depend on it (bit-identical for any batch size and row range); the speed
              does: 20 k windows/s at the reference's default of 32, 100 k at 512, 103-108 k at 1024 on an MI355X (the CLI scores at
              least 1024 at a time)
    rows    : optional (r0, r1) image rows to score (multi-GPU row sharding); other rows are left at 0
    gpus    : device indices (the script's ``-g 0 1 2 3``, which wraps the model in ``DataParallel`` and re-scatters
              every batch, cnn_pred_pipeline.py:113-116).  Here every listed GPU gets its own copy of the weights and
              of the 69 MB padded plane and scores a contiguous block of image rows from its own host thread; the
              blocks are copied to the first device once.  A negative index (the reference's CPU run) is refused.
    route   : the convolutions' arithmetic -- ``"split"`` (default: fp32 operands as fp16 hi + lo halves on the fp16 matrix cores,
              the float32 tolerance class; through the C driver the trunk through inception3b is shared between the overlapping
              windows -- ``"split_conv3"`` / ``"split_unshared"``: less / none of that, the same bits), ``"winograd"`` (fp32
              matrix cores), ``"direct"``.  On the split routes the per-layer
              activation scales are calibrated on this plane and a batch whose activations leave float16's range is scored again
              on the fp32 matrix cores inside the call (with a warning): route and overflow handling are per call, so concurrent
              threads / streams / GPUs cannot disturb each other.
    scales  : split route only: the per-layer activation scales (``sf_cnn_num_scales()`` powers of two) instead of the
              calibration on this plane (a campaign that wants ONE set of scales for all its flightlines; the tests)
    lanes   : concurrent parts of the row range on one device (default 2: two host threads, two HIP streams, two workspaces -- a
              second batch in flight fills the tails of the first one's small launches, +5-7 %; 1: one stream).  The map does not depend
              on it unless a batch overflows float16: the re-scored batches are then cut at other windows
    info    : optional dict; receives ``rescued_batches``, ``shared_batches``, ``batches`` (launched by the C driver: the row range is
              scored as two concurrent halves on two streams when it is long enough), ``route`` and the ``scales`` used
    """
    torch = _torch()
    if gpus is not None and len(gpus) > 0:
        gpus = [int(g) for g in gpus]
        if any(g < 0 for g in gpus):
            raise _ffi.SrcfinderError("gpus=%r: srcfinder_amd has no CPU path (the reference's -g -1)" % (gpus,))
        if any(g >= torch.cuda.device_count() for g in gpus):
            raise _ffi.SrcfinderError("gpus=%r: only %d device(s) visible" % (gpus, torch.cuda.device_count()))
        if len(gpus) > 1:
            if net is not None or rows is not None:
                raise ValueError("gpus=[...] builds one network per device: do not pass net= or rows=")
            return _predict_multi_gpu(cmf2d, model, weights, batch, gpus, to_numpy, precision, route, info, scales, lanes)
        if net is None:
            if weights is None:
                raise ValueError("weights (a GoogLeNet state_dict) are required")
            net = GoogLeNetHIP(weights, device=torch.device("cuda", gpus[0]), precision=precision)
    if net is None:
        if weights is None:
            raise ValueError("weights (a GoogLeNet state_dict) are required")      # the script exits 1 (:93-95)
        net = GoogLeNetHIP(weights, precision=precision)
    ds = FlightlineConvolve(cmf2d, model, device=net.device)
    H, W = ds.inshape[1], ds.inshape[2]
    out = torch.zeros(H * W, dtype=torch.float32, device=net.device)
    r0, r1 = (0, H) if rows is None else rows
    code = net._route_code(route)
    rescued = 0
    if code in (0, 3, 5) and not net.half:
        # ONE set of scales for the call, whichever driver sequences the graph: the caller's, or sf_cnn_calibrate on this plane
        if scales is not None:
            if len(scales) != len(net.ascale):
                raise ValueError("scales must hold %d values" % len(net.ascale))
            net.ascale = [float(v) for v in scales]
        else:
            net.calibrate(ds, batch)
    if not net.half and getattr(net, "c_driver", True):
        # the C-side driver sequences the whole graph for the row range: one library call (shared trunk, per-batch overflow slots
        # and the fp32 re-scoring of raised batches included)
        rescued = _score_rows_c(net, ds, r0, r1, batch, out, code, lanes) if r1 > r0 else 0
    else:
        rescued = score_tiles(net, ds, r0 * W, r1 * W, batch, out, route=code)
    if info is not None:
        info.update(rescued_batches=rescued, route=code, scales=list(net.ascale) if code in (0, 3, 5) else None,
                    shared_batches=getattr(net, "last_shared_batches", 0) if code in (0, 5) else 0,
                    batches=getattr(net, "last_batches", None))
    out = out.view(H, W)
    return out.cpu().numpy() if to_numpy else out


def _predict_multi_gpu(cmf2d, model, weights, batch, gpus, to_numpy, precision, route=None, info=None, scales=None, lanes=None):
    """Row blocks of the saliency map on several GPUs of one process (one host thread per device)."""
    import threading
    torch = _torch()
    if weights is None:
        raise ValueError("weights (a GoogLeNet state_dict) are required")
    plane = cmf2d.detach().cpu().numpy() if torch.is_tensor(cmf2d) else np.asarray(cmf2d, dtype=np.float32)
    H = plane.shape[0]
    n = len(gpus)
    parts, errs, infos = [None] * n, [None] * n, [dict() for _ in range(n)]
    # the library's tuning knobs are per calling thread: hand the caller's CNN knobs to the workers (the convolutions' route is
    # resolved HERE, once, and passed to every worker as an argument)
    route = GoogLeNetHIP._route_code(_RouteOnly(precision), route)
    import ctypes
    L = _ffi.lib()
    knobs = {}
    for key in (16, 17, 18):
        v = ctypes.c_int(0)
        _ffi.check(L.sf_debug_get(key, ctypes.byref(v)), "sf_debug_get")
        knobs[key] = v.value

    def work(i):
        try:
            for key, v in knobs.items():
                L.sf_debug_set(key, v)
            dev = torch.device("cuda", gpus[i])
            with torch.cuda.device(dev):
                net = GoogLeNetHIP(weights, device=dev, precision=precision)
                r0, r1 = i * H // n, (i + 1) * H // n
                sal = predict_flightline(plane, model, net=net, batch=batch, rows=(r0, r1), route=route, info=infos[i], scales=scales,
                                         lanes=lanes)
                parts[i] = sal[r0:r1].to(torch.device("cuda", gpus[0]), non_blocking=False)
        except Exception as e:                                  # surfaced in the caller's thread
            errs[i] = e

    threads = [threading.Thread(target=work, args=(i,)) for i in range(n)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errs:
        if e is not None:
            raise e
    if info is not None:
        info.update(rescued_batches=sum(d.get("rescued_batches", 0) for d in infos), route=route,
                    per_block_rescued=[d.get("rescued_batches", 0) for d in infos], scales=infos[0].get("scales"))
    out = torch.cat(parts, 0)
    return out.cpu().numpy() if to_numpy else out


class _RouteOnly(object):
    """(just enough of a network for GoogLeNetHIP._route_code before the per-device networks exist)"""

    def __init__(self, precision):
        self.route, self.half = None, precision == "fp16"


def fcn_predict_flightline(cmf2d, model="COVID_QC", weights=None, scale=32, batch=8, net=None, to_numpy=False,
                           precision="fp32", shifts=None, route=None):
    """The reference's FCN shift-and-stitch fast mode (cnn/fcn_pred_pipeline.py:32-95, :157-160, :225-249): the trunk
    runs fully convolutionally over the whole flightline once per (top, left) shift, the 1x1 head gives one
    probability per ``scale`` x ``scale`` cell and the ``scale**2`` maps are interlaced.  An approximation of
    :func:`predict_flightline` by construction (the reference's own), ~44x cheaper.

    shifts : optional (s0, s1) range of shift indices (multi-GPU: shifts are independent; other pixels stay 0)
    """
    torch = _torch()
    if scale != 32:
        raise ValueError("the GoogLeNet trunk has a total stride of 32")
    if net is None:
        if weights is None:
            raise ValueError("weights (a GoogLeNet state_dict) are required")      # the script exits 1 (:139-141)
        net = GoogLeNetHIP(weights, precision=precision)
    vmin, vmax, mean, std = _parse_transform(model)
    plane = cmf2d if torch.is_tensor(cmf2d) else torch.from_numpy(np.array(cmf2d, dtype=np.float32, order="C", copy=True))
    plane = plane.to(device=net.device, dtype=torch.float32).contiguous()
    H, W = plane.shape
    Hc, Wc = H + (scale - H % scale) + scale, W + (scale - W % scale) + scale       # div_pad + shift pad (:44-65)
    out = torch.zeros((H, W), dtype=torch.float32, device=net.device)
    s0, s1 = (0, scale * scale) if shifts is None else shifts
    L = _ffi.lib()
    code = net._route_code(route)
    with torch.cuda.device(net.device):
        st = _ffi.stream_ptr()
        if code in (0, 3, 5) and not net.half:
            # the split route's activation scales: calibrated on this plane's 256 x 256 windows (the same trunk, the same statistics)
            net.calibrate(FlightlineConvolve(plane, (mean, std) if (vmin, vmax) == (0.0, 4000.0) else
                                             Compose([ClampCH4(int(vmin), int(vmax)), Normalize([mean], [std])]), device=net.device))
        canvas = torch.empty((batch, Hc, Wc), dtype=torch.float32, device=net.device)
        starts = list(range(s0, s1, batch))
        slots = net.overflow_slots(len(starts)) if (code in (0, 3, 5) and not net.half) else None

        def run(i, rt):
            a = starts[i]
            n = min(batch, s1 - a)
            _ffi.check(L.sf_cnn_fcn_prepare(_ffi.ptr(plane), H, W, float(vmin), float(vmax), float(mean), float(std), scale,
                                            a, n, Hc, Wc, _ffi.ptr(canvas), st), "sf_cnn_fcn_prepare")
            pred = net.forward_fcn(canvas[:n], route=rt, overflow=None if (slots is None or rt not in (0, 3, 5)) else slots[i:i + 1])
            _ffi.check(L.sf_cnn_fcn_stitch(_ffi.ptr(pred), n, a, scale, pred.shape[1], pred.shape[2], _ffi.ptr(plane), H, W,
                                           NODATA, _ffi.ptr(out), st), "sf_cnn_fcn_stitch")

        for i in range(len(starts)):
            run(i, code)
        redo = [] if slots is None else [i for i, v in enumerate(slots.cpu().tolist()) if v]
        for i in redo:                               # the split-operand kernels' float16 range (see predict_flightline)
            run(i, 4)
        if redo:
            _overflow_warning("%d of %d shift batches" % (len(redo), len(starts)))
    return out.cpu().numpy() if to_numpy else out
