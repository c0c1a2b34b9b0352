"""`build_encoder(cfg)` / `ResUNet`: the per-frame image encoder with the reference's interface and
state_dict keys (libs/encoders/UNet.py:133-242), so `load_state_dict(strict=True)` of a reference checkpoint works.

Per frame, not per ray: V=3 source images [V,3,H,W] -> feature maps [V,32,H/4,W/4], ~120 GFLOP of convolutions at 512x512.
Everything runs in hand-written HIP kernels on channels-last activations (csrc/gpnerf_conv.hip): the convolutions as implicit
GEMMs on the matrix cores, reflection padding as index arithmetic, InstanceNorm fused with the residual add and activation
behind it, the two bilinear upsamplings as one launch each -- ONE launch chain (55 launches, replayed as a HIP graph) in two
arithmetic forms (`ResUNet.precision`):
  "fp32"  (default) fp32 operands on v_mfma_f32_32x32x2_f32: every dot product an fp32 FMA chain like the reference's own.  No
          operand range.  The form whose end-to-end chain (encoder -> head -> renderer) stays within 1e-4 of the reference on
          every map at the config-5 size (tests/test_gpu_renderer.py);
  "split" fp32 operands split into f16 hi + lo on the f16 MFMA (three per k-step, f32 accumulation: ~23 bits per operand): the
          fast mode (`encoder.file hip_encoder_fast`), the same 3e-5 on the feature maps but rounding that is uncorrelated with the
          reference's, which the head amplifies to 2.5e-4 on depth at that size; operand range below.  The result leaves with channels-last strides (logical NCHW, physical NHWC), which
is the layout the render kernel gathers from, so `Frame` takes it as is, without a re-layout launch.  The nn.Conv2d /
nn.InstanceNorm2d sub-modules are parameter containers under the reference's names; there is no torch-operator path (the
one the tests check against is oracle/producers_ref.py `encoder`).

Operand range (precision "split" only).  The split-f16 convolutions hold |w| < 16 and |x| < 4 095.  No checkpoint is refused for that: an operand beyond
the range splits into f16 infinities, the outputs it meets are NaN, and the convolutions raise a flag word when they see a
non-finite sum in their InstanceNorm table (`_Run.flag`, include/gpnerf_hip.h `range_flag`) -- a DATA-dependent guard that costs
the kernels nothing.  A flagged pass is discarded and the frame encoded again by `forward_exact` (the "fp32" form).  `check_operand_range` is the fast accept in front of it: when
the parameters alone bound every activation (sqrt(h w) |gamma| + |beta| ...) the flag cannot be raised and is never looked at.

Network (UNet.py:154-234): 7x7/2 stem -> three residual stages of [3,4,6] two-conv units at 64/128/256 channels, every stage
entered with stride 2 (there is no max-pool), all 3x3 / 7x7 convolutions reflect-padded, every normalisation an affine
InstanceNorm without running statistics -> two (bilinear x2, align_corners) + conv + concat-skip decoder steps with
InstanceNorm + ELU -> 1x1 output convolution.
"""
import threading

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L

# UNet.py:151 fixes the stage depths to [3,4,6] whatever the name says.  The names resnet50/101/152 only widen the decoder's
# expected skip channels (UNet.py:142-145) while the stages stay 64/128/256 wide, so the reference's own forward fails
# on them; they are rejected here at construction.
_NAMES = ("resnet18", "resnet34")


def _inorm(ch):
    return nn.InstanceNorm2d(ch, track_running_stats=False, affine=True)


def _mk_conv(cin, cout, k, stride=1, bias=False):
    return nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, bias=bias, padding_mode="reflect")


def _require_gpu_inference(x, training):
    """The encoder is built for GPU inference; there is no CPU / training path in the product -- the torch-operator
    formulation the tests check against is oracle/producers_ref.py `encoder`."""
    if not x.is_cuda or training:
        raise L.GpnerfError("the HIP image encoder runs on GPU tensors in eval mode only (no CPU / training fallback)")


def _st(x):
    return torch.cuda.current_stream(x.device).cuda_stream


def _nhwc(x):
    """logical [N,C,H,W] with channels-last strides = physical [N,H,W,C] (a copy only if it is not laid out so already)"""
    return x.contiguous(memory_format=torch.channels_last)


def _exact():
    """the arithmetic form of the convolutions called from here (set by `_form`): True = fp32 operands (exact = 1)"""
    return bool(getattr(_current, "exact", True))


class _form:
    """with _form(exact): every convolution called inside runs in that arithmetic form (include/gpnerf_hip.h `exact`)"""

    def __init__(self, exact):
        self.exact = bool(exact)

    def __enter__(self):
        self.prev = getattr(_current, "exact", True)
        _current.exact = self.exact

    def __exit__(self, *exc):
        _current.exact = self.prev


def _packed_weight(conv):
    """gpnerf_conv_pack_weight image of a conv's weight in the current form (fp32, or f16 hi/lo; MFMA A-operand order), re-packed when the parameter changes.
    The image lives ON the module (not in a table keyed by id(): ids and device pointers are re-used once a model is freed, and
    a second checkpoint would have found the first one's image), so it is freed with the module; the fp32 source the pack kernel
    reads needs no keeping -- stream order protects it."""
    w = conv.weight
    exact = _exact()
    key = (str(w.device), w.data_ptr(), w._version)
    slot = "_gpnerf_packed_f32" if exact else "_gpnerf_packed"
    hit = conv.__dict__.get(slot)
    if hit is None or hit[0] != key:
        lib = L.lib()
        cout, cin, ks, _ = w.shape
        buf = torch.empty((int(lib.gpnerf_conv_packed_bytes(cout, cin, ks)),), dtype=torch.uint8, device=w.device)
        src = w.detach().float().contiguous()
        L.check(lib.gpnerf_conv_pack_weight(src.data_ptr(), cout, cin, ks, int(exact), buf.data_ptr(), _st(w)), "gpnerf_conv_pack_weight")
        hit = (key, buf)
        conv.__dict__[slot] = hit
    return hit[1]


def _conv(conv, x, stats=False):
    """nn.Conv2d(..., padding=k//2, padding_mode='reflect') on a channels-last tensor -> channels-last tensor (gpnerf_conv2d_nhwc).
    stats=True: also returns the per-tile (sum, sum of squares, M2) of every output channel (what `_conv_norm`'s table is merged from)."""
    if not x.is_cuda:
        raise L.GpnerfError("the HIP image encoder runs on GPU tensors only (no CPU fallback)")
    x = _nhwc(x)
    n, cin, h, w = x.shape
    cout, _, ks, _ = conv.weight.shape
    stride = conv.stride[0]
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    out = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    lib = L.lib()
    ts = torch.empty((n, int(lib.gpnerf_conv_out_tiles(h, w, cin, ks, stride)), cout, 3), device=x.device, dtype=torch.float32) if stats else None
    L.check(lib.gpnerf_conv2d_nhwc(x.data_ptr(), n, h, w, cin, _packed_weight(conv).data_ptr(),
                                   bias.data_ptr() if bias is not None else None, cout, ks, stride, out.data_ptr(),
                                   ts.data_ptr() if ts is not None else None, None if _exact() else _run_of(x).flag_ptr, int(_exact()), _st(x)),
            "gpnerf_conv2d_nhwc")
    return (out, ts) if stats else out


def _conv_exact(conv, x):
    """The same nn.Conv2d through gpnerf_conv2d_nhwc_exact: fp32 operands straight from the PyTorch weight, one scalar load per
    operand, any odd kernel size / stride / channel count.  NOT on the encoder's path (its "fp32" form is the fused kernels with
    exact = 1): the independent restatement tests hold them against."""
    if not x.is_cuda:
        raise L.GpnerfError("the HIP image encoder runs on GPU tensors only (no CPU fallback)")
    x = _nhwc(x.float())
    n, cin, h, w = x.shape
    cout, _, ks, _ = conv.weight.shape
    stride = conv.stride[0]
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    out = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    wt = conv.weight.detach().float().contiguous()
    L.check(L.lib().gpnerf_conv2d_nhwc_exact(x.data_ptr(), n, h, w, cin, wt.data_ptr(), bias.data_ptr() if bias is not None else None,
                                             cout, ks, stride, out.data_ptr(), _st(x)), "gpnerf_conv2d_nhwc_exact")
    return out


def _norm_act(norm, x, act, residual=None):
    """act(InstanceNorm(x) [+ residual]) on channels-last tensors; act: 0 none, 1 ReLU, 2 ELU (gpnerf_instance_norm_act_nhwc: the
    statistics from a double-precision pass over x).  The stand-alone operator: the encoder's path takes its tables from the
    convolutions' own epilogues (`_conv_norm`), which the tests hold against this."""
    lib = L.lib()
    x = _nhwc(x)
    n, c, h, w = x.shape
    out = torch.empty_like(x, memory_format=torch.channels_last)
    res = _nhwc(residual) if residual is not None else None
    scratch = torch.empty((int(lib.gpnerf_instance_norm_nhwc_scratch_bytes(n, h * w, c)),), dtype=torch.uint8, device=x.device)
    L.check(lib.gpnerf_instance_norm_act_nhwc(x.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(),
                                              res.data_ptr() if res is not None else None, n, h * w, c, float(norm.eps), act,
                                              out.data_ptr(), scratch.data_ptr(), _st(x)), "gpnerf_instance_norm_act_nhwc")
    return out


class _Run:
    """The words one pass of the encoder's fused convolutions shares on the device: the ticket words in which the workgroups of an
    (image, channel group) count themselves (zero before a launch, left zero by it) and the range flag (module docstring).  Two
    passes that may overlap in time must not share them -- counts would mix -- so there is one _Run per (device, stream) for eager
    calls and one per captured graph (a graph bakes the pointers in and may be replayed on any stream).
    The flag lives in pinned host memory: the device writes it (only ever a 1, and only when something is out of range), the
    host reads it after a stream synchronisation without a copy."""

    def __init__(self, dev):
        self.tickets = torch.zeros((4096,), dtype=torch.int32, device=dev)
        self.flag = torch.zeros((1,), dtype=torch.int32).pin_memory()
        self.flag_ptr = self.flag.data_ptr()

    def raised(self):
        """True when a convolution of the passes since the last clear() saw an operand beyond the f16 range.  The caller has
        synchronised with the stream the pass ran on."""
        return bool(int(self.flag[0]) != 0)

    def clear(self):
        self.flag.zero_()


_runs = {}                       # (device, stream handle) -> _Run of the eager calls on that stream
_current = threading.local()     # .run: the _Run a capture / replay pins for the calls it makes


def _run_of(x):
    run = getattr(_current, "run", None)
    if run is not None:
        return run
    key = (str(x.device), _st(x))
    run = _runs.get(key)
    if run is None:
        if len(_runs) > 64:
            # drop the OLDEST entry only (dicts keep insertion order): clearing the table in the middle of a pass handed the pass's
            # later convolutions fresh words -- forward() then read a flag the raise never reached
            _runs.pop(next(iter(_runs)))
        run = _runs[key] = _Run(x.device)
    else:
        _runs[key] = _runs.pop(key)       # most recently used last
    return run


class _pinned_run:
    """with _pinned_run(run): every fused convolution called inside uses `run`'s words"""

    def __init__(self, run):
        self.run = run

    def __enter__(self):
        self.prev = getattr(_current, "run", None)
        _current.run = self.run
        return self.run

    def __exit__(self, *exc):
        _current.run = self.prev


def _ticket_words(dev):
    """the ticket words of the current stream's eager calls on `dev` (tests look at them: a launch must leave them zero)"""
    return _run_of(torch.empty((0,), device=dev)).tickets


def _conv_norm(conv, norm, x, in_tab=None, in_act=0):
    """conv(x) together with the table (mean, gamma * rstd, beta per channel) of the InstanceNorm `norm` behind it, computed by the
    convolution's last workgroup (gpnerf_conv2d_norm_nhwc): no reduction launch.  in_tab: the table of an InstanceNorm in FRONT of
    the convolution -- it then reads act((x - mean) * scale + beta) while staging x (in_act 1 = ReLU), and that tensor is never
    written.  Returns (y, table [N,3,cout])."""
    if not x.is_cuda:
        raise L.GpnerfError("the HIP image encoder runs on GPU tensors only (no CPU fallback)")
    x = _nhwc(x)
    n, cin, h, w = x.shape
    cout, _, ks, _ = conv.weight.shape
    stride = conv.stride[0]
    pad = ks // 2
    ho, wo = (h + 2 * pad - ks) // stride + 1, (w + 2 * pad - ks) // stride + 1
    lib = L.lib()
    out = torch.empty((n, cout, ho, wo), device=x.device, dtype=torch.float32, memory_format=torch.channels_last)
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    ts = torch.empty((n, int(lib.gpnerf_conv_out_tiles(h, w, cin, ks, stride)), cout, 3), device=x.device, dtype=torch.float32)
    tab = torch.empty((n, 3, cout), device=x.device, dtype=torch.float32)
    run = _run_of(x)
    tick = run.tickets
    if n * ((cout + 31) // 32) > tick.numel():
        raise L.GpnerfError("too many (image, channel group) pairs for the ticket words")
    L.check(lib.gpnerf_conv2d_norm_nhwc(x.data_ptr(), n, h, w, cin, in_tab.data_ptr() if in_tab is not None else None, int(in_act),
                                        _packed_weight(conv).data_ptr(), bias.data_ptr() if bias is not None else None, cout, ks, stride,
                                        out.data_ptr(), ts.data_ptr(), norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps),
                                        tab.data_ptr(), tick.data_ptr(), None if _exact() else run.flag_ptr, int(_exact()), _st(x)),
            "gpnerf_conv2d_norm_nhwc")
    return out, tab


def _conv_norm_cat(conv, norm, xa, xb):
    """_conv_norm on torch.cat([xa, xb], 1) without writing the concatenation (gpnerf_conv2d_norm_cat_nhwc: a 3x3 stride-1 convolution
    whose channel blocks come from one tensor or the other).  Returns (y, table)."""
    xa, xb = _nhwc(xa), _nhwc(xb)
    n, ca, h, w = xa.shape
    cb = xb.shape[1]
    cout = conv.weight.shape[0]
    lib = L.lib()
    out = torch.empty((n, cout, h, w), device=xa.device, dtype=torch.float32, memory_format=torch.channels_last)
    bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
    ts = torch.empty((n, int(lib.gpnerf_conv_out_tiles(h, w, ca + cb, 3, 1)), cout, 3), device=xa.device, dtype=torch.float32)
    tab = torch.empty((n, 3, cout), device=xa.device, dtype=torch.float32)
    run = _run_of(xa)
    tick = run.tickets
    if n * ((cout + 31) // 32) > tick.numel():
        raise L.GpnerfError("too many (image, channel group) pairs for the ticket words")
    L.check(lib.gpnerf_conv2d_norm_cat_nhwc(xa.data_ptr(), ca, xb.data_ptr(), cb, n, h, w, _packed_weight(conv).data_ptr(),
                                            bias.data_ptr() if bias is not None else None, cout, out.data_ptr(), ts.data_ptr(),
                                            norm.weight.data_ptr(), norm.bias.data_ptr(), float(norm.eps), tab.data_ptr(), tick.data_ptr(),
                                            None if _exact() else run.flag_ptr, int(_exact()), _st(xa)), "gpnerf_conv2d_norm_cat_nhwc")
    return out, tab


def _fusable_input_norm(conv):
    """convolutions that can apply an InstanceNorm (+ ReLU) to their input while they read it: 3x3 (staging) and 1x1 (splitting)
    on whole 16-channel blocks"""
    return conv.kernel_size in ((3, 3), (1, 1)) and conv.in_channels % 16 == 0


def _apply(x, tab, act, residual=None, res_tab=None):
    """act((x - mean) * scale + beta [+ residual]) from a `_conv_norm` table; res_tab: the residual's own table (projected shortcut)"""
    n, c, h, w = x.shape
    out = torch.empty_like(x, memory_format=torch.channels_last)
    res = _nhwc(residual) if residual is not None else None
    L.check(L.lib().gpnerf_norm_apply_nhwc(x.data_ptr(), tab.data_ptr(), res.data_ptr() if res is not None else None,
                                           res_tab.data_ptr() if res_tab is not None else None, n, h * w, c, act, out.data_ptr(), _st(x)),
            "gpnerf_norm_apply_nhwc")
    return out


def _upsample2x(x):
    x = _nhwc(x)
    n, c, h, w = x.shape
    out = torch.empty((n, c, 2 * h, 2 * w), device=x.device, dtype=x.dtype, memory_format=torch.channels_last)
    L.check(L.lib().gpnerf_upsample2x_nhwc(x.data_ptr(), n, h, w, c, out.data_ptr(), _st(x)), "gpnerf_upsample2x_nhwc")
    return out


class ResidualUnit(nn.Module):
    """Two 3x3 convolutions + identity / projected shortcut (UNet.py:17-53).  Attribute names are the checkpoint's."""

    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1, self.bn1 = _mk_conv(cin, cout, 3, stride), _inorm(cout)
        self.conv2, self.bn2 = _mk_conv(cout, cout, 3), _inorm(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(_mk_conv(cin, cout, 1, stride), _inorm(cout))

    def forward(self, x, x_tab=None):
        # conv1 -> [bn1 + ReLU applied while conv2 stages its input] -> conv2 -> one pass: relu(bn2(.) + shortcut), the projected
        # shortcut's own InstanceNorm applied on the fly: 4 launches per unit (5 with a projection) instead of 7 (10).
        # x_tab: x is a convolution's raw output whose InstanceNorm + ReLU is still pending (its table).  A unit with a projected
        # shortcut reads x twice, through two convolutions that can both apply the table as they read -- x is then never written
        # normalised; any other unit needs the tensor itself (the identity shortcut adds it).
        act0 = 0
        if x_tab is not None:
            if self.downsample is not None and _fusable_input_norm(self.conv1) and _fusable_input_norm(self.downsample[0]):
                act0 = 1
            else:
                x, x_tab = _apply(x, x_tab, 1), None
        y1, t1 = _conv_norm(self.conv1, self.bn1, x, in_tab=x_tab, in_act=act0)
        if _fusable_input_norm(self.conv2):
            y2, t2 = _conv_norm(self.conv2, self.bn2, y1, in_tab=t1, in_act=1)
        else:
            y2, t2 = _conv_norm(self.conv2, self.bn2, _apply(y1, t1, 1))
        if self.downsample is None:
            return _apply(y2, t2, 1, residual=x)
        d, td = _conv_norm(self.downsample[0], self.downsample[1], x, in_tab=x_tab, in_act=act0)
        return _apply(y2, t2, 1, residual=d, res_tab=td)


class ConvNormELU(nn.Module):
    """conv (with bias, reflect pad) -> InstanceNorm -> ELU (UNet.py:107-120); sub-modules `conv`, `bn`."""

    def __init__(self, cin, cout, k):
        super().__init__()
        self.conv, self.bn = _mk_conv(cin, cout, k, bias=True), _inorm(cout)

    def forward(self, x):
        if isinstance(x, tuple):                     # (a, b): the concatenation [a, b] on channels, read in place
            return _apply(*_conv_norm_cat(self.conv, self.bn, *x), 2)
        return _apply(*_conv_norm(self.conv, self.bn, x), 2)


class UpsampleConv(nn.Module):
    """bilinear x`scale` (align_corners) then ConvNormELU (UNet.py:123-131); the conv block is the sub-module `conv`."""

    def __init__(self, cin, cout, k, scale):
        super().__init__()
        self.scale = scale
        self.conv = ConvNormELU(cin, cout, k)

    def forward(self, x):
        if self.scale != 2 or x.dtype != torch.float32:
            raise L.GpnerfError("the upsampling kernel is built for the reference's x2 bilinear steps on fp32 (UNet.py:185-188)")
        return self.conv(_upsample2x(x))


def _stage(cin, cout, n):
    return nn.Sequential(*[ResidualUnit(cin if i == 0 else cout, cout, 2 if i == 0 else 1) for i in range(n)])


def _concat_skip(skip, up):
    """Zero-pad the skip tensor (centred, extra on the far side) to the upsampled size, then [up, skip] on channels
    (UNet.py:199-211)."""
    dy, dx = up.shape[2] - skip.shape[2], up.shape[3] - skip.shape[3]
    if dy or dx:
        skip = F.pad(skip, (dx // 2, dx - dx // 2, dy // 2, dy - dy // 2))
    if up.is_cuda and up.shape[1] % 16 == 0 and skip.shape[1] % 16 == 0 and up.dtype == torch.float32 and skip.dtype == torch.float32:
        return up, skip                              # the convolution behind reads the two tensors in place (ConvNormELU.forward)
    return torch.cat([up, skip], dim=1)


class ResUNet(nn.Module):
    def __init__(self, encoder="resnet34", out_ch=32, norm_layer=None):
        super().__init__()
        if encoder not in _NAMES:
            raise ValueError(f"encoder type {encoder!r}: only {_NAMES} give a runnable network")
        if norm_layer is not None:
            raise ValueError("the reference always runs InstanceNorm here (UNet.py:152-153); other norms are not built")
        n1, n2, n3 = 3, 4, 6
        skip1, skip2, deep = 64, 128, 256
        self.conv1, self.bn1 = _mk_conv(3, 64, 7, stride=2), _inorm(64)
        self.layer1, self.layer2, self.layer3 = _stage(64, 64, n1), _stage(64, 128, n2), _stage(128, 256, n3)
        self.upconv3 = UpsampleConv(deep, 128, 3, 2)
        self.iconv3 = ConvNormELU(skip2 + 128, 128, 3)
        self.upconv2 = UpsampleConv(128, 64, 3, 2)
        self.iconv2 = ConvNormELU(skip1 + 64, out_ch, 3)
        self.out_conv = nn.Conv2d(out_ch, out_ch, 1, 1)

    # ---- arithmetic form (module docstring): "fp32" (default) or "split"; `net.precision = "split"` per module, or the plugin
    # `encoder.file hip_encoder_fast`.  Not a parameter: state_dict keys are the reference's 108.
    @property
    def precision(self):
        return self.__dict__.get("_gpnerf_precision", "fp32")

    @precision.setter
    def precision(self, v):
        if v not in ("fp32", "split"):
            raise ValueError(f"precision {v!r}: 'fp32' (default, the reference's arithmetic) or 'split' (f16 hi/lo, fast)")
        self.__dict__["_gpnerf_precision"] = v

    # ---- operand range of the split-f16 convolutions (csrc/gpnerf_conv.hip: weights staged as 2^12 w, activations as 2^4 x) ----
    W_LIMIT, X_LIMIT = 15.99, 4094.0
    exact_frames = 0             # frames this module has encoded through forward_exact (instances count in their __dict__)

    def check_operand_range(self, H, W, params_key=None):
        """Which guard a frame of H x W images needs, from the PARAMETERS alone (cached per (H, W, parameter versions): one
        device-to-host copy per parameter change).  Never refuses finite parameters:
          "static"   every convolution BEHIND the stem has its operands inside the split-f16 range whatever the images: weights
                     |w| < 16 directly, activations through the bound InstanceNorm gives them -- a normalised value is at most
                     sqrt(h w - 1) in magnitude, so a norm's output is bounded by sqrt(h w) max|gamma| + max|beta|, a residual
                     unit's by that plus its shortcut's bound, and upsampling / concatenation / ReLU / ELU do not raise a bound.
                     The stem reads the IMAGE: a pixel of magnitude >= 4094 or a non-finite one still leaves the range, so the
                     flag is read for this class too (round 5) -- deferred to the caller's own synchronisation on the
                     render path, where it costs nothing (the reference's initialisation; any InstanceNorm scale below ~10 at 512^2);
          "dynamic"  the bound does not hold (it grows with the image size and is attained by a one-hot image only): the split form
                     runs and its range flag says whether THIS frame left the range -- if so the frame is encoded again by
                     forward_exact;
          "exact"    a weight is 16 or more in magnitude (or a parameter is not finite): every frame goes through forward_exact.
        `self.range_report` names the first layer that decided a "dynamic" / "exact" answer."""
        # params_key: the (storage, version) tuple of the parameters when the caller has just computed it (forward_graphed: walking
        # the 108 parameters costs the host ~0.1 ms, and this runs before the frame's first launch with the device idle)
        if self.precision == "fp32":
            return "exact"                 # (the default form has no operand range at all: nothing to classify)
        key = (int(H), int(W), params_key if params_key is not None else _graph_key(self, None)[3])
        hit = self.__dict__.get("_gpnerf_range_class")
        if hit is not None and hit[0] == key:
            return hit[1]
        names, tensors = zip(*[(n, p) for n, p in self.named_parameters()])
        mx = dict(zip(names, torch.stack([t.detach().abs().max().float() for t in tensors]).cpu().tolist()))
        verdict, report = "static", None

        def settle(v, why):
            nonlocal verdict, report
            if report is None or (v == "exact" and verdict != "exact"):
                report = why
            if v == "exact" or verdict == "static":
                verdict = v

        for n, v in mx.items():
            if n.endswith("weight") and n.rsplit(".", 1)[0] in self._conv_names() and not v < self.W_LIMIT:
                settle("exact", f"|{n}| reaches {v:.3g}; the split-f16 convolution holds weights below 16")
            elif not v < float("inf"):
                settle("exact", f"{n} is not finite")

        def norm_bound(prefix, hw):
            return (hw ** 0.5) * mx[prefix + ".weight"] + mx[prefix + ".bias"]

        def need(name, bound):
            if not bound < self.X_LIMIT:
                settle("dynamic", f"the input of {name} can reach {bound:.4g} on a one-hot image (InstanceNorm scales x sqrt(h w)); the "
                                  f"split-f16 convolution holds activations below {self.X_LIMIT:.0f}")

        half = lambda n: (n - 1) // 2 + 1
        h, w = half(int(H)), half(int(W))
        b = norm_bound("bn1", h * w)
        sizes, bounds = {}, {}
        for stage, units in (("layer1", 3), ("layer2", 4), ("layer3", 6)):
            for u in range(units):
                p = f"{stage}.{u}"
                if u == 0:
                    need(p + ".downsample.0", b)
                    need(p + ".conv1", b)
                    h, w = half(h), half(w)
                    idn = norm_bound(p + ".downsample.1", h * w)
                else:
                    need(p + ".conv1", b)
                    idn = b
                need(p + ".conv2", norm_bound(p + ".bn1", h * w))
                b = norm_bound(p + ".bn2", h * w) + idn
            sizes[stage], bounds[stage] = (h, w), b
        need("upconv3.conv.conv", bounds["layer3"])
        h, w = 2 * sizes["layer3"][0], 2 * sizes["layer3"][1]
        b = max(norm_bound("upconv3.conv.bn", h * w), bounds["layer2"])
        need("iconv3.conv", b)
        b = norm_bound("iconv3.bn", h * w)
        need("upconv2.conv.conv", b)
        h, w = 2 * h, 2 * w
        b = max(norm_bound("upconv2.conv.bn", h * w), bounds["layer1"])
        need("iconv2.conv", b)
        need("out_conv", norm_bound("iconv2.bn", h * w))
        self.__dict__["_gpnerf_range_class"] = (key, verdict)
        self.__dict__["range_report"] = report
        return verdict

    def _conv_names(self):
        names = self.__dict__.get("_gpnerf_conv_names")
        if names is None:
            names = self.__dict__["_gpnerf_conv_names"] = {n for n, m in self.named_modules() if isinstance(m, nn.Conv2d)}
        return names

    def out_shape(self, H, W):
        """(C, h, w) of forward()'s result for [.,3,H,W] images: the stem and the three stages each halve with ceil (k = 7 / 3,
        pad k // 2, stride 2), the decoder doubles twice and pads the skips up to that size (UNet.py:199-211)."""
        h, w = int(H), int(W)
        for _ in range(4):
            h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        return self.out_conv.out_channels, 4 * h, 4 * w

    def forward(self, x):
        """x [V,3,H,W] -> [V,out_ch,H/4,W/4].  On the GPU the result leaves with channels-last strides, which `Frame`
        recognises (no re-layout launch).  precision "fp32": the launch chain in the exact form, nothing to check, no
        synchronisation.  precision "split": the split-f16 form; a frame that leaves its operand range (check_operand_range: the
        host then waits for the stream once) is encoded again by forward_exact.  `self.exact_frames` counts those."""
        _require_gpu_inference(x, self.training)
        cls = self.check_operand_range(x.shape[-2], x.shape[-1])
        if cls == "exact":
            with _form(True):
                return self.forward_fast(x)
        run = _run_of(x)
        # the flag word starts every checked pass at zero: a pass that was aborted between launch and check, or an earlier pass on
        # out-of-range data whose flag nobody looked at, must not send THIS frame to the exact form.  The host owns the
        # word here: the previous pass on this stream was synchronised with before its flag was read, or never read at all.
        torch.cuda.current_stream(x.device).synchronize()
        run.clear()
        with _pinned_run(run), _form(False):             # the whole pass on these words, whatever happens to the table meanwhile
            y = self.forward_fast(x)
        # "static" parameters bound every operand for images in the documented range only: an out-of-range or non-finite PIXEL still
        # overflows the stem's f16 split, and ReLU turns the NaNs into silent zeros -- so the flag is read for both classes
        torch.cuda.current_stream(x.device).synchronize()
        if run.raised():
            run.clear()
            return self.forward_exact(x)
        return y

    def forward_fast(self, x):
        """The launch chain alone, in the arithmetic form the caller set (`_form`; fp32 by default) -- what a HIP graph captures.
        In the split form the caller looks at the range flag."""
        y0, t0 = _conv_norm(self.conv1, self.bn1, x.float())      # bn1 + ReLU: applied by the two convolutions that read y0
        x1 = self.layer1[0](y0, t0)
        for unit in list(self.layer1)[1:]:
            x1 = unit(x1)
        x2 = self.layer2(x1)
        x3 = self.layer3(x2)
        x = self.iconv3(_concat_skip(x2, self.upconv3(x3)))
        x = self.iconv2(_concat_skip(x1, self.upconv2(x)))
        return _conv(self.out_conv, x)

    def forward_exact(self, x):
        """The network (UNet.py:154-234) in the fp32 form, eagerly: what a split-precision frame falls back to when it raised the
        range flag (`exact_frames` counts those calls).  The same launch chain as every "fp32" frame."""
        _require_gpu_inference(x, self.training)
        self.__dict__["exact_frames"] = self.exact_frames + 1
        with _form(True):
            return self.forward_fast(x)


class _EncoderGraph:
    """One HIP graph of ResUNet.forward_fast for one (input shape, device, parameter versions, arithmetic form): the 55 launches of
    a frame's encoding replayed with ONE host call.  The kernels themselves take as long as before (a replay is as GPU-bound as
    the eager launches) -- what the graph removes is the ~1.4 ms the HOST needs to enqueue them one by one through Python, during
    which it cannot prepare the rest of the frame (tools/probes/render_phases.py)."""

    def __init__(self, net, x, exact):
        dev = x.device
        self.exact = bool(exact)
        self.run = _Run(dev)                          # the graph's own ticket words and range flag (baked into its nodes)
        self.static_in = torch.empty(tuple(x.shape), device=dev, dtype=torch.float32)
        self.static_in.copy_(x)
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(cur)
        with _pinned_run(self.run), _form(self.exact):
            with torch.cuda.stream(side):             # eager warm-up: packs the weights, sets the kernels' LDS attributes
                net.forward_fast(self.static_in)
                net.forward_fast(self.static_in)
            cur.wait_stream(side)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.static_out = net.forward_fast(self.static_in)
        torch.cuda.current_stream(dev).synchronize()
        self.run.clear()                              # (the warm-up ran on this frame's data; the first replay decides for itself)

    def __call__(self, x):
        self.static_in.copy_(x)
        self.graph.replay()
        return self.static_out.clone()                # the caller owns its result (the next replay overwrites static_out)


def _graph_key(net, x):
    # the parameter OBJECTS are looked up once (walking the module tree costs ~0.1 ms per call, and this runs before the frame's
    # first launch, with the device idle); their storage and version are what is compared per call
    plist = net.__dict__.get("_gpnerf_params")
    if plist is None:
        plist = net.__dict__["_gpnerf_params"] = list(net.parameters())
    pk = tuple([(p.data_ptr(), p._version) for p in plist])
    return (tuple(x.shape), x.device.index, x.dtype, pk) if x is not None else (None, None, None, pk)


def forward_graphed(net, x, defer_range_check=False):
    """net(x) through a cached HIP graph (re-captured when the input shape, the device, the arithmetic form or any parameter
    changes).  Same bits as the eager call (tests/test_encoder.py).  "Any parameter changes" = the storage pointer or the version
    counter of one of the module's Parameter OBJECTS (load_state_dict, optimiser steps, .to(), in-place edits); the list of those
    objects is looked up once -- after REPLACING a Parameter object (`net.conv1.weight = nn.Parameter(...)`) call `forget_graph(net)`.
    precision "fp32": that is all.  precision "split" (ResUNet.check_operand_range): parameters beyond the split's range take the
    fp32 graph; otherwise the replay's range flag decides -- here, after waiting for the stream, or, with defer_range_check=True,
    whenever the caller next synchronises anyway: the replay's `_Run` is then left in net.__dict__["_gpnerf_pending_run"] for the
    caller to take (Renderer keeps it with the frame's record), and a raised flag means: discard what was computed from the
    result and encode again with `net.forward_exact`."""
    _require_gpu_inference(x, net.training)
    key = _graph_key(net, x)
    cls = net.check_operand_range(x.shape[-2], x.shape[-1], params_key=key[3])
    exact = cls == "exact"
    slot = "_gpnerf_graph_f32" if exact else "_gpnerf_graph"
    hit = net.__dict__.get(slot)
    if hit is None or hit[0] != key:
        hit = (key, _EncoderGraph(net, x.float(), exact))
        net.__dict__[slot] = hit
    if exact:
        return hit[1](x)
    if net.__dict__.get("_gpnerf_pending_run") is None:
        hit[1].run.clear()                 # nothing in flight whose verdict is still owed: this replay starts from a zero flag
    y = hit[1](x)
    # both classes: "static" parameters exclude an overflow for in-range images only (see forward())
    if defer_range_check:
        net.__dict__["_gpnerf_pending_run"] = hit[1].run
    else:
        torch.cuda.current_stream(x.device).synchronize()
        if hit[1].run.raised():
            hit[1].run.clear()
            return net.forward_exact(x)
    return y


def range_check_pending(net):
    """After forward_graphed(..., defer_range_check=True) and a synchronisation with its stream: True when that replay left the
    split-f16 operand range (its result is NaN-ridden and must be replaced by net.forward_exact's).  Clears the flag."""
    run = net.__dict__.pop("_gpnerf_pending_run", None)
    if run is None or not run.raised():
        return False
    run.clear()
    return True


def forget_graph(net):
    """Drop the cached graph and parameter list of `net` (see forward_graphed)."""
    net.__dict__.pop("_gpnerf_graph", None)
    net.__dict__.pop("_gpnerf_graph_f32", None)
    net.__dict__.pop("_gpnerf_params", None)


def build_encoder(cfg, precision="fp32"):
    """Same cfg keys as UNet.py:236-242.  precision: the arithmetic form (plugins hip_encoder = "fp32", hip_encoder_fast = "split")."""
    net = ResUNet(encoder=cfg.encoder.name, out_ch=cfg.encoder.out_ch)
    net.precision = precision
    return net
