"""GPU: the image encoder's HIP operators on channels-last activations (csrc/gpnerf_conv.hip) against plain PyTorch fp32/fp64
references of the same ops on the CPU: reflect-padded convolution (MFMA implicit GEMM in both arithmetic forms: fp32 operands on
the fp32 MFMA = the default, f16 hi/lo operands on the f16 MFMA = the fast mode), InstanceNorm + residual + activation, bilinear
x2 upsampling."""
import importlib

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def enc():
    return importlib.import_module("gp-nerf_amd.encoder")


@pytest.fixture(autouse=True, params=["fp32", "split"])
def form(request, enc):
    """every test of this module runs in both arithmetic forms of the convolutions (include/gpnerf_hip.h `exact`): fp32 operands
    on the fp32 MFMA (the encoder's default) and f16 hi/lo operands on the f16 MFMA (ResUNet.precision = "split")"""
    with enc._form(request.param == "fp32"):
        yield request.param


CONVS = [  # cin, cout, ks, stride, H, W, bias
    (3, 64, 7, 2, 40, 56, False),        # stem (narrow input)
    (3, 64, 7, 2, 37, 45, False),        # stem: ragged tiles, odd sizes
    (3, 64, 7, 2, 130, 70, True),        # stem: several tiles, bias
    (3, 32, 7, 2, 8, 8, False),          # stem: one output tile, one 32-channel tile
    (3, 96, 7, 2, 21, 90, False),        # stem: three output tiles (one per workgroup)
    (4, 64, 7, 2, 24, 24, False),        # stem: four input channels
    (64, 128, 3, 2, 33, 47, False),      # stride-2 3x3: ragged tiles, odd sizes
    (128, 64, 3, 2, 70, 130, True),      # stride-2 3x3: several tiles
    (64, 64, 3, 1, 33, 47, False),       # ragged pixel tiles
    (64, 128, 3, 2, 32, 32, False),
    (64, 128, 1, 2, 32, 32, False),      # projected shortcut
    (128, 128, 3, 1, 16, 20, False),
    (256, 256, 3, 1, 9, 7, False),       # smaller than one pixel tile
    (256, 128, 3, 1, 12, 12, True),      # decoder conv with bias
    (128, 32, 3, 1, 24, 24, True),
    (32, 32, 1, 1, 10, 10, True),        # output conv
    (16, 36, 3, 1, 8, 8, True),          # cout not a multiple of 32
]


@pytest.mark.parametrize("cin,cout,ks,stride,H,W,bias", CONVS)
def test_conv2d_nhwc_matches_torch(cin, cout, ks, stride, H, W, bias, form, enc):
    g = torch.Generator().manual_seed(cin * 1000 + cout + ks)
    conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=bias, padding_mode="reflect")
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    x = torch.randn((3, cin, H, W), generator=g) * 2.0 + 0.3
    with torch.no_grad():
        ref = conv.double()(x.double()).float()
        conv = conv.float().to("cuda:0")
        got = enc._conv(conv, x.to("cuda:0"))
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    err = float((got.cpu() - ref).abs().max())
    assert err < 2e-5 * max(1.0, float(ref.abs().max())), err
    # re-packing follows parameter updates (each form keeps its own image of the weight)
    with torch.no_grad():
        conv.weight.mul_(2.0)
        got2 = enc._conv(conv, x.to("cuda:0"))
    ref2 = 2 * ref - (conv.bias.detach().cpu()[None, :, None, None] if bias else 0)
    assert float((got2.cpu() - ref2).abs().max()) < 4e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("cin,cout,ks,stride,H,W,bias", CONVS)
def test_fp32_form_matches_torch_on_operands_of_any_size(cin, cout, ks, stride, H, W, bias, form, enc):
    """The default arithmetic form (include/gpnerf_hip.h `exact = 1`: fp32 operands on v_mfma_f32_32x32x2_f32, the 3x3 kernels'
    sums blocked per 16 input channels) against float64 torch: an fp32 summation's distance (3e-6 relative to the output range),
    operands of ANY magnitude (x 1e4 here: the split form's range is 4 094), re-packed on a parameter change -- and against the
    independent per-operand restatement gpnerf_conv2d_nhwc_exact (enc._conv_exact: one FMA chain per output straight from the
    PyTorch weight), from which it differs by the order of the sum only."""
    if form != "fp32":
        pytest.skip("the fp32 form's own test")
    g = torch.Generator().manual_seed(cin * 1000 + cout + ks + 7)
    conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=bias, padding_mode="reflect")
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) * (2.0 / (cin * ks * ks)) ** 0.5)
        if bias:
            conv.bias.copy_(torch.randn(cout, generator=g) * 0.1)
    x = (torch.randn((3, cin, H, W), generator=g) * 2.0 + 0.3) * 1e4
    with torch.no_grad():
        ref = conv.double()(x.double()).float()
        conv = conv.float().to("cuda:0")
        with enc._form(True):
            got = enc._conv(conv, x.to("cuda:0"))
        chain = enc._conv_exact(conv, x.to("cuda:0"))
    scale = max(1.0, float(ref.abs().max()))
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    assert float((got.cpu() - ref).abs().max()) < 3e-6 * scale
    assert float((chain.cpu() - ref).abs().max()) < 3e-6 * scale and float((chain - got).abs().max()) < 3e-6 * scale
    assert "_gpnerf_packed_f32" in conv.__dict__ and "_gpnerf_packed" not in conv.__dict__
    with torch.no_grad():
        conv.weight.mul_(2.0)
        with enc._form(True):
            got2 = enc._conv(conv, x.to("cuda:0"))
    ref2 = 2 * ref - (conv.bias.detach().cpu()[None, :, None, None] if bias else 0)
    assert float((got2.cpu() - ref2).abs().max()) < 6e-6 * scale


@pytest.mark.parametrize("c,H,W,act,res", [(64, 37, 41, 1, False), (128, 16, 16, 1, True), (32, 24, 40, 2, False), (256, 5, 5, 0, False)])
def test_instance_norm_act_nhwc_matches_torch(c, H, W, act, res, enc):
    g = torch.Generator().manual_seed(c + H)
    norm = torch.nn.InstanceNorm2d(c, track_running_stats=False, affine=True)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(c, generator=g))
        norm.bias.copy_(0.2 * torch.randn(c, generator=g))
    x = torch.randn((3, c, H, W), generator=g) * 3.0 + 1.5
    r = torch.randn((3, c, H, W), generator=g) if res else None
    with torch.no_grad():
        y = norm.double()(x.double()) + (r.double() if res else 0)
        ref = (F.relu(y) if act == 1 else F.elu(y) if act == 2 else y).float()
        norm = norm.float().to("cuda:0")
        got = enc._norm_act(norm, x.to("cuda:0"), act, residual=r.to("cuda:0") if res else None)
    assert float((got.cpu() - ref).abs().max()) < 2e-5


def test_norm_from_the_convolutions_tile_sums_equals_the_separate_pass(enc):
    """The table a convolution's last workgroup merges from the tiles' (sum, M2) (gpnerf_conv2d_norm_nhwc) against the stand-alone
    operator's double-precision pass over the convolution's output -- on ordinary activations and on a NEARLY CONSTANT channel
    (offset 1e3, spread 1e-2: E[y^2] - mean^2 in float32 tile sums would have no correct digit left; sums of squares about the
    means keep the variance to float32's own accuracy)."""
    g = torch.Generator().manual_seed(11)
    for cin, cout, ks, stride, H, W in ((64, 64, 3, 1, 33, 47), (64, 128, 3, 2, 32, 36), (3, 64, 7, 2, 40, 56), (128, 32, 3, 1, 24, 24), (64, 64, 1, 1, 40, 24)):
        conv = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=True, padding_mode="reflect").to("cuda:0")
        norm = torch.nn.InstanceNorm2d(cout, track_running_stats=False, affine=True).to("cuda:0")
        x = (torch.randn((3, cin, H, W), generator=g) * 2 + 0.5).to("cuda:0")
        with torch.no_grad():
            y, ts = enc._conv(conv, x, stats=True)
            y2, tab = enc._conv_norm(conv, norm, x)
            a = enc._apply(y2, tab, 2)
            b = enc._norm_act(norm, y, 2)
        assert ts.shape[0] == 3 and ts.shape[2:] == (cout, 3) and torch.equal(y, y2)
        assert float((a - b).abs().max()) < 2e-6, (cin, cout, ks, stride)
        if cin >= 16 and ks == 3 and stride == 1:
            with torch.no_grad():
                conv.weight.mul_(1e-5)
                conv.bias.fill_(1000.0)
                y3, tab3 = enc._conv_norm(conv, norm, x)
                a3, b3 = enc._apply(y3, tab3, 0), enc._norm_act(norm, y3, 0)
                ref = norm.double()(y3.double())
            # (a float32 value near 1000 carries 6e-5 of absolute noise against a spread of ~1e-4: the normalised values are only
            # meaningful to ~10 %; what is asserted is that the two ways of taking the statistics agree with each other and float64)
            assert float((a3 - b3).abs().max()) < 2e-3 and float((a3.double() - ref).abs().max()) < 2e-3, (cin, cout)


def test_upsample2x_nhwc_matches_torch(enc):
    x = torch.randn((2, 64, 9, 13), generator=torch.Generator().manual_seed(5))
    ref = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    got = enc._upsample2x(x.to("cuda:0"))
    assert got.shape == ref.shape and float((got.cpu() - ref).abs().max()) < 1e-5


def test_conv_entry_point_rejects_unsupported_shapes():
    L = importlib.import_module("gp-nerf_amd._lib")
    lib = L.lib()
    p = 0x1000
    assert lib.gpnerf_conv2d_nhwc(p, 1, 8, 8, 24, p, None, 32, 3, 1, p, None, None, 0, None) == -1      # cin neither < 8 nor a multiple of 16
    assert lib.gpnerf_conv2d_nhwc(p, 1, 8, 8, 16, p, None, 32, 5, 1, p, None, None, 0, None) == -1      # 5x5 is not built
    assert lib.gpnerf_conv2d_nhwc(p, 1, 1, 8, 16, p, None, 32, 3, 1, p, None, None, 0, None) == -1      # reflection needs pad < size
    assert lib.gpnerf_conv2d_nhwc(p, 1, 8, 8, 16, p, None, 30, 3, 1, p, None, None, 0, None) == -1      # cout not a multiple of 4
    assert lib.gpnerf_conv2d_nhwc(None, 0, 8, 8, 16, p, None, 32, 3, 1, p, None, None, 0, None) == 0     # nothing to do
    assert lib.gpnerf_conv_out_tiles(128, 128, 64, 3, 1) == 16 * 4 and lib.gpnerf_conv_out_tiles(32, 32, 256, 3, 1) == 8 and lib.gpnerf_conv_out_tiles(64, 64, 128, 3, 1) == 32 and lib.gpnerf_conv_out_tiles(128, 128, 64, 3, 2) == 32
    assert lib.gpnerf_conv_packed_bytes(64, 3, 7) == 14 * 2 * 2048      # the 3-channel stem: 7 kernel rows x 2 chunks of (2 x 2 columns x 4 channels)


@pytest.mark.parametrize("cin,cout,H,W", [(64, 64, 33, 47), (128, 128, 16, 20), (256, 256, 9, 7), (64, 96, 40, 40)])
def test_fused_norms_around_a_convolution_match_the_separate_launches(cin, cout, H, W, enc):
    """gpnerf_conv2d_norm_nhwc: (a) the table its last workgroup writes equals the separate reduction's normalisation;
    (b) reading relu(norm(x)) while staging equals convolving the materialised tensor -- bit for bit (same arithmetic, same order);
    (c) the ticket words are zero again; (d) against float64 torch."""
    g = torch.Generator().manual_seed(cin + cout + H)
    dev = "cuda:0"
    c0 = torch.nn.Conv2d(cin, cin, 3, padding=1, bias=False, padding_mode="reflect").to(dev)
    n0 = torch.nn.InstanceNorm2d(cin, track_running_stats=False, affine=True).to(dev)
    c1 = torch.nn.Conv2d(cin, cout, 3, padding=1, bias=True, padding_mode="reflect").to(dev)
    n1 = torch.nn.InstanceNorm2d(cout, track_running_stats=False, affine=True).to(dev)
    with torch.no_grad():
        for m in (n0, n1):
            m.weight.copy_(1 + 0.2 * torch.randn(m.weight.shape, generator=g))
            m.bias.copy_(0.2 * torch.randn(m.bias.shape, generator=g))
        x = (torch.randn((3, cin, H, W), generator=g) * 2 + 0.5).to(dev)
        y0, t0 = enc._conv_norm(c0, n0, x)
        a_sep = enc._norm_act(n0, enc._conv(c0, x), 1)                               # the stand-alone operator: statistics from a double-precision pass
        a_tab = enc._apply(y0, t0, 1)
        assert float((a_sep - a_tab).abs().max()) < 2e-6
        y_mat, t_mat = enc._conv_norm(c1, n1, a_tab)                                  # materialised input
        y_fus, t_fus = enc._conv_norm(c1, n1, y0, in_tab=t0, in_act=1)                # normalised + ReLU'd while staging
        assert torch.equal(y_mat, y_fus) and torch.equal(t_mat, t_fus)
        assert int(enc._ticket_words(x.device).abs().sum()) == 0
        ref = n1.double()(c1.double()(F.relu(n0.double()(c0.double()(x.double())))))
        assert float((enc._apply(y_fus, t_fus, 0).double() - ref).abs().max()) < 5e-5
        if cin == cout:
            out = enc._apply(y_fus, t_fus, 1, residual=y0, res_tab=t0)                # shortcut normalised on the fly
            assert float((out.double() - F.relu(ref + n0(c0(x.double())))).abs().max()) < 5e-5


@pytest.mark.parametrize("ks,stride", [(1, 2), (1, 1), (3, 2)])
def test_input_norm_is_applied_by_every_convolution_that_can_read_through_it(ks, stride, enc):
    """A pending InstanceNorm + ReLU in front of a 1x1 convolution (applied where the direct kernel splits its pixels) and of the
    stride-2 3x3 (applied while it stages its patch): bit for bit the convolution of the materialised tensor -- the stem's output is
    read this way by the first residual unit's two convolutions and never written normalised."""
    g = torch.Generator().manual_seed(10 * ks + stride)
    dev = "cuda:0"
    cin, cout = 64, 96
    c0 = torch.nn.Conv2d(cin, cin, 3, padding=1, bias=False, padding_mode="reflect").to(dev)
    n0 = torch.nn.InstanceNorm2d(cin, track_running_stats=False, affine=True).to(dev)
    c1 = torch.nn.Conv2d(cin, cout, ks, stride=stride, padding=ks // 2, bias=False, padding_mode="reflect").to(dev)
    n1 = torch.nn.InstanceNorm2d(cout, track_running_stats=False, affine=True).to(dev)
    with torch.no_grad():
        n0.weight.copy_(1 + 0.2 * torch.randn(n0.weight.shape, generator=g)); n0.bias.copy_(0.2 * torch.randn(n0.bias.shape, generator=g))
        x = (torch.randn((3, cin, 37, 45), generator=g) * 2 + 0.5).to(dev)
        y0, t0 = enc._conv_norm(c0, n0, x)
        assert enc._fusable_input_norm(c1)
        y_mat, t_mat = enc._conv_norm(c1, n1, enc._apply(y0, t0, 1))
        y_fus, t_fus = enc._conv_norm(c1, n1, y0, in_tab=t0, in_act=1)
        assert torch.equal(y_mat, y_fus) and torch.equal(t_mat, t_fus)
        ref = c1.double()(F.relu(n0.double()(c0.double()(x.double()))))
        assert float((y_fus.double() - ref).abs().max()) < 5e-5 * max(1.0, float(ref.abs().max()))


@pytest.mark.parametrize("ca,cb,cout,H,W", [(128, 128, 128, 64, 64), (64, 64, 32, 72, 40), (32, 16, 64, 9, 33), (128, 128, 128, 24, 40)])
def test_concatenation_read_in_place_equals_the_convolution_of_the_concatenated_tensor(ca, cb, cout, H, W, enc):
    """gpnerf_conv2d_norm_cat_nhwc (4-row and 8-row tile forms): bit for bit `_conv_norm` on torch.cat([a, b], 1), table included."""
    g = torch.Generator().manual_seed(ca + cb + H)
    dev = "cuda:0"
    conv = torch.nn.Conv2d(ca + cb, cout, 3, padding=1, bias=True, padding_mode="reflect").to(dev)
    norm = torch.nn.InstanceNorm2d(cout, track_running_stats=False, affine=True).to(dev)
    a = (torch.randn((3, ca, H, W), generator=g) * 2 + 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    b = (torch.randn((3, cb, H, W), generator=g) - 0.5).to(dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        y_cat, t_cat = enc._conv_norm(conv, norm, torch.cat([a, b], 1))
        y_inp, t_inp = enc._conv_norm_cat(conv, norm, a, b)
        if (H, W) == (24, 40):
            # 144 workgroups: the materialised tensor takes the K-split form (two halves' sums added at the end), the in-place
            # read does not -- the same terms in another order
            assert float((y_cat - y_inp).abs().max()) < 2e-5 and float((t_cat - t_inp).abs().max()) < 2e-5
        else:
            assert torch.equal(y_cat, y_inp) and torch.equal(t_cat, t_inp)
        assert int(enc._ticket_words(a.device).abs().sum()) == 0
        ref = conv.double()(torch.cat([a, b], 1).double())
        assert float((y_inp.double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
