"""U-Nets: state_dict compatibility and outputs against vectors recorded from the reference modules
(tests/golden/unet_golden.npz); the HIP epilogues against plain-torch float32 post-processing."""
import numpy as np
import pytest
import torch

from helpers import GOLDEN
from mpp_cnn_rs_object_detection_amd import unet


def recipe_state_dict(module, seed):
    """Same recipe as tests/golden/make_golden.py: weights keyed by position in the sorted key list."""
    sd = module.state_dict()
    new = {}
    for i, k in enumerate(sorted(sd.keys())):
        v = sd[k]
        g = torch.Generator().manual_seed(seed * 1000 + i)
        if k.endswith("num_batches_tracked"):
            new[k] = torch.zeros_like(v)
        elif k.endswith("running_var"):
            new[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            new[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif v.dim() >= 2:
            new[k] = torch.randn(v.shape, generator=g) * (2.0 / v[0].numel()) ** 0.5
        elif k.endswith("weight"):
            new[k] = 1.0 + 0.1 * torch.randn(v.shape, generator=g)
        else:
            new[k] = 0.05 * torch.randn(v.shape, generator=g)
    return new


@pytest.fixture(scope="module")
def nets():
    pos, shp = unet.PosNet(), unet.ShapeNet()
    pos.load_state_dict(recipe_state_dict(pos, 1))
    shp.load_state_dict(recipe_state_dict(shp, 2))
    return pos.eval(), shp.eval()


def test_state_dict_keys_and_sizes_match_reference(nets):
    z = np.load(f"{GOLDEN}/unet_golden.npz")
    pos, shp = nets
    assert sorted(pos.state_dict().keys()) == [str(k) for k in z["keys_pos"]]
    assert sorted(shp.state_dict().keys()) == [str(k) for k in z["keys_shp"]]
    assert sum(p.numel() for p in pos.parameters()) == int(z["n_params_pos"]) == 1928483
    assert sum(p.numel() for p in shp.parameters()) == int(z["n_params_shp"]) == 1931552


def test_forward_matches_reference_outputs_cpu(nets):
    z = np.load(f"{GOLDEN}/unet_golden.npz")
    pos, shp = nets
    img = torch.from_numpy(z["image"])
    H, W = img.shape[1:]
    with torch.no_grad():
        padded, pad = unet.pad_before_infer(img, 3)
        assert tuple(padded.shape[1:]) == (48, 56) and pad == [4, 4]
        out = pos(padded.unsqueeze(0))[0]
        logits = [t[0] for t in shp(padded.unsqueeze(0))]
    np.testing.assert_allclose(out[:, :H, :W].numpy(), z["pos_out"], rtol=1e-4, atol=1e-5)
    det = unet.detection_map_torch(out, H, W, float(z["div_w"]), float(z["div_b"]))
    np.testing.assert_allclose(det.numpy(), z["det"], rtol=1e-4, atol=1e-6)
    marks = unet.marks_torch(logits, H, W)
    got = np.stack([m.permute(2, 0, 1).numpy() for m in marks])
    np.testing.assert_allclose(got[:, :, ::7, ::5], z["shape_out_sample"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(got, z["shape_out"].astype(np.float32), atol=2e-3)


@pytest.mark.gpu
def test_hip_epilogues_match_torch_float32(nets):
    from mpp_cnn_rs_object_detection_amd import hip_api
    z = np.load(f"{GOLDEN}/unet_golden.npz")
    pos, shp = nets
    import copy
    runner = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0,
                               div_clf=(float(z["div_w"]), float(z["div_b"])))
    img = torch.from_numpy(z["image"]).permute(1, 2, 0).contiguous()
    det, marks = runner.infer(img)
    torch.cuda.synchronize()
    # against the reference's recorded outputs (whole pipeline on the GPU)
    np.testing.assert_allclose(det.cpu().numpy(), z["det"], rtol=2e-3, atol=2e-4)
    got = np.stack([m.permute(2, 0, 1).cpu().numpy() for m in marks])
    np.testing.assert_allclose(got, z["shape_out"].astype(np.float32), atol=3e-3)
    # epilogues alone, same inputs, against plain torch float32 of the same op (tolerance 1e-5:
    # float32 expf differs in the last ulp between ocml and torch's kernels)
    g = torch.Generator().manual_seed(0)
    for (H, W, Hp, Wp) in ((44, 52, 48, 56), (64, 64, 64, 64), (33, 129, 40, 136), (1, 70, 8, 72)):
        pos_out = torch.randn((3, Hp, Wp), generator=g).cuda()
        logits = (3.0 * torch.randn((32, Hp, Wp), generator=g)).cuda()
        d = torch.empty((H, W), device="cuda")
        m = torch.empty((H, W, 32), device="cuda")
        runner.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        runner.ctx.posnet_epilogue(pos_out, H, W, -10.8, -2.1, d)
        runner.ctx.shapenet_epilogue(logits, H, W, m)
        torch.cuda.synchronize()
        if H > 1:
            ref_d = unet.detection_map_torch(pos_out.cpu(), H, W, -10.8, -2.1)
            np.testing.assert_allclose(d.cpu().numpy(), ref_d.numpy(), rtol=1e-5, atol=1e-6)
        ref_m = unet.marks_torch([logits.cpu()], H, W)[0]
        np.testing.assert_allclose(m.cpu().numpy(), ref_m.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(m.sum(-1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.gpu
def test_fused_conv_epilogue_matches_the_module_path(nets):
    """mpp_affine_relu (bias + folded BatchNorm + ReLU in one pass) against nn.Sequential(conv, bn, relu), float32 and
    bf16, vector and odd plane sizes; and the whole fused forward against the module forward on a 1-Mpx image."""
    import copy
    from mpp_cnn_rs_object_detection_amd import hip_api
    pos, shp = nets
    runner = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0, layout="nchw")
    g = torch.Generator().manual_seed(1)
    for (C, H, W) in ((32, 64, 64), (64, 6, 7), (256, 3, 5)):
        x = torch.randn((1, C, H, W), generator=g).cuda()
        scale = (torch.rand(C, generator=g) + 0.5).cuda()
        shift = torch.randn(C, generator=g).cuda()
        ref = torch.relu(x * scale[None, :, None, None] + shift[None, :, None, None])
        y = x.clone()
        runner.ctx.set_stream(torch.cuda.current_stream().cuda_stream)
        runner.ctx.affine_relu(y, scale, shift)
        torch.cuda.synchronize()
        np.testing.assert_allclose(y.cpu().numpy(), ref.cpu().numpy(), rtol=1e-6, atol=1e-6)
        yb = x.to(torch.bfloat16).clone()
        refb = torch.relu(yb.float() * scale[None, :, None, None] + shift[None, :, None, None]).to(torch.bfloat16)
        runner.ctx.affine_relu(yb, scale, shift)
        torch.cuda.synchronize()
        assert torch.equal(yb, refb)                              # same float32 arithmetic, same rounding to bf16
    img = torch.rand((1024, 1024, 3), generator=g)
    det_f, marks_f = runner.infer(img)                                # >= 1 Mpx: fused path
    runner.fused = False
    det_u, marks_u = runner.infer(img)
    torch.cuda.synchronize()
    np.testing.assert_allclose(det_f.cpu().numpy(), det_u.cpu().numpy(), rtol=1e-3, atol=1e-4)
    for a, b in zip(marks_f, marks_u):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-4)


def _nhwc(x):
    """[1,C,H,W] values in NHWC memory (what the channels-last path passes around)."""
    return x.contiguous(memory_format=torch.channels_last)


@pytest.mark.gpu
def test_nhwc_glue_matches_torch(nets):
    """mpp_nhwc_glue = reflect_pad(relu(affine(maxpool(cat)))) in one pass, against the torch ops it replaces
    (F.pad reflect, BatchNorm folded + ReLU, MaxPool2d(2), torch.cat): same float32 arithmetic, so bit-exact in
    float32 and in bfloat16; vector and one-element-per-lane kernels; in place; the float32 -> bfloat16 stem."""
    import copy
    import torch.nn.functional as F
    pos, shp = nets
    ctx = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0).ctx
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(5)
    for dtype in (torch.float32, torch.bfloat16):
        for (C0, C1, H, W) in ((32, 0, 16, 24), (64, 64, 8, 12), (3, 0, 10, 14), (32, 32, 2, 2), (8, 24, 6, 10), (5, 3, 4, 6)):
            x0 = _nhwc(torch.randn((1, C0, H, W), generator=g).cuda().to(dtype))
            x1 = _nhwc(torch.randn((1, C1, H, W), generator=g).cuda().to(dtype)) if C1 else None
            C = C0 + C1
            scale = (torch.rand(C, generator=g) - 0.3).cuda()           # some negative scales: max-pool must come after f
            shift = torch.randn(C, generator=g).cuda()
            cat = torch.cat([x0, x1], dim=1) if C1 else x0

            def f(t):
                return torch.relu(t.float() * scale[None, :, None, None] + shift[None, :, None, None]).to(dtype)

            cases = [
                (dict(pad=1), F.pad(cat, (1, 1, 1, 1), mode="reflect")),
                (dict(pad=1, scale=scale, shift=shift), F.pad(f(cat), (1, 1, 1, 1), mode="reflect")),
                (dict(pad=0, scale=scale, shift=shift), f(cat)),
                (dict(pad=0), cat),
            ]
            if H >= 4 and W >= 4:
                cases += [
                    (dict(pad=1, pool=True), F.pad(F.max_pool2d(cat.float(), 2).to(dtype), (1, 1, 1, 1), mode="reflect")),
                    (dict(pad=1, pool=True, scale=scale, shift=shift),
                     F.pad(F.max_pool2d(f(cat).float(), 2).to(dtype), (1, 1, 1, 1), mode="reflect")),
                ]
            for kw, ref in cases:
                y = ctx.nhwc_glue(x0, x1, **kw)
                torch.cuda.synchronize()
                assert y.shape == ref.shape and y.is_contiguous(memory_format=torch.channels_last)
                assert torch.equal(y, ref), (dtype, C0, C1, H, W, kw)
            if not C1:                                                    # in place
                y = x0.clone(memory_format=torch.preserve_format)
                out = ctx.nhwc_glue(y, pad=0, scale=scale, shift=shift, out=y)
                torch.cuda.synchronize()
                assert out.data_ptr() == y.data_ptr() and torch.equal(y, f(x0))
    # the stem: float32 image -> bfloat16 padded activation
    img = _nhwc(torch.rand((1, 3, 12, 20), generator=g).cuda())
    y = ctx.nhwc_glue(img, pad=1, out_dtype=torch.bfloat16)
    torch.cuda.synchronize()
    assert torch.equal(y, F.pad(img, (1, 1, 1, 1), mode="reflect").to(torch.bfloat16))
    # errors are reported, not swallowed
    with pytest.raises(Exception):
        ctx.nhwc_glue(img.contiguous(), pad=1)                            # NCHW memory
    with pytest.raises(Exception):
        ctx.nhwc_glue(img, pad=1, out=img)                                # in place with a pad


@pytest.mark.gpu
def test_nhwc_epilogues_equal_the_planar_ones(nets):
    """Same arithmetic in the same order, only the addressing differs: bit-identical outputs for float32; for bfloat16
    inputs identical to the planar kernels fed the same (bf16-rounded) values."""
    import copy
    pos, shp = nets
    ctx = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0).ctx
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    g = torch.Generator().manual_seed(9)
    for dtype in (torch.float32, torch.bfloat16):
        for (H, W, Hp, Wp) in ((44, 52, 48, 56), (64, 64, 64, 64), (33, 129, 40, 136), (1, 70, 8, 72), (5, 1, 8, 8)):
            pos_out = torch.randn((1, 3, Hp, Wp), generator=g).cuda().to(dtype)
            logits = (3.0 * torch.randn((1, 32, Hp, Wp), generator=g)).cuda().to(dtype)
            d0, d1 = torch.empty((H, W), device="cuda"), torch.empty((H, W), device="cuda")
            m0, m1 = torch.empty((H, W, 32), device="cuda"), torch.empty((H, W, 32), device="cuda")
            ctx.posnet_epilogue(pos_out[0].float().contiguous(), H, W, -10.8, -2.1, d0)
            ctx.shapenet_epilogue(logits[0].float().contiguous(), H, W, m0)
            ctx.posnet_epilogue_nhwc(_nhwc(pos_out), H, W, -10.8, -2.1, d1)
            ctx.shapenet_epilogue_nhwc(_nhwc(logits), H, W, m1)
            torch.cuda.synchronize()
            assert torch.equal(d0, d1), (dtype, H, W)
            assert torch.equal(m0, m1), (dtype, H, W)


@pytest.mark.gpu
def test_channels_last_forward_matches_the_module_path(nets):
    """The whole channels-last forward (MIOpen NHWC convolutions + mpp_nhwc_glue) against the plain nn.Module forward,
    float32 within 1e-3 (different convolution algorithms), on a size that needs the 2^depth padding; and the bf16
    variant within bf16 accuracy of it."""
    import copy
    pos, shp = nets
    g = torch.Generator().manual_seed(2)
    img = torch.rand((200, 330, 3), generator=g)
    runner = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0, layout="nhwc")
    runner.min_fused_pixels = 0                                           # small image, but through the fused path
    det_c, marks_c = runner.infer(img)
    runner.fused = False                                                  # nn.Module path
    det_u, marks_u = runner.infer(img)
    torch.cuda.synchronize()
    np.testing.assert_allclose(det_c.cpu().numpy(), det_u.cpu().numpy(), rtol=1e-3, atol=1e-4)
    for a, b in zip(marks_c, marks_u):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-3, atol=1e-4)
    rb = unet.ScoreMapNets(copy.deepcopy(pos), copy.deepcopy(shp), device=0, layout="nhwc", dtype=torch.bfloat16)
    rb.min_fused_pixels = 0
    det_b, marks_b = rb.infer(img)
    torch.cuda.synchronize()
    assert float((det_b - det_u).abs().max()) < 0.08
    for a, b in zip(marks_b, marks_u):
        assert float((a - b).abs().max()) < 0.08
        np.testing.assert_allclose(a.sum(-1).cpu().numpy(), 1.0, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cin,H,W", [(32, 64, 128), (64, 40, 72), (32, 17, 33), (64, 256, 320)])
def test_mfma_conv3x3_c32_equals_the_library_convolution(cin, H, W):
    """``mpp_conv3x3_c32`` (csrc/mpp_conv.hip): Conv2d(C_in -> 32, 3x3, reflect) + folded BatchNorm + ReLU of a DoubleConv
    (unet_parts.py:12-31), the concat of Up as a second source, the producer's BatchNorm + ReLU at the load -- against
    F.pad(mode='reflect') + F.conv2d in float32 (the summation order differs: 1e-4 relative)."""
    import torch.nn.functional as F
    from mpp_cnn_rs_object_detection_amd import hip_api
    torch.manual_seed(cin + H)
    dev = torch.device("cuda", 0)
    ctx = hip_api.MppContext(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    x = torch.randn((1, cin, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn((32, cin, 3, 3), device=dev) / (3.0 * cin ** 0.5)
    sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    isc, ish = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    wp = w.permute(2, 3, 1, 0).reshape(9, cin // 32, 32, 32).permute(1, 0, 2, 3).contiguous()
    x0 = x[:, :32].contiguous(memory_format=torch.channels_last)
    x1 = x[:, 32:].contiguous(memory_format=torch.channels_last) if cin == 64 else None
    for in_aff in (False, True):
        xin = x.clone()
        if in_aff:
            xin[:, :32] = torch.relu(xin[:, :32] * isc.view(1, -1, 1, 1) + ish.view(1, -1, 1, 1))
        ref = torch.relu(F.conv2d(F.pad(xin, (1, 1, 1, 1), mode="reflect"), w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
        got = ctx.conv3x3_c32(x0, wp, x1=x1, in_scale=isc if in_aff else None, in_shift=ish if in_aff else None,
                              out_scale=sc, out_shift=sh, relu=True)
        torch.cuda.synchronize()
        assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
        np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)
    # no epilogue: the raw convolution
    raw = ctx.conv3x3_c32(x0, wp, x1=x1, relu=False)
    np.testing.assert_allclose(raw.cpu().numpy(), F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w).cpu().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,ldh,ldw", [(64, 128, 64, 128), (37, 75, 48, 80), (5, 31, 16, 32), (130, 33, 144, 48)])
def test_fused_shapenet_heads_equal_convolutions_and_softmax(H, W, ldh, ldw):
    """``mpp_shapenet_heads`` (csrc/mpp_conv.hip): the three Conv2d(32, 32, 1x1) heads of shape_net.py:12-46 with their
    biases and the softmax over the 32 classes in one pass, cropped from the padded activations [ldh][ldw] to [H][W] --
    against F.conv2d + torch.softmax in float32 (row lengths that are not multiples of the 32-pixel groups included)."""
    import torch.nn.functional as F
    from mpp_cnn_rs_object_detection_amd import hip_api
    torch.manual_seed(H * 1000 + W)
    dev = torch.device("cuda", 0)
    ctx = hip_api.MppContext(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    h = (torch.randn((1, 32, ldh, ldw), device=dev) * 2.0).contiguous(memory_format=torch.channels_last)
    w = torch.randn((3, 32, 32), device=dev) / 3.0
    b = torch.randn((3, 32), device=dev)
    marks = [torch.full((H, W, 32), -1.0, device=dev) for _ in range(3)]
    ctx.shapenet_heads(h, w, b, H, W, marks)
    torch.cuda.synchronize()
    for k in range(3):
        logits = F.conv2d(h, w[k].reshape(32, 32, 1, 1), b[k])[0, :, :H, :W]
        ref = torch.softmax(logits, dim=0).permute(1, 2, 0)
        np.testing.assert_allclose(marks[k].cpu().numpy(), ref.cpu().numpy(), rtol=2e-5, atol=1e-7)
        np.testing.assert_allclose(marks[k].sum(-1).cpu().numpy(), 1.0, atol=1e-5)
    with pytest.raises(ValueError):
        ctx.shapenet_heads(h, w[:2], b, H, W, marks)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("H,W", [(64, 128), (37, 75), (17, 16), (130, 33)])
def test_stem_convolution_equals_the_library_convolution(H, W):
    """``mpp_conv3x3_stem`` (csrc/mpp_conv.hip): Conv2d(3 -> 32, 3x3, reflect) + folded BatchNorm + ReLU of the first
    DoubleConv (unet_parts.py:12-31) in one pass over the picture -- against F.pad(mode='reflect') + F.conv2d in float32,
    sizes that are not multiples of the 16 x 16 tiles included."""
    import torch.nn.functional as F
    from mpp_cnn_rs_object_detection_amd import hip_api
    torch.manual_seed(H * 7 + W)
    dev = torch.device("cuda", 0)
    ctx = hip_api.MppContext(0)
    ctx.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    x = torch.rand((1, 3, H, W), device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn((32, 3, 3, 3), device=dev) / 5.0
    sc, sh = torch.rand(32, device=dev) + 0.5, torch.randn(32, device=dev) * 0.1
    got = ctx.conv3x3_stem(x, w.permute(2, 3, 1, 0).reshape(9, 3, 32).contiguous(), sc, sh)
    torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), w) * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    assert got.shape == ref.shape and got.is_contiguous(memory_format=torch.channels_last)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-5)
    ctx.close()
