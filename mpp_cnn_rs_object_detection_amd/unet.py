"""PosNet / ShapeNet: the two U-Nets that produce the score maps the sampler runs on.

Same architecture and the same ``state_dict`` keys as the reference (``model_parts/unet/unet.py:24-60``,
``unet_parts.py:12-67``, ``models/position_net/pos_net.py:9-30``, ``models/shape_net/shape_net.py:12-46``) so a
user's ``model.pt`` loads unchanged:  3x3 reflect-padded conv + BatchNorm + ReLU twice per level,
2x2 max-pool down, 2x2 stride-2 transposed conv up, skip concat ``[skip, up]``, 1x1 heads.

MI355X use: convolutions go through PyTorch-ROCm (MIOpen -> MFMA) on channels-last (NHWC) activations; everything
BETWEEN two convolutions -- reflect padding, BatchNorm + bias + ReLU, 2x2 max-pool, the skip concat, the cast to the
compute type -- is one hand-written HBM-bound pass (``mpp_nhwc_glue``, ``csrc/mpp_maps.hip``), and the
post-processing of both nets (sigmoid / divergence / 1x1 classifier / sigmoid, and the softmax over a pixel's 32
contiguous logits) is fused into two more kernels whose outputs stay on the device in exactly the layout the
sampler reads -- the reference's pickle round trip (``pos_net_model.py:407-424`` -> ``data_loaders.py:30-71``)
disappears.
"""
from __future__ import annotations

import json
import os
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from torch import Tensor, nn

HIDDEN_DIMS = (32, 64, 128, 256)
DIV_CLF_W, DIV_CLF_B = -10.812359809875488, -2.128434181213379   # models_storage/posnet/posvec_dota/model_div_clf.pt


class DoubleConv(nn.Module):
    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        self.double_conv = nn.Sequential(
            nn.Conv2d(c_in, c_out, kernel_size=(3, 3), padding=1, padding_mode="reflect"),
            nn.BatchNorm2d(c_out), nn.ReLU(inplace=True),
            nn.Conv2d(c_out, c_out, kernel_size=(3, 3), padding=1, padding_mode="reflect"),
            nn.BatchNorm2d(c_out), nn.ReLU(inplace=True))

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(c_in, c_out))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    def __init__(self, c_in: int, c_out: int):
        super().__init__()
        self.up = nn.ConvTranspose2d(c_in, c_in // 2, kernel_size=(2, 2), stride=(2, 2))
        self.conv = DoubleConv(c_in, c_out)

    def forward(self, x, skip):
        return self.conv(torch.cat([skip, self.up(x)], dim=1))


class Unet(nn.Module):
    def __init__(self, hidden_dims: Sequence[int] = HIDDEN_DIMS, in_channels: int = 3, device=None):
        super().__init__()
        self.descending_path = nn.ModuleList()
        c = in_channels
        for i, h in enumerate(hidden_dims):
            self.descending_path.append(DoubleConv(c, h) if i == 0 else Down(c, h))
            c = h
        self.ascending_path = nn.ModuleList()
        for h in list(hidden_dims)[::-1][1:]:
            self.ascending_path.append(Up(c, h))
            c = h
        self.out_channels = c
        self.depth = len(hidden_dims) - 1

    def forward(self, x: Tensor) -> Tensor:
        x = x.float()
        skips = []
        for down in self.descending_path:
            x = down(x)
            skips.append(x)
        for up, skip in zip(self.ascending_path, skips[::-1][1:]):
            x = up(x, skip)
        return x


class PosNet(nn.Module):
    """3 output channels: vector field (2) + mask logit (reference ``pos_net.py:9-30``)."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, device=None, hidden_dims: Sequence[int] = HIDDEN_DIMS):
        super().__init__()
        self.backbone = Unet(hidden_dims, in_channels)
        self.final_layer = nn.Conv2d(self.backbone.out_channels, out_channels, kernel_size=(1, 1))

    def forward(self, x: Tensor) -> Tensor:
        return self.final_layer(self.backbone(x))


class ShapeNet(nn.Module):
    """Three 1x1 heads of 32 logits each: size, ratio, angle (reference ``shape_net.py:12-46``)."""

    def __init__(self, in_channels: int = 3, out_features: int = 3, out_feat_size=32, device=None,
                 hidden_dims: Sequence[int] = HIDDEN_DIMS):
        super().__init__()
        self.backbone = Unet(hidden_dims, in_channels)
        sizes = [out_feat_size] * out_features if isinstance(out_feat_size, int) else list(out_feat_size)
        self.final_layers = nn.ModuleList(
            [nn.Sequential(nn.Conv2d(self.backbone.out_channels, s, kernel_size=(1, 1))) for s in sizes])

    def forward(self, x: Tensor) -> List[Tensor]:
        h = self.backbone(x)
        return [fl(h) for fl in self.final_layers]

    def forward_with_softmax(self, x: Tensor, temperature: float = 1.0) -> List[Tensor]:
        return [torch.softmax(t / temperature, dim=1) for t in self.forward(x)]


def pad_before_infer(image: Tensor, depth: int) -> Tuple[Tensor, List[int]]:
    """Zero-pad bottom/right to a multiple of 2**depth (reference ``unet.py:9-21``); image is [C,H,W]."""
    div = 2 ** depth
    pad = [(div - s % div) % div for s in image.shape[1:]]
    if pad[0] or pad[1]:
        image = F.pad(image, pad=(0, pad[1], 0, pad[0]))
    return image, pad


# ---- plain-torch post-processing: the float32 reference the HIP epilogues are tested against ----------
def torch_divergence_ij(vec: Tensor) -> Tensor:
    """d vec[0]/d row + d vec[1]/d col with ``torch.gradient`` (reference ``torch_div.py:8-27``, 'ij')."""
    return torch.gradient(vec[0], dim=0)[0] + torch.gradient(vec[1], dim=1)[0]


def detection_map_torch(pos_out: Tensor, H: int, W: int, w: float = DIV_CLF_W, b: float = DIV_CLF_B) -> Tensor:
    """pos_out [3,Hp,Wp] -> det [H,W] (reference ``pos_net_model.py:186-200`` crop, then ``:338-346``)."""
    out = pos_out[:, :H, :W].float()
    mask = torch.sigmoid(out[2])
    return torch.sigmoid(w * (torch_divergence_ij(out[:2]) * mask) + b)


def marks_torch(logits: Sequence[Tensor], H: int, W: int) -> List[Tensor]:
    """three [32,Hp,Wp] logits -> three [H,W,32] softmax maps (``shape_net_model.py:139-141`` + ``data_loaders.py:54``)."""
    return [torch.softmax(t[:, :H, :W].float(), dim=0).permute(1, 2, 0).contiguous() for t in logits]


def load_div_clf(model_dir: str) -> Tuple[float, float]:
    """The two scalars of the 1x1 'div_clf' conv: ``model_div_clf.pt`` (reference) or its JSON twin."""
    pt, js = os.path.join(model_dir, "model_div_clf.pt"), os.path.join(model_dir, "model_div_clf.json")
    if os.path.exists(pt):
        sd = torch.load(pt, map_location="cpu", weights_only=True)
        return float(sd["1.weight"].flatten()[0]), float(sd["1.bias"].flatten()[0])
    if os.path.exists(js):
        with open(js) as f:
            d = json.load(f)
        return float(d["weight"]), float(d["bias"])
    return DIV_CLF_W, DIV_CLF_B


def load_torch_model(module: nn.Module, model_dir: str) -> bool:
    """``TorchModel._load`` (reference ``base/base_model.py:35-49``): model.pt, else the last checkpoint_*.pt."""
    import glob
    path = os.path.join(model_dir, "model.pt")
    if not os.path.exists(path):
        ck = sorted(glob.glob(os.path.join(model_dir, "checkpoint_*.pt")))
        if not ck:
            return False
        path = ck[-1]
    sd = torch.load(path, map_location="cpu", weights_only=True)
    module.load_state_dict(sd.get("model_state_dict", sd) if isinstance(sd, dict) else sd)
    return True


#: receptive-field radius of the three-level U-Net (two 3x3 convolutions per level: 2 + 4 + 8 + 16 + 8 + 4 + 2 = 44 px
#: at full resolution) plus the one-pixel stencil of the divergence (torch_div.py:8-43), rounded up to a multiple of 8
HALO = 48


def halo_crop(region, shape, halo: int = HALO, align: int = 8):
    """(x0, x1, y0, y1) -> the crop the nets must see to reproduce the maps of that region (see ``infer_region``)"""
    x0, x1, y0, y1 = (int(v) for v in region)
    H, W = int(shape[0]), int(shape[1])
    return (max(0, x0 - halo) // align * align, min(H, x1 + halo), max(0, y0 - halo) // align * align, min(W, y1 + halo))


class ScoreMapNets:
    """PosNet + ShapeNet inference on one GPU with the fused HIP epilogues."""

    def __init__(self, posnet: PosNet, shapenet: ShapeNet, device: int = 0, div_clf: Tuple[float, float] = None,
                 dtype: torch.dtype = torch.float32, ctx=None, layout: Optional[str] = None):
        from .hip_api import MppContext
        self.device = torch.device("cuda", device)
        # "nhwc": channels-last activations, everything between two convolutions in one `mpp_nhwc_glue` pass;
        # "nchw": MIOpen on contiguous NCHW with torch's reflection pad and `mpp_affine_relu`
        self.layout = layout or os.environ.get("MPP_UNET_LAYOUT", "nhwc")
        if self.layout not in ("nhwc", "nchw"):
            raise ValueError("layout must be 'nhwc' or 'nchw'")
        # (torch's F.pad(mode="reflect") only has a contiguous-NCHW kernel: channels-last *modules* pay two layout
        # conversions per 3x3 convolution -- 55 % of the forward in the first profile; hence the hand-written glue)
        self.pos = posnet.to(self.device).eval()
        self.shp = shapenet.to(self.device).eval()
        self.div_w, self.div_b = div_clf or (DIV_CLF_W, DIV_CLF_B)
        self.dtype = dtype
        self.ctx = ctx or MppContext(device)
        self.fused = os.environ.get("MPP_UNET_UNFUSED", "0") != "1"
        self.mfma_conv = os.environ.get("MPP_UNET_MFMA_CONV", "1") != "0"      # csrc/mpp_conv.hip for the 32-channel level
        self.two_streams = os.environ.get("MPP_UNET_TWO_STREAMS", "1") != "0"  # PosNet and ShapeNet on a stream each
        self._streams = None
        # below this many pixels the forward is launch-bound and the plain nn.Module path (fewer host calls) is faster:
        # 512x512 3.4 ms vs 4.8 ms; 2048x2048 41 ms (nchw) / 36 ms (nhwc) float32, 31 / 15 ms bfloat16
        self.min_fused_pixels = 1 << 20
        self._fold_cache = {}

    # -- fused inference path: conv (MIOpen) + ONE pass of bias/BatchNorm/ReLU (mpp_affine_relu) per convolution ------
    def _folded(self, conv: nn.Conv2d, bn: nn.BatchNorm2d):
        key = id(conv)
        if key not in self._fold_cache:
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float()
            bias = conv.bias.float() if conv.bias is not None else torch.zeros_like(scale)
            shift = (bn.bias + (bias - bn.running_mean) * scale).float()
            self._fold_cache[key] = (scale.contiguous(), shift.contiguous())
        return self._fold_cache[key]

    def _double_conv(self, dc: DoubleConv, x: Tensor) -> Tensor:
        seq = dc.double_conv
        for conv, bn in ((seq[0], seq[1]), (seq[3], seq[4])):
            scale, shift = self._folded(conv, bn)
            y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="reflect"), conv.weight, None)
            if not y.is_contiguous():
                y = y.contiguous()
            self.ctx.affine_relu(y, scale, shift)
            x = y
        return x

    # -- channels-last path: pad / pool / concat / BatchNorm / ReLU between two convolutions are ONE HIP pass ------------
    def _weights(self, mod):
        """(weight, bias) of a convolution in the compute dtype, channels-last; the 3x3 convolutions' bias lives in the
        folded shift, so only the transposed convolutions and the 1x1 heads use theirs."""
        key = (id(mod), self.dtype)
        if key not in self._fold_cache:
            w = mod.weight.detach().to(self.dtype).contiguous(memory_format=torch.channels_last)
            b = mod.bias.detach().to(self.dtype) if mod.bias is not None else None
            self._fold_cache[key] = (w, b)
        return self._fold_cache[key]

    @staticmethod
    def _cl(x: Tensor) -> Tensor:
        return x if x.is_contiguous(memory_format=torch.channels_last) else x.contiguous(memory_format=torch.channels_last)

    def _packed(self, conv: nn.Conv2d) -> Tensor:
        """weights of a Conv2d(C_in -> 32, 3x3) repacked for ``mpp_conv3x3_c32``: [C_in / 32][kh * 3 + kw][32 in][32 out]"""
        key = (id(conv), "c32")
        if key not in self._fold_cache:
            w = conv.weight.detach().float()                                    # [32, C_in, 3, 3]
            cin = w.shape[1]
            self._fold_cache[key] = (w.permute(2, 3, 1, 0).reshape(9, cin // 32, 32, 32).permute(1, 0, 2, 3).contiguous(),)
        return self._fold_cache[key][0]

    def _packed_heads(self):
        """ShapeNet's heads as one [3,32,32] weight block and one [3,32] bias block for ``mpp_shapenet_heads`` (None unless
        the heads are the reference's three Conv2d(32, 32, 1x1) and the MFMA kernels are on)."""
        key = (id(self.shp), "heads")
        if key not in self._fold_cache:
            convs = [fl[0] for fl in self.shp.final_layers]
            ok = (self.mfma_conv and len(convs) == 3 and all(isinstance(c, nn.Conv2d) and c.kernel_size == (1, 1) and c.in_channels == 32
                                                              and c.out_channels == 32 and len(fl) == 1
                                                              for c, fl in zip(convs, self.shp.final_layers)))
            if ok:
                w = torch.stack([c.weight.detach().float().reshape(32, 32) for c in convs]).contiguous()
                b = torch.stack([c.bias.detach().float() if c.bias is not None else torch.zeros(32, device=w.device) for c in convs]).contiguous()
                self._fold_cache[key] = (w, b)
            else:
                self._fold_cache[key] = None
        return self._fold_cache[key]

    def _folded_after_bias(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, x1_bias: Tensor):
        """Scale / shift of ``conv`` + ``bn`` when the LAST len(x1_bias) input channels arrive without the per-channel
        constant ``x1_bias`` their producer (a ConvTranspose2d) should have added: a constant image stays constant under
        reflect padding, so the convolution of that constant is the per-output constant sum_{c,taps} w[o][c][tap] * b[c],
        which goes into the shift -- one whole read + write pass over the upsampled activations less per Up block."""
        key = (id(conv), id(bn), "after_bias")
        if key not in self._fold_cache:
            scale, shift = self._folded(conv, bn)
            w = conv.weight.detach().float()
            const = (w[:, w.shape[1] - x1_bias.numel():].sum(dim=(2, 3)) @ x1_bias.float())
            self._fold_cache[key] = (scale, (shift + scale * const).contiguous())
        return self._fold_cache[key]

    def _double_conv_nhwc(self, dc: DoubleConv, x0: Tensor, x1: Optional[Tensor] = None, pool: bool = False,
                          x1_bias: Optional[Tensor] = None) -> Tensor:
        seq = dc.double_conv
        s1, t1 = self._folded(seq[0], seq[1]) if x1_bias is None else self._folded_after_bias(seq[0], seq[1], x1_bias)
        s2, t2 = self._folded(seq[3], seq[4])
        # The 32-channel, full-resolution level in float32: the hand-written MFMA convolution (csrc/mpp_conv.hip) takes the
        # unpadded activations (reflect padding is index arithmetic), the concat as a second source, the producer's
        # BatchNorm + ReLU at the load and its own in the epilogue -- no glue pass around it.
        c32 = (self.mfma_conv and self.dtype == torch.float32 and not pool and seq[0].out_channels == 32 and seq[3].in_channels == 32
               and seq[3].out_channels == 32)
        if c32 and seq[0].in_channels in (32, 64) and x0.shape[1] == 32 and (x1 is None) == (seq[0].in_channels == 32):
            h = self.ctx.conv3x3_c32(x0, self._packed(seq[0]), x1=x1, out_scale=s1, out_shift=t1)
            return self.ctx.conv3x3_c32(h, self._packed(seq[3]), out_scale=s2, out_shift=t2)
        if c32 and x1 is None and seq[0].in_channels == 3 and x0.shape[1] == 3 and x0.dtype == torch.float32:
            # the stem (3 -> 32): HBM-bound, one pass with its BatchNorm + ReLU (csrc/mpp_conv.hip: k_conv3x3_stem)
            key = (id(seq[0]), "stem")
            if key not in self._fold_cache:
                self._fold_cache[key] = (seq[0].weight.detach().float().permute(2, 3, 1, 0).reshape(9, 3, 32).contiguous(),)
            h = self.ctx.conv3x3_stem(x0, self._fold_cache[key][0], s1, t1)
            return self.ctx.conv3x3_c32(h, self._packed(seq[3]), out_scale=s2, out_shift=t2)
        if c32 and x1 is None:                    # the stem (3 -> 32) by the library, its BatchNorm + ReLU at the next load
            y = self.ctx.nhwc_glue(x0, None, pad=1, pool=False, out_dtype=self.dtype)
            r = self._cl(F.conv2d(y, self._weights(seq[0])[0], None))
            return self.ctx.conv3x3_c32(r, self._packed(seq[3]), in_scale=s1, in_shift=t1, out_scale=s2, out_shift=t2)
        y = self.ctx.nhwc_glue(x0, x1, pad=1, pool=pool, out_dtype=self.dtype)
        r = self._cl(F.conv2d(y, self._weights(seq[0])[0], None))
        y = self.ctx.nhwc_glue(r, pad=1, scale=s1, shift=t1)
        r = self._cl(F.conv2d(y, self._weights(seq[3])[0], None))
        return self.ctx.nhwc_glue(r, pad=0, scale=s2, shift=t2, out=r)

    def _backbone_nhwc(self, net: Unet, x: Tensor) -> Tensor:
        skips = []
        for i, down in enumerate(net.descending_path):
            x = self._double_conv_nhwc(down if i == 0 else down.maxpool_conv[1], x, pool=i > 0)
            skips.append(x)
        for up, skip in zip(net.ascending_path, skips[::-1][1:]):
            w, b = self._weights(up.up)
            # (the transposed convolution's bias is folded into the next convolution's shift: _folded_after_bias)
            x = self._double_conv_nhwc(up.conv, skip, self._cl(F.conv_transpose2d(x, w, None, stride=2)), x1_bias=b)
        return x

    def _head(self, conv: nn.Conv2d, h: Tensor) -> Tensor:
        w, b = self._weights(conv)
        return F.conv2d(h, w, b)

    def _backbone(self, net: Unet, x: Tensor) -> Tensor:
        skips = []
        for i, down in enumerate(net.descending_path):
            x = self._double_conv(down, x) if i == 0 else self._double_conv(down.maxpool_conv[1], down.maxpool_conv[0](x))
            skips.append(x)
        for up, skip in zip(net.ascending_path, skips[::-1][1:]):
            x = self._double_conv(up.conv, torch.cat([skip, up.up(x).to(skip.dtype)], dim=1))
        return x

    @torch.no_grad()
    def infer(self, image) -> Tuple[Tensor, List[Tensor]]:
        """image: [H,W,3] float in [0,1] (numpy or tensor) -> det [H,W] f32, marks 3 x [H,W,32] f32, on the GPU."""
        img = torch.as_tensor(np.asarray(image) if not torch.is_tensor(image) else image)
        img = img[..., :3].permute(2, 0, 1).float().to(self.device)
        H, W = img.shape[1:]
        padded, _ = pad_before_infer(img, self.pos.backbone.depth)
        x = padded.unsqueeze(0).contiguous()
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        big = padded.shape[1] * padded.shape[2] >= self.min_fused_pixels
        if self.layout == "nhwc" and self.fused and big and padded.shape[1] >= 16 and padded.shape[2] >= 16:
            xi = padded.permute(1, 2, 0).contiguous().unsqueeze(0).permute(0, 3, 1, 2)      # [1,3,H,W] over NHWC memory
            det = torch.empty((H, W), dtype=torch.float32, device=self.device)
            marks = [torch.empty((H, W, 32), dtype=torch.float32, device=self.device) for _ in range(3)]
            cur = torch.cuda.current_stream(self.device)

            def pos_part():
                pos_out = self._cl(self._head(self.pos.final_layer, self._backbone_nhwc(self.pos.backbone, xi)))
                self.ctx.posnet_epilogue_nhwc(pos_out, H, W, self.div_w, self.div_b, det)
                return pos_out

            def shp_part():
                h = self._backbone_nhwc(self.shp.backbone, xi)
                heads = self._packed_heads()
                if heads is not None and h.dtype == torch.float32:
                    # the three 1x1 heads, their biases and the softmax in ONE pass over h (csrc/mpp_conv.hip): 8.6 GB of
                    # traffic on a 4096 x 4096 image instead of 38 GB
                    h = self._cl(h)
                    self.ctx.shapenet_heads(h, heads[0], heads[1], H, W, marks)
                    return h
                logits = [self._cl(self._head(fl[0], h)) for fl in self.shp.final_layers]
                for k in range(3):
                    self.ctx.shapenet_epilogue_nhwc(logits[k], H, W, marks[k])
                return logits

            if self.two_streams:
                # the two networks are independent until the chains read both maps: PosNet on one stream, ShapeNet on
                # another -- the tail of one network's kernel overlaps the head of the other's, launch gaps disappear
                if self._streams is None:
                    self._streams = (torch.cuda.Stream(self.device), torch.cuda.Stream(self.device))
                keep = []
                for st, part in zip(self._streams, (pos_part, shp_part)):
                    st.wait_stream(cur)
                    with torch.cuda.stream(st):
                        self.ctx.set_stream(st.cuda_stream)
                        keep.append(part())
                for st in self._streams:
                    cur.wait_stream(st)
                self.ctx.set_stream(cur.cuda_stream)
                for t in [det] + marks:
                    for st in self._streams:
                        t.record_stream(st)
                self._keep = tuple(keep)
                return det, marks
            self._keep = (pos_part(), shp_part())    # alive until the kernels on this stream have consumed them
            return det, marks
        with torch.autocast("cuda", dtype=self.dtype, enabled=self.dtype != torch.float32):
            # (small images are launch-bound: there the extra host calls of the fused path cost more than its two
            # saved passes per convolution bring -- 512x512: 3.5 ms unfused vs 4.4 ms fused; 2048x2048: 45 vs 41 ms)
            if self.fused and big:
                pos_out = self.pos.final_layer(self._backbone(self.pos.backbone, x.float()))
                h = self._backbone(self.shp.backbone, x.float())
                logits = [fl(h) for fl in self.shp.final_layers]
            else:
                pos_out = self.pos(x)
                logits = self.shp(x)
        return self._epilogues(pos_out, logits, H, W)

    @torch.no_grad()
    def infer_region(self, image, region) -> Tuple[Tensor, List[Tensor]]:
        """Score maps of the image region ``(x0, x1, y0, y1)`` only: the nets run on the region plus ``HALO`` pixels
        (origin rounded down to a multiple of 2**depth so that pooling and the bottom/right zero padding of
        ``pad_before_infer`` fall as they do on the whole image, crop clamped to the image so that image borders
        keep their reflect padding / one-sided differences); everything an interior crop border can influence lies
        inside the halo, which is cut off.  Up to the convolution library's choice of algorithm per shape the result
        equals ``infer(image)[region]`` -- this is how a rank of a multi-GPU run gets the maps of its own tiles
         without a forward over the whole image."""
        x0, x1, y0, y1 = (int(v) for v in region)
        cx0, cx1, cy0, cy1 = halo_crop(region, tuple(image.shape[:2]), align=2 ** self.pos.backbone.depth)
        det, marks = self.infer(image[cx0:cx1, cy0:cy1])
        sl = (slice(x0 - cx0, x1 - cx0), slice(y0 - cy0, y1 - cy0))
        return det[sl], [m[sl] for m in marks]

    def _epilogues(self, pos_out: Tensor, logits: List[Tensor], H: int, W: int) -> Tuple[Tensor, List[Tensor]]:
        pos_out = pos_out[0].float().contiguous()
        logits = [t[0].float().contiguous() for t in logits]
        det = torch.empty((H, W), dtype=torch.float32, device=self.device)
        marks = [torch.empty((H, W, 32), dtype=torch.float32, device=self.device) for _ in range(3)]
        self.ctx.set_stream(torch.cuda.current_stream(self.device).cuda_stream)
        self.ctx.posnet_epilogue(pos_out, H, W, self.div_w, self.div_b, det)
        for k in range(3):
            self.ctx.shapenet_epilogue(logits[k], H, W, marks[k])
        self._keep = (pos_out, logits)        # alive until the kernels on this stream have consumed them
        return det, marks
