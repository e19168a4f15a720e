"""MPViT encoder (MonoViT's depth encoder; BASELINE configs[4]) with the reference's interface.

Same public names, constructor arguments, forward signature and state-dict keys as the reference's
`networksvit/mpvit.py` (`MPViT` :602-737, the `mpvit_tiny/xsmall/small/base` factories :756-846), so that
`encoder.pth` files written by either side load in the other - including the alias keys the reference's
module sharing creates (one `ConvPosEnc` / `ConvRelPosEnc` per path is registered under the path AND under
every block of it, mpvit.py:449-465).  The body is this build's own:

* tokens live in ONE layout per path, `[B, N, C]` row-major, which is at the same time the NHWC
  (channels-last) image `[B, H, W, C]`: LayerNorm / Linear GEMMs read it directly and the depth-wise
  convolutions (position encodings) see a zero-copy channels-last view - the reference's
  `transpose().contiguous().view()` / einops `rearrange` round trips (4 per block) are gone;
* factorised attention (`softmax_N(k)^T v`, then `q @ .`, mpvit.py:372-381) is evaluated per head with
  batched GEMMs on `[B, h, N, Ch]` views; the per-head scale is folded into the small `[Ch, Ch]` matrix;
* timm / mmcv / mmseg are not dependencies: stochastic depth, truncated-normal initialisation and the
  BatchNorm factory are restated here (`mpvit.py:20-32` only uses those four entry points).
"""
import math
from functools import partial

import torch
import torch.nn as nn

from .. import ops

__all__ = ["MPViT", "mpvit_tiny", "mpvit_xsmall", "mpvit_small", "mpvit_base"]

_BN = dict(type="BN")


def _norm2d(norm_cfg, channels):
    """mmcv `build_norm_layer(norm_cfg, c)[1]` for the only type the reference uses."""
    if norm_cfg.get("type", "BN") != "BN":
        raise ValueError("only BatchNorm2d ('BN') is used by MonoViT")
    bn = nn.BatchNorm2d(channels, eps=norm_cfg.get("eps", 1e-5))
    for p in bn.parameters():
        p.requires_grad = norm_cfg.get("requires_grad", True)
    return bn


def _he_fan_out(conv, per_group=False):
    fan_out = conv.kernel_size[0] * conv.kernel_size[1] * conv.out_channels
    if per_group:
        fan_out //= conv.groups
    conv.weight.data.normal_(0.0, math.sqrt(2.0 / fan_out))
    if conv.bias is not None:
        conv.bias.data.zero_()


class _DropPathBank:
    """All stochastic-depth draws of one encoder forward from ONE uniform draw (GPU training only).  The encoder makes 76
    draws of [B] per forward (two per block); as separate `bernoulli_` + `div_` calls that is ~150 tiny launches.  The
    first GPU forward records the sequence of rates while drawing per call; later forwards draw a [draws, B] matrix up
    front and hand out its rows in call order.  Same distribution per draw; CPU runs keep the per-call draws (whose
    order on the CPU generator is what the reference fixtures pin)."""

    def __init__(self):
        self.rates, self.keep, self.masks, self.at, self.recording = None, None, None, 0, None

    def begin(self, x):
        self.masks, self.at, self.recording = None, 0, None
        if not (x.is_cuda and ops.FUSED_TOKEN_GLUE):
            return
        if self.rates is None:
            self.recording = []
            return
        if self.keep is None or self.keep.device != x.device:
            self.keep = torch.tensor([1.0 - r for r in self.rates], device=x.device).view(-1, 1)
        u = torch.rand(len(self.rates), x.shape[0], device=x.device)
        self.masks = (u < self.keep).float() / self.keep

    def end(self):
        if self.recording is not None:
            self.rates, self.recording = list(self.recording), None
        self.masks = None

    def draw(self, rate, batch):
        """Row of the pre-drawn matrix for this call, or None (caller draws itself)."""
        if self.recording is not None:
            self.recording.append(rate)
            return None
        if self.masks is None or self.at >= len(self.rates) or self.rates[self.at] != rate or self.masks.shape[1] != batch:
            self.masks = None          # call pattern changed (a sub-module run on its own): per-call draws from here on
            return None
        self.at += 1
        return self.masks[self.at - 1]


class DropPath(nn.Module):
    """Stochastic depth per sample (timm 0.6.12 `DropPath`): a Bernoulli(keep) mask over the batch
    dimension, divided by keep; identity in eval mode."""

    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob, self.scale_by_keep = float(drop_prob), scale_by_keep
        self.bank = None           # set by MPViT: pre-drawn masks on the GPU

    def scale(self, x):
        """The per-sample factor of this call ([B]: 0 or 1/keep), or None when the layer is the identity - the same
        draw `forward` makes, for callers that fold the multiply into their own kernel."""
        if self.drop_prob == 0.0 or not self.training:
            return None
        keep = 1.0 - self.drop_prob
        if self.bank is not None and keep > 0.0 and self.scale_by_keep:
            mask = self.bank.draw(self.drop_prob, x.shape[0])
            if mask is not None:
                return mask
        mask = x.new_empty((x.shape[0],)).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return mask

    def forward(self, x):
        mask = self.scale(x)
        return x if mask is None else x * mask.view((x.shape[0],) + (1,) * (x.ndim - 1))


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.0):
        super().__init__()
        self.fc1 = nn.Linear(in_features, hidden_features or in_features)
        self.act = act_layer()
        self.fc2 = nn.Linear(hidden_features or in_features, out_features or in_features)
        self.drop = nn.Dropout(drop)

    def forward(self, x):
        if _hip_tokens(x):          # bias gradients by the column-sum kernel (csrc/bbd_tokens.hip)
            return self.drop(ops.linear_tokens(self.drop(self.act(ops.linear_tokens(x, self.fc1))), self.fc2))
        return self.drop(self.fc2(self.drop(self.act(self.fc1(x)))))


class Conv2d_BN(nn.Module):
    """conv (no bias) -> BatchNorm -> optional activation (mpvit.py:82-122)."""

    def __init__(self, in_ch, out_ch, kernel_size=1, stride=1, pad=0, dilation=1, groups=1, bn_weight_init=1,
                 act_layer=None, norm_cfg=_BN):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, pad, dilation, groups, bias=False)
        self.bn = _norm2d(norm_cfg, out_ch)
        nn.init.constant_(self.bn.weight, bn_weight_init)
        nn.init.constant_(self.bn.bias, 0)
        _he_fan_out(self.conv)
        self.act_layer = act_layer() if act_layer is not None else nn.Identity()

    def forward(self, x):
        return self.act_layer(self.bn(self.conv(x)))


class DWConv2d_BN(nn.Module):
    """depth-wise k x k -> point-wise 1x1 -> BatchNorm -> activation (mpvit.py:125-174)."""

    def __init__(self, in_ch, out_ch, kernel_size=1, stride=1, norm_layer=nn.BatchNorm2d, act_layer=nn.Hardswish,
                 bn_weight_init=1, norm_cfg=_BN):
        super().__init__()
        self.dwconv = nn.Conv2d(in_ch, out_ch, kernel_size, stride, (kernel_size - 1) // 2, groups=out_ch, bias=False)
        self.pwconv = nn.Conv2d(out_ch, out_ch, 1, 1, 0, bias=False)
        self.bn = _norm2d(norm_cfg, out_ch)
        self.act = act_layer() if act_layer is not None else nn.Identity()
        _he_fan_out(self.dwconv)
        _he_fan_out(self.pwconv)
        self.bn.weight.data.fill_(bn_weight_init)
        self.bn.bias.data.zero_()

    def forward(self, x):
        return self.act(self.bn(self.pwconv(self.dwconv(x))))


class DWCPatchEmbed(nn.Module):
    def __init__(self, in_chans=3, embed_dim=768, patch_size=16, stride=1, pad=0, act_layer=nn.Hardswish, norm_cfg=_BN):
        super().__init__()
        self.patch_conv = DWConv2d_BN(in_chans, embed_dim, kernel_size=patch_size, stride=stride,
                                      act_layer=nn.Hardswish, norm_cfg=norm_cfg)

    def forward(self, x):
        return self.patch_conv(x)


class Patch_Embed_stage(nn.Module):
    """A chain of `num_path` patch embeddings; path p attends over the output of embedding p, so the
    paths see receptive fields 3, 5, 7 (mpvit.py:208-237).  Only the first one strides."""

    def __init__(self, embed_dim, num_path=4, isPool=False, norm_cfg=_BN):
        super().__init__()
        self.patch_embeds = nn.ModuleList([
            DWCPatchEmbed(embed_dim, embed_dim, patch_size=3, stride=2 if (isPool and p == 0) else 1, pad=1,
                          norm_cfg=norm_cfg) for p in range(num_path)])

    def forward(self, x):
        outs = []
        for embed in self.patch_embeds:
            x = embed(x)
            outs.append(x)
        return outs


def _hip_tokens(t):
    """fp32 GPU tokens take the HIP depth-wise kernels (BBD_FUSED_NN=0 keeps the MIOpen path for A/B runs)."""
    return t.is_cuda and t.dtype == torch.float32 and ops.FUSED_NN


def _as_image(tokens, size):
    """[B, N, C] tokens -> [B, C, H, W] view in channels-last memory format (no copy)."""
    B, N, C = tokens.shape
    return tokens.view(B, size[0], size[1], C).permute(0, 3, 1, 2)


def _as_tokens(image):
    """[B, C, H, W] (any memory format) -> [B, N, C] row-major tokens (no copy if channels-last)."""
    B, C, H, W = image.shape
    return image.permute(0, 2, 3, 1).reshape(B, H * W, C)


class ConvPosEnc(nn.Module):
    """x + depth-wise 3x3 conv of x laid out as an image (mpvit.py:240-259)."""

    def __init__(self, dim, k=3):
        super().__init__()
        self.proj = nn.Conv2d(dim, dim, k, 1, k // 2, groups=dim)

    def forward(self, x, size, share=None):
        if _hip_tokens(x):           # one HIP launch, residual folded in (csrc/bbd_vit.hip)
            return ops.dwconv_tokens(x, size, [self.proj], add_input=True, share=share)
        img = _as_image(x, size)
        return _as_tokens(self.proj(img) + img)


class ConvRelPosEnc(nn.Module):
    """q * depth-wise conv(v): head groups get window sizes 3 / 5 / 7 (mpvit.py:262-330).
    `window` is an int (all heads) or {window size: number of heads}."""

    def __init__(self, Ch, h, window):
        super().__init__()
        if isinstance(window, int):
            window = {window: h}
        elif not isinstance(window, dict):
            raise ValueError()
        self.window = window
        self.conv_list = nn.ModuleList()
        self.head_splits = []
        for size, heads in window.items():
            self.conv_list.append(nn.Conv2d(heads * Ch, heads * Ch, kernel_size=(size, size),
                                            padding=(size // 2, size // 2), groups=heads * Ch))
            self.head_splits.append(heads)
        self.channel_splits = [heads * Ch for heads in self.head_splits]

    def conv_v(self, v_tokens, size):
        """v as `[B, N, h*Ch]` tokens (head-major channels) -> conv(v) in the same layout."""
        if _hip_tokens(v_tokens):    # the three window sizes write their channel groups of one output
            return ops.dwconv_tokens(v_tokens, size, list(self.conv_list))
        img = _as_image(v_tokens, size)
        parts = torch.split(img, self.channel_splits, dim=1)
        return _as_tokens(torch.cat([conv(p) for conv, p in zip(self.conv_list, parts)], dim=1))

    def forward(self, q, v, size):
        """Reference signature: q, v `[B, h, N, Ch]` -> `[B, h, N, Ch]`."""
        B, h, N, Ch = q.shape
        conv = self.conv_v(v.transpose(1, 2).reshape(B, N, h * Ch), size)
        return q * conv.view(B, N, h, Ch).transpose(1, 2)


class FactorAtt_ConvRelPosEnc(nn.Module):
    """Factorised attention + convolutional relative position encoding (mpvit.py:333-394):
    out = scale * q (softmax_N(k)^T v) + q * conv(v), then the output projection."""

    def __init__(self, dim, num_heads=8, qkv_bias=False, qk_scale=None, attn_drop=0.0, proj_drop=0.0, shared_crpe=None):
        super().__init__()
        self.num_heads = num_heads
        self.scale = qk_scale or (dim // num_heads) ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.attn_drop = nn.Dropout(attn_drop)       # unused by the reference as well
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(proj_drop)
        self.crpe = shared_crpe

    def forward(self, x, size, share=None):
        B, N, C = x.shape
        h = self.num_heads
        hip = _hip_tokens(x)
        qkv = ops.linear_tokens(x, self.qkv) if hip else self.qkv(x)   # [B, N, 3C] = q | k | v, head-major channels
        if _hip_tokens(qkv) and ops.factor_attention_supported(C, h):
            # two HIP ops on the packed activation: conv(v) read in place from the v third, then column
            # statistics of k + [Ch x Ch] contexts + the token-parallel output (csrc/bbd_vit.hip)
            if ops.FUSED_TOKEN_GLUE:     # one autograd node: the conv's data gradient lands in gqkv's v third in place
                out = ops.factor_attention_crpe(qkv, size, list(self.crpe.conv_list), h, self.scale, share=share)
            else:
                convv = self.crpe.conv_v(qkv[:, :, 2 * C:], size)
                out = ops.factor_attention(qkv, convv, h, self.scale)
            return self.proj_drop(ops.linear_tokens(out, self.proj))
        qkv = qkv.view(B, N, 3, h, C // h)
        q, k, v = qkv[:, :, 0], qkv[:, :, 1], qkv[:, :, 2]             # [B, N, h, Ch] strided views
        # softmax over the N tokens, then the [Ch, Ch] context of every head: (k^T v) is tiny, so the
        # scale goes there instead of onto the [N, Ch] product
        ctx = torch.einsum("bnhk,bnhv->bhkv", k.softmax(dim=1), v) * self.scale
        att = torch.einsum("bnhk,bhkv->bnhv", q, ctx)
        crpe = self.crpe.conv_v(v.reshape(B, N, C), size).view(B, N, h, C // h)
        out = (att + q * crpe).reshape(B, N, C)
        return self.proj_drop(self.proj(out))


class MHCABlock(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=3, drop_path=0.0, qkv_bias=True, qk_scale=None,
                 norm_layer=partial(nn.LayerNorm, eps=1e-6), shared_cpe=None, shared_crpe=None):
        super().__init__()
        self.cpe = shared_cpe
        self.crpe = shared_crpe
        self.factoratt_crpe = FactorAtt_ConvRelPosEnc(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                                      shared_crpe=shared_crpe)
        self.mlp = Mlp(in_features=dim, hidden_features=dim * mlp_ratio)
        self.drop_path = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm1 = norm_layer(dim)
        self.norm2 = norm_layer(dim)

    def _drop_scale(self, x):
        return self.drop_path.scale(x) if isinstance(self.drop_path, DropPath) else None

    def forward(self, x, size, share=None):
        """`share` = (cpe sink, crpe sink, layer index, layer count) from the MHCAEncoder whose blocks share the position
        encodings: their weight gradients are summed inside the HIP launches (ops.SharedGrads)."""
        if self.cpe is not None:
            x = self.cpe(x, size, (share[0], share[2], share[3])) if share is not None else self.cpe(x, size)
        if _hip_tokens(x) and ops.FUSED_TOKEN_GLUE and ops.token_glue_supported(x):
            # residual + stochastic depth + the NEXT LayerNorm as one pass each way (csrc/bbd_tokens.hip); the two
            # stochastic-depth draws happen in the reference's order
            # (x comes back as an output of the LayerNorm node: the residual's gradient meets LayerNorm's inside the kernel)
            x, z = ops.layernorm_tokens(x, self.norm1, passthrough=True)
            att = self.factoratt_crpe(z, size, (share[1], share[2], share[3]) if share is not None else None)
            x, z = ops.residual_layernorm(x, att, self._drop_scale(x), self.norm2)
            return ops.residual_add(x, self.mlp(z), self._drop_scale(x))
        x = x + self.drop_path(self.factoratt_crpe(self.norm1(x), size))
        return x + self.drop_path(self.mlp(self.norm2(x)))


class MHCAEncoder(nn.Module):
    """One path of a stage: `num_layers` blocks sharing one pair of position encodings."""

    def __init__(self, dim, num_layers=1, num_heads=8, mlp_ratio=3, drop_path_list=[], qk_scale=None,
                 crpe_window={3: 2, 5: 3, 7: 3}):
        super().__init__()
        self.num_layers = num_layers
        self.cpe = ConvPosEnc(dim, k=3)
        self.crpe = ConvRelPosEnc(Ch=dim // num_heads, h=num_heads, window=crpe_window)
        self.MHCA_layers = nn.ModuleList([
            MHCABlock(dim, num_heads=num_heads, mlp_ratio=mlp_ratio, drop_path=drop_path_list[i], qk_scale=qk_scale,
                      shared_cpe=self.cpe, shared_crpe=self.crpe) for i in range(num_layers)])

    def forward(self, x, size):
        """[B, N, C] tokens -> [B, C, H, W] feature map (channels-last memory, zero-copy)."""
        fused = _hip_tokens(x) and ops.FUSED_TOKEN_GLUE and ops.token_glue_supported(x) and torch.is_grad_enabled()
        n = len(self.MHCA_layers)
        # gradient sinks of the shared position encodings: one pair per forward pass (they live in the layers' autograd
        # nodes), so passes never share a buffer
        cpe_grads, crpe_grads = ops.SharedGrads(), ops.SharedGrads()
        for i, layer in enumerate(self.MHCA_layers):
            x = layer(x, size, (cpe_grads, crpe_grads, i, n) if (fused and n > 1) else None)
        return _as_image(x, size)


class ResBlock(nn.Module):
    """The convolutional (local) path of a stage: 1x1 -> depth-wise 3x3 -> 1x1, residual (mpvit.py:483-531)."""

    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.Hardswish, norm_cfg=_BN):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.conv1 = Conv2d_BN(in_features, hidden_features, act_layer=act_layer, norm_cfg=norm_cfg)
        self.dwconv = nn.Conv2d(hidden_features, hidden_features, 3, 1, 1, bias=False, groups=hidden_features)
        self.norm = _norm2d(norm_cfg, hidden_features)
        self.act = act_layer()
        self.conv2 = Conv2d_BN(hidden_features, out_features, norm_cfg=norm_cfg)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                _he_fan_out(m, per_group=True)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def forward(self, x):
        return x + self.conv2(self.act(self.norm(self.dwconv(self.conv1(x)))))


class MHCA_stage(nn.Module):
    """Local conv path + `num_path` attention paths, concatenated and mixed by a 1x1 conv (mpvit.py:534-583)."""

    def __init__(self, embed_dim, out_embed_dim, num_layers=1, num_heads=8, mlp_ratio=3, num_path=4, norm_cfg=_BN,
                 drop_path_list=[]):
        super().__init__()
        self.mhca_blks = nn.ModuleList([
            MHCAEncoder(embed_dim, num_layers, num_heads, mlp_ratio, drop_path_list=drop_path_list)
            for _ in range(num_path)])
        self.InvRes = ResBlock(in_features=embed_dim, out_features=embed_dim, norm_cfg=norm_cfg)
        self.aggregate = Conv2d_BN(embed_dim * (num_path + 1), out_embed_dim, act_layer=nn.Hardswish, norm_cfg=norm_cfg)

    def forward(self, inputs):
        outs = [self.InvRes(inputs[0])]
        for x, path in zip(inputs, self.mhca_blks):
            outs.append(path(_as_tokens(x), size=x.shape[2:]))
        return self.aggregate(torch.cat(outs, dim=1)), outs


def dpr_generator(drop_path_rate, num_layers, num_stages):
    """Linearly increasing stochastic-depth rates, split per stage."""
    rates = [r.item() for r in torch.linspace(0, drop_path_rate, sum(num_layers))]
    out, at = [], 0
    for i in range(num_stages):
        out.append(rates[at:at + num_layers[i]])
        at += num_layers[i]
    return out


class MPViT(nn.Module):
    """Multi-Path ViT backbone; `forward(img)` -> 5 feature maps at strides 2, 4, 8, 16, 32."""

    def __init__(self, num_classes=80, in_chans=3, num_stages=4, num_layers=[1, 1, 1, 1], mlp_ratios=[8, 8, 4, 4],
                 num_path=[4, 4, 4, 4], embed_dims=[64, 128, 256, 512], num_heads=[8, 8, 8, 8], drop_path_rate=0.2,
                 norm_cfg=_BN, norm_eval=False, pretrained=None):
        super().__init__()
        self.num_classes, self.num_stages = num_classes, num_stages
        self.conv_norm_cfg, self.norm_eval = norm_cfg, norm_eval
        dpr = dpr_generator(drop_path_rate, num_layers, num_stages)
        self.stem = nn.Sequential(
            Conv2d_BN(in_chans, embed_dims[0] // 2, kernel_size=3, stride=2, pad=1, act_layer=nn.Hardswish, norm_cfg=norm_cfg),
            Conv2d_BN(embed_dims[0] // 2, embed_dims[0], kernel_size=3, stride=1, pad=1, act_layer=nn.Hardswish, norm_cfg=norm_cfg))
        self.patch_embed_stages = nn.ModuleList([
            Patch_Embed_stage(embed_dims[i], num_path=num_path[i], isPool=True, norm_cfg=norm_cfg) for i in range(num_stages)])
        self.mhca_stages = nn.ModuleList([
            MHCA_stage(embed_dims[i], embed_dims[i + 1] if i + 1 != num_stages else embed_dims[i], num_layers[i],
                       num_heads[i], mlp_ratios[i], num_path[i], norm_cfg=norm_cfg, drop_path_list=dpr[i])
            for i in range(num_stages)])
        self.num_ch_enc = [embed_dims[0]] + [embed_dims[min(i + 1, num_stages - 1)] for i in range(num_stages)]
        self._drop_bank = _DropPathBank()
        for m in self.modules():
            if isinstance(m, DropPath):
                m.bank = self._drop_bank
        if pretrained is not None:
            self.init_weights(pretrained)

    def init_weights(self, pretrained=None):
        """Truncated-normal Linear weights / unit LayerNorms (mpvit.py:684-708); `pretrained` = a
        checkpoint path whose `model` (or top-level) state dict is loaded non-strictly on top."""
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.trunc_normal_(m.weight, std=0.02, a=-2.0, b=2.0)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)
            elif isinstance(m, nn.LayerNorm):
                nn.init.constant_(m.bias, 0)
                nn.init.constant_(m.weight, 1.0)
        if isinstance(pretrained, str):
            load_pretrained(self, pretrained)
        elif pretrained is not None:
            raise TypeError("pretrained must be a str or None")

    def forward_features(self, x):
        if self.training:
            self._drop_bank.begin(x)
        x = self.stem(x)
        outs = [x]
        for embed, stage in zip(self.patch_embed_stages, self.mhca_stages):
            x, _ = stage(embed(x))
            outs.append(x)
        if self.training:
            self._drop_bank.end()
        return outs

    def forward(self, x):
        return self.forward_features(x)

    def train(self, mode=True):
        super().train(mode)
        if mode and self.norm_eval:
            for m in self.modules():
                if isinstance(m, nn.modules.batchnorm._BatchNorm):
                    m.eval()
        return self


def load_pretrained(model, path):
    """Non-strict load of an ImageNet MPViT checkpoint (`{'model': state_dict}` as published, or a bare
    state dict); classifier-head keys that MonoViT does not have are ignored, like mmcv's loader."""
    state = torch.load(path, map_location="cpu")
    state = state.get("model", state) if isinstance(state, dict) else state
    own = model.state_dict()
    model.load_state_dict({k: v for k, v in state.items() if k in own and v.shape == own[k].shape}, strict=False)
    return model


_VARIANTS = {
    "tiny": dict(num_layers=[1, 2, 4, 1], embed_dims=[64, 96, 176, 216], mlp_ratios=[2, 2, 2, 2]),
    "xsmall": dict(num_layers=[1, 2, 4, 1], embed_dims=[64, 128, 192, 256], mlp_ratios=[4, 4, 4, 4]),
    "small": dict(num_layers=[1, 3, 6, 3], embed_dims=[64, 128, 216, 288], mlp_ratios=[4, 4, 4, 4]),
    "base": dict(num_layers=[1, 3, 8, 3], embed_dims=[128, 224, 368, 480], mlp_ratios=[4, 4, 4, 4]),
}


def _build(variant, checkpoint, kwargs):
    import os
    model = MPViT(num_stages=4, num_path=[2, 3, 3, 3], num_heads=[8, 8, 8, 8], **_VARIANTS[variant], **kwargs)
    if checkpoint is not None and os.path.isfile(checkpoint):
        load_pretrained(model, checkpoint)
    return model


def mpvit_tiny(**kwargs):
    return _build("tiny", None, kwargs)


def mpvit_xsmall(checkpoint="./ckpt/mpvit_xsmall.pth", **kwargs):
    return _build("xsmall", checkpoint, kwargs)


def mpvit_small(checkpoint="./ckpt/mpvit_small.pth", **kwargs):
    """MonoViT's encoder: paths [2,3,3,3], layers [1,3,6,3], channels [64,128,216,288], MLP ratio 4.
    The reference loads `./ckpt/mpvit_small.pth` (ImageNet, git-ignored upstream) unconditionally
    (mpvit.py:815); here it is loaded when the file exists, otherwise the network keeps its initialisation."""
    return _build("small", checkpoint, kwargs)


def mpvit_base(**kwargs):
    return _build("base", None, kwargs)
