"""Convolutional backbones of the terrain encoder, re-stated in plain `torch.nn` (they run on MIOpen / rocBLAS, where the
MFMA units are used; no hand kernels here -- BASELINE.json north_star keeps the encoder backbone on the vendor libraries).

The reference gets these from third-party packages that are NOT part of /root/reference and are not installed here:
  * `efficientnet_pytorch==0.7.1`  `EfficientNet.from_pretrained("efficientnet-b0")` (lss.py:55), used through its
    private members `_conv_stem`, `_bn0`, `_swish`, `_blocks`, `_global_params.drop_connect_rate` (lss.py:73-94)
  * `torchvision.models.resnet.resnet18(zero_init_residual=True)` `layer1..3`, `bn1`, `relu` (lss.py:105-112)
They are restated from the published architectures with the SAME module / parameter names, so a reference checkpoint's
`state_dict` keys and shapes line up (`from_pretrained`, lss.py:293-302).  Numerical parity of these blocks is
"unpinned": neither the packages nor any weights are available offline; only names/shapes are tested.
"""
import math
import os
import types

import torch
from torch import nn
from torch.nn import functional as F


# ------------------------------------------------------------------------------------------------------------
# EfficientNet-B0 (Tan & Le 2019), efficientnet_pytorch naming
# ------------------------------------------------------------------------------------------------------------
class Swish(nn.Module):
    """x * sigmoid(x) (efficientnet_pytorch's `MemoryEfficientSwish`) as ONE fused kernel each way (`F.silu`): the
    backbone calls it ~50 times per forward, and written out it is two kernels forward and three backward."""

    def forward(self, x):
        return torch.nn.functional.silu(x)


def lean():
    """The launch diet below (same arithmetic, fewer / better kernels) is on unless MF_BACKBONE_LEAN=0 -- kept switchable for
    the A/B in profiles/ (tools/ab_c4_backbone.sh)."""
    return os.environ.get('MF_BACKBONE_LEAN', '1') != '0'


class _DepthwiseNative(torch.autograd.Function):
    """Depthwise convolution through ATen's own depthwise kernels.  MIOpen's immediate mode has no tuned solver for these
    shapes on gfx950 without a find-db and falls back to `naive_conv_*` (8.9 % of the encoder step's kernel time,
    profiles/r4e_c4_kernel_stats.csv); the backend is chosen again in the backward, hence a Function and not just a
    context manager around the forward."""

    # (under autocast the forward runs in the autocast dtype and the backward sees the same dtypes: without these decorators the
    #  saved half-precision input met a float32 weight in `convolution_backward`.  ADVICE r4.)
    @staticmethod
    @torch.amp.custom_fwd(device_type='cuda')
    def forward(ctx, x, w, stride, padding):
        if w.dtype != x.dtype:
            w = w.to(x.dtype)
        ctx.save_for_backward(x, w)
        ctx.conf = (stride, padding)
        with torch.backends.cudnn.flags(enabled=False):
            return F.conv2d(x, w, None, stride, padding, 1, w.shape[0])

    @staticmethod
    @torch.amp.custom_bwd(device_type='cuda')
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        stride, padding = ctx.conf
        g = g.to(x.dtype)
        with torch.backends.cudnn.flags(enabled=False):
            gx, gw, _ = torch.ops.aten.convolution_backward(g.contiguous(), x, w, None, list(stride), list(padding), [1, 1], False, [0, 0],
                                                            w.shape[0], [ctx.needs_input_grad[0], ctx.needs_input_grad[1], False])
        return gx, gw, None, None


class Conv2dStaticSame(nn.Conv2d):
    """Conv with TensorFlow-style "same" padding fixed at construction for even feature maps: total pad k - stride,
    the extra pixel on the bottom/right (k3 s2 -> (0,1), k5 s2 -> (1,2), stride 1 -> symmetric).  A symmetric pad is
    handed to the convolution itself (no pad kernel, no slice in the backward); `static_padding` stays as a member
    because the package the reference uses has it."""

    def __init__(self, cin, cout, k, stride=1, groups=1, bias=False):
        super().__init__(cin, cout, k, stride=stride, groups=groups, bias=bias)
        total = max(k - stride, 0)
        lo = total // 2
        self.static_padding = nn.ZeroPad2d((lo, total - lo, lo, total - lo)) if total > 0 else nn.Identity()
        self._sym = lo if total == 2 * lo else None

    def forward(self, x):
        fast = lean()
        pad = 0
        if fast and self._sym is not None:
            pad = self._sym
        else:
            x = self.static_padding(x)
        if fast and x.is_cuda and self.groups > 1 and self.groups == self.in_channels == self.out_channels and self.bias is None:
            return _DepthwiseNative.apply(x, self.weight, tuple(self.stride), (pad, pad))
        return F.conv2d(x, self.weight, self.bias, self.stride, pad, self.dilation, self.groups)


def drop_connect(x, p, training):
    if not training or not p:
        return x
    keep = 1.0 - p
    mask = torch.floor(keep + torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device))
    return x / keep * mask


def drop_connect_add(x, inputs, p, training):
    """`drop_connect(x) + inputs` as one elementwise kernel each way: the same uniform draw, floor(keep + r) == 1 iff r >= p."""
    if not training or not p:
        return x + inputs
    keep = 1.0 - p
    scale = torch.floor(keep + torch.rand(x.shape[0], 1, 1, 1, dtype=x.dtype, device=x.device)) / keep
    return torch.addcmul(inputs, x, scale)


class MBConvBlock(nn.Module):
    """Mobile inverted bottleneck with squeeze-excitation (expand 1x1 -> depthwise kxk -> SE -> project 1x1)."""

    def __init__(self, cin, cout, k, stride, expand, se_ratio=0.25, bn_mom=0.01, bn_eps=1e-3):
        super().__init__()
        self.stride, self.cin, self.cout, self.expand = stride, cin, cout, expand
        mid = cin * expand
        if expand != 1:
            self._expand_conv = Conv2dStaticSame(cin, mid, 1)
            self._bn0 = nn.BatchNorm2d(mid, momentum=bn_mom, eps=bn_eps)
        self._depthwise_conv = Conv2dStaticSame(mid, mid, k, stride=stride, groups=mid)
        self._bn1 = nn.BatchNorm2d(mid, momentum=bn_mom, eps=bn_eps)
        sq = max(1, int(cin * se_ratio))
        self._se_reduce = Conv2dStaticSame(mid, sq, 1, bias=True)
        self._se_expand = Conv2dStaticSame(sq, mid, 1, bias=True)
        self._project_conv = Conv2dStaticSame(mid, cout, 1)
        self._bn2 = nn.BatchNorm2d(cout, momentum=bn_mom, eps=bn_eps)
        self._swish = Swish()

    def forward(self, inputs, drop_connect_rate=None):
        x = inputs
        if self.expand != 1:
            x = self._swish(self._bn0(self._expand_conv(x)))
        x = self._swish(self._bn1(self._depthwise_conv(x)))
        if lean():
            # squeeze-excitation on the pooled (B, C) rows: the two 1x1 convolutions of a 1x1 map are two small GEMMs
            s = x.mean((2, 3))
            s = F.linear(F.silu(F.linear(s, self._se_reduce.weight.flatten(1), self._se_reduce.bias)),
                         self._se_expand.weight.flatten(1), self._se_expand.bias)
            x = torch.sigmoid(s)[:, :, None, None] * x
        else:
            s = F.adaptive_avg_pool2d(x, 1)
            s = self._se_expand(self._swish(self._se_reduce(s)))
            x = torch.sigmoid(s) * x
        x = self._bn2(self._project_conv(x))
        if self.stride == 1 and self.cin == self.cout:
            if lean():
                return drop_connect_add(x, inputs, drop_connect_rate, self.training)
            x = drop_connect(x, drop_connect_rate, self.training) + inputs
        return x


class BatchNorm2dCounted(nn.BatchNorm2d):
    """BatchNorm2d whose `num_batches_tracked` is bumped by its owner's `_CounterBank` (one kernel for all layers) instead of by
    its own forward.  A class, not a per-instance `forward` patch: `torch.save(model)` pickles it, `copy.deepcopy` keeps it, and
    `SyncBatchNorm.convert_sync_batchnorm` replaces it like any other batch norm (the converted layers count for themselves)."""

    def forward(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, self.training or self.running_mean is None,
                            self.momentum, self.eps)


class _CounterBank:
    """All `num_batches_tracked` buffers of a model as views of ONE int64 row, bumped by one kernel per training forward
    instead of one per batch-norm layer (70 launches per encoder step).  `.to()` / `.cuda()` re-create the buffers one by
    one: the bank notices (device or base changed) and gathers them again, outside any stream capture.
    A tracked layer that is no longer its parent's child -- `SyncBatchNorm.convert_sync_batchnorm` replaced it; the replacement shares
    the old buffer tensor and counts for itself -- leaves the bank (it was bumped twice per step), as do layers that never
    run (`sites` lists (parent, name, module): the owner's live batch norms)."""

    def __init__(self, sites):
        self.sites, self.flat = list(sites), None

    @property
    def mods(self):
        return [m for _, _, m in self.sites]

    def _gather(self):
        mods = self.mods
        vals = torch.stack([m.num_batches_tracked.reshape(()) for m in mods])
        self.flat = vals.clone()
        for i, m in enumerate(mods):
            m._buffers['num_batches_tracked'] = self.flat[i]

    def __call__(self, root, args=None):
        if not root.training or not self.sites:
            return
        live = [st for st in self.sites if st[0]._modules.get(st[1]) is st[2] and type(st[2]) is BatchNorm2dCounted]
        if len(live) != len(self.sites):      # some layers were replaced: they keep their (old) buffer and count for themselves
            self.sites, self.flat = live, None
            if not live:
                return
        mods = self.mods
        if self.flat is None or any(m.num_batches_tracked._base is not self.flat for m in mods):
            self._gather()
        self.flat.add_(1)


def fuse_batchnorm_counters(root, owners=None):
    """Batch-norm layers with a fixed momentum do not read `num_batches_tracked`; they only count.  Count for all layers of an
    OWNER with one kernel: the owner calls `bump_batchnorm_counters(self)` where its batch norms run (CamEncode.get_eff_depth,
    BevEncode.forward) -- not a forward hook of the root: training code that calls the sub-modules directly (`get_cam_feats`,
    `camencode.get_depth_and_context`, `get_voxels` + `bevencode`; the fused lift does, and the reference's notebooks do) then
    advances the counters exactly like a forward of the root.  State dicts keep the keys."""
    if not lean():
        return root
    for owner in (owners if owners is not None else [root]):
        # layers that never run keep their own counter at 0, like the reference's (lss.py:78-92 walks the EfficientNet trunk up to its
        # blocks: the head's `_bn1` is loaded from the checkpoint and never called)
        idle = {id(getattr(e, '_bn1', None)) for e in owner.modules() if isinstance(e, EfficientNetB0)}
        sites, seen = [], set(idle)
        for parent in owner.modules():
            for name, m in parent._modules.items():
                if (type(m) in (nn.BatchNorm2d, BatchNorm2dCounted) and m.track_running_stats and m.momentum is not None
                        and m.num_batches_tracked is not None and id(m) not in seen):      # (a layer shared by two parents counts once)
                    seen.add(id(m))
                    sites.append((parent, name, m))
        for _, _, m in sites:
            m.__class__ = BatchNorm2dCounted
        object.__setattr__(owner, '_bn_counter_bank', _CounterBank(sites))
    return root


def bump_batchnorm_counters(owner):
    """One `num_batches_tracked` bump for every batch norm of `owner` (training mode only); no-op without a bank."""
    bank = owner.__dict__.get('_bn_counter_bank')
    if bank is not None:
        bank(owner)


class _GlobalParams:
    drop_connect_rate = 0.2


# (repeats, kernel, stride, expand, in, out)
B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
             (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]


class EfficientNetB0(nn.Module):
    """Trunk with the member names lss.py:73-94 walks (`_conv_stem`, `_bn0`, `_swish`, `_blocks`, `_global_params`), plus
    the unused head (`_conv_head`, `_bn1`, `_fc`) so checkpoints load strictly."""

    def __init__(self, in_channels=3, num_classes=1000):
        super().__init__()
        self._global_params = _GlobalParams()
        self._conv_stem = Conv2dStaticSame(in_channels, 32, 3, stride=2)
        self._bn0 = nn.BatchNorm2d(32, momentum=0.01, eps=1e-3)
        blocks = []
        for r, k, s, e, ci, co in B0_STAGES:
            for i in range(r):
                blocks.append(MBConvBlock(ci if i == 0 else co, co, k, s if i == 0 else 1, e))
        self._blocks = nn.ModuleList(blocks)
        self._conv_head = Conv2dStaticSame(320, 1280, 1)
        self._bn1 = nn.BatchNorm2d(1280, momentum=0.01, eps=1e-3)
        self._fc = nn.Linear(1280, num_classes)
        self._swish = Swish()
        for m in self.modules():     # TF-style init (the reference overwrites it with ImageNet weights it downloads)
            if isinstance(m, nn.Conv2d):
                fan_out = m.kernel_size[0] * m.kernel_size[1] * m.out_channels // m.groups
                nn.init.normal_(m.weight, 0, math.sqrt(2.0 / fan_out))
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    @classmethod
    def from_pretrained(cls, name='efficientnet-b0', in_channels=3, **kw):
        """API-compatible constructor.  ImageNet weights cannot be downloaded here: random init (documented in DESIGN.md)."""
        assert name == 'efficientnet-b0'
        return cls(in_channels=in_channels)


# ------------------------------------------------------------------------------------------------------------
# ResNet-18 pieces (He et al. 2015), torchvision naming
# ------------------------------------------------------------------------------------------------------------
class BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class ResNet18Trunk(nn.Module):
    """`bn1`, `relu`, `layer1..3` of torchvision's resnet18(zero_init_residual=True)."""

    def __init__(self, zero_init_residual=True):
        super().__init__()
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.zeros_(m.bn2.weight)


def resnet18(zero_init_residual=True):
    return ResNet18Trunk(zero_init_residual)
