"""Stage A host side: ResNet50 / ResNet50-IBN-a on the HIP layer kernels.

`ResNet` / `ResNet_IBN` mirror the module tree of modelling/backbones/resnet.py:90-133 and
resnet_ibn_a.py:77-141 as pure PARAMETER HOLDERS (same attribute names -> same state_dict keys:
conv1.weight, bn1.running_mean, layer1.0.downsample.0.weight, ...; master weights stay fp32 OIHW
like the reference's checkpoints).  The arithmetic is done by `BackboneEngine`, an explicit
forward/backward schedule over NHWC activations that calls the C ABI (implicit-GEMM convs on the
MFMA pipe with BN statistics fused into the epilogue, fused BN+residual+ReLU, transposing-LDS
wgrad); torch only provides memory, the stream and the autograd hook at the module boundary.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os

import torch
from torch import nn

from . import _lib as L

LAYERS = (3, 4, 6, 3)
PLANES = (64, 128, 256, 512)


# ----------------------------------------------------------------------------- holders
class Conv2d(nn.Module):
    def __init__(self, cin, cout, k, stride=1, padding=0):
        super().__init__()
        self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding = cin, cout, k, stride, padding
        w = torch.empty(cout, cin, k, k)
        w.normal_(0, math.sqrt(2.0 / (k * k * cout)))           # resnet.py:156-160 random_init
        self.weight = nn.Parameter(w)


class BatchNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5, momentum=0.1):
        super().__init__()
        self.num_features, self.eps, self.momentum = c, eps, momentum
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


class BasicBlock(nn.Module):
    """modelling/backbones/resnet.py:22-48 (resnet18 / resnet34): conv3x3(stride) - bn - relu - conv3x3 - bn, + residual, relu."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


class ResNet(nn.Module):
    """Parameter tree of modelling/backbones/resnet.py:90-120 (Bottleneck, layers [3,4,6,3])."""
    arch = "resnet50"
    stem_relu = False            # resnet.py:97,125 -- the stem ReLU is commented out upstream

    def __init__(self, last_stride=2, block=Bottleneck, layers=LAYERS):
        super().__init__()
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, 2, 3)
        self.bn1 = BatchNorm2d(64)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=last_stride)
        self.out_channels = 512 * block.expansion          # Baseline.in_planes: 2048, or 512 for the BasicBlock networks

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, 0),
                                       BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def load_param(self, model_path):
        """resnet.py:135-154: strip 'backbone.base.' / 'base.' prefixes, skip fc/bottleneck/classifier."""
        param_dict = torch.load(model_path, map_location="cpu", weights_only=False)
        if "state_dict" in param_dict:
            param_dict = param_dict["state_dict"]
        own = self.state_dict()
        for i in param_dict:
            if "backbone" in i:
                name = i[14:]
            elif "base" in i:
                name = i[5:]
            else:
                name = i
            if any(t in i for t in ("fc", "bottleneck", "classifier", "transformer")):
                continue
            own[name].copy_(param_dict[i])


class InstanceNorm2d(nn.Module):
    def __init__(self, c, eps=1e-5):
        super().__init__()
        self.num_features, self.eps = c, eps
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class IBN(nn.Module):
    """resnet_ibn_a.py:18-32: first half of the channels InstanceNorm2d(affine), second half BatchNorm2d."""

    def __init__(self, planes):
        super().__init__()
        self.half = int(planes / 2)
        self.IN = InstanceNorm2d(self.half)
        self.BN = BatchNorm2d(planes - self.half)


class Bottleneck_IBN(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, ibn=False, stride=1, downsample=None):
        super().__init__()
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = IBN(planes) if ibn else BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride, 1)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, planes * 4, 1)
        self.bn3 = BatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride


class _UnusedFC(nn.Module):
    """resnet_ibn_a.py:92-93: the IBN backbone carries an unused Linear(2048 -> 1000) in its state_dict."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.zeros(cout, cin), requires_grad=False)
        self.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)


class ResNet_IBN(ResNet):
    """Parameter tree of modelling/backbones/resnet_ibn_a.py:77-124 (IBN in bn1 of layer1-3, stem WITH ReLU)."""
    arch = "resnet50_ibn_a"
    stem_relu = True             # resnet_ibn_a.py:129

    def __init__(self, last_stride=2, layers=LAYERS, num_classes=1000):
        nn.Module.__init__(self)
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, 7, 2, 3)
        self.bn1 = BatchNorm2d(64)
        self.layer1 = self._make_ibn_layer(64, layers[0])
        self.layer2 = self._make_ibn_layer(128, layers[1], stride=2)
        self.layer3 = self._make_ibn_layer(256, layers[2], stride=2)
        self.layer4 = self._make_ibn_layer(512, layers[3], stride=last_stride)
        self.out_channels = 2048
        self.fc = _UnusedFC(512 * 4, num_classes)

    def _make_ibn_layer(self, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * 4:
            downsample = nn.Sequential(Conv2d(self.inplanes, planes * 4, 1, stride, 0), BatchNorm2d(planes * 4))
        ibn = planes != 512                                                       # resnet_ibn_a.py:116-118
        layers = [Bottleneck_IBN(self.inplanes, planes, ibn, stride, downsample)]
        self.inplanes = planes * 4
        for _ in range(1, blocks):
            layers.append(Bottleneck_IBN(self.inplanes, planes, ibn))
        return nn.Sequential(*layers)

    def load_param(self, model_path):
        """resnet_ibn_a.py:143-162."""
        param_dict = torch.load(model_path, map_location="cpu", weights_only=False)
        if "state_dict" in param_dict:
            param_dict = param_dict["state_dict"]
        own = self.state_dict()
        for i in param_dict:
            if any(t in i for t in ("fc", "bottleneck", "reduce_embeddings.weight", "classifier")):
                continue
            own[i[5:] if "base" in i else i].copy_(param_dict[i])


def resnet50_ibn_a(last_stride, **kwargs):
    return ResNet_IBN(last_stride, LAYERS, **kwargs)


# the deeper Bottleneck variants of MODEL.NAME (modelling/baseline.py:73-81, resnet_ibn_a.py:173-181): same kernels, same
# engine, other block counts (the engine walks whatever blocks the parameter tree holds)
ARCH_LAYERS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3), "resnet152": (3, 8, 36, 3),
               "resnet50_ibn_a": (3, 4, 6, 3), "resnet101_ibn_a": (3, 4, 23, 3),
               "resnet18": (2, 2, 2, 2), "resnet34": (3, 4, 6, 3)}          # BasicBlock networks (modelling/baseline.py:56-65), round 6


def build_backbone(arch, last_stride):
    if arch not in ARCH_LAYERS:
        raise NotImplementedError(f"MODEL.NAME={arch!r}: the reference's Baseline knows {sorted(ARCH_LAYERS)} (modelling/baseline.py:56-81)")
    layers = ARCH_LAYERS[arch]
    if arch.endswith("_ibn_a"):
        net = ResNet_IBN(last_stride, layers)
    else:
        net = ResNet(last_stride, block=BasicBlock if arch in ("resnet18", "resnet34") else Bottleneck, layers=layers)
    net.arch = arch
    return net


# ----------------------------------------------------------------------------- engine
def _desc(B, H, W, cin, cout, k, stride, pad):
    oh = (H + 2 * pad - k) // stride + 1
    ow = (W + 2 * pad - k) // stride + 1
    return L.ConvDesc(B, H, W, cin, oh, ow, cout, k, k, stride, pad), oh, ow


class _ConvUnit:
    """conv + BN bookkeeping for one (holder conv, holder bn) pair."""
    __slots__ = ("conv", "bn", "ibn", "w_krsc", "w_crsk", "k", "stride", "pad", "cin", "cout", "fold")

    def __init__(self, conv, bn):
        self.ibn = bn if isinstance(bn, IBN) else None
        self.conv, self.bn = conv, (bn.BN if isinstance(bn, IBN) else bn)
        self.k, self.stride, self.pad = conv.kernel_size, conv.stride, conv.padding
        self.cin, self.cout = conv.in_channels, conv.out_channels
        self.w_krsc = self.w_crsk = None
        self.fold = None               # eval mode: BatchNorm as a per-channel affine, float [2][cout] (view of one flat buffer)


class BackboneEngine:
    """Explicit forward / backward schedule for one backbone instance."""

    def __init__(self, net: ResNet, dtype=torch.bfloat16):
        self.net = net
        self.dtype = dtype
        self.dt = L._DT[dtype]
        self.eval_ds_side = os.environ.get("CREID_EVAL_DS_SIDE", "0") == "1"   # eval forward: downsample conv on a side stream
        self.units = []
        self.stem = _ConvUnit(net.conv1, net.bn1)
        self.blocks = []
        # BasicBlock networks (resnet18 / resnet34, round 6): two 3 x 3 convolutions per block, no c3 -- the same kernels on a
        # plain schedule (_forward_basic / _backward_basic: none of the bottleneck-specific carriers and fusions)
        self.basic = isinstance(net.layer1[0], BasicBlock)
        self.cout = int(getattr(net, "out_channels", 2048))
        for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
            for blk in layer:
                u = dict(c1=_ConvUnit(blk.conv1, blk.bn1), c2=_ConvUnit(blk.conv2, blk.bn2),
                         c3=None if self.basic else _ConvUnit(blk.conv3, blk.bn3),
                         ds=_ConvUnit(blk.downsample[0], blk.downsample[1]) if blk.downsample is not None else None)
                self.blocks.append(u)
        self.weights_dirty = True      # set by writers torch cannot see (our ctypes optimiser kernels)
        self._wsig = None              # (data_ptr, _version) of every conv weight at the last prep_weights()
        self._wprep_key = None
        self.wprep_side = os.environ.get("CREID_WPREP_SIDE", "0") == "1"   # measured r06: +0.17 ms (both branches are HBM-bound; profiles/r06_graph_branches.md)
        # nn.Module.load_state_dict copies through `param.copy_` under no_grad (bumps _version), but a post-hook makes
        # the refresh independent of how a loader writes
        net.register_load_state_dict_post_hook(lambda *_a: setattr(self, "weights_dirty", True))
        self._ws = None
        # CREID_WGRAD_STREAM=1: weight gradients on a second HIP stream (a parallel branch of the captured graph).
        # Measured r01: the concurrent dgrad / wgrad kernels contend for LDS and L2 and the step gets 6 % SLOWER
        # (7970 vs 8480 img/s), so the default is one stream.
        self.wgrad_stream = os.environ.get("CREID_WGRAD_STREAM", "0") == "1"
        self._side = None
        self._keep = []
        self.reduce_stream = os.environ.get("CREID_REDUCE_STREAM", "0") == "1"
        self._ws2 = [None, None]
        # weight-gradient split reductions ride in the first workgroups of the NEXT data-gradient launch
        # (creid_conv2d_dgrad_fused_nhwc) instead of 53 stand-alone 5-25 us launches; CREID_WRED_PIGGYBACK=0: round-1 path
        self.wred_piggyback = os.environ.get("CREID_WRED_PIGGYBACK", "1") == "1" and not self.wgrad_stream \
            and not self.reduce_stream
        # BatchNorm-backward finalizes ride in the first workgroups of a weight-gradient launch issued between the data
        # gradient that produced their column sums and the apply that needs them (CREID_BNFIN_PIGGYBACK=0: own launches)
        # CREID_FIN_CARRIER: "wgrad" (above) | "wred" (finalize and the pending split reduction share one launch between the data
        # gradient and the apply; launch order wgrad -> dgrad as before) | "none"
        carrier = os.environ.get("CREID_FIN_CARRIER", "wgrad")
        is16 = dtype in (torch.bfloat16, torch.float16)
        self.loss_scaler = None        # f16 training: solver.LossScaler; backward() multiplies the incoming head gradient by its scale
        if os.environ.get("CREID_BNFIN_PIGGYBACK", "1") != "1" or not (self.wred_piggyback and is16):
            carrier = "none"
        self.bnfin_piggyback = carrier == "wgrad"
        self.fin_with_wred = carrier == "wred"
        self._bn_sums = {}             # id(unit) -> coefficient tensor whose finalize has already been issued
        self.wgrad_first = os.environ.get("CREID_WGRAD_FIRST", "0") == "1" or not self.bnfin_piggyback   # launch order
        self.carrier_max_bytes = int(float(os.environ.get("CREID_CARRIER_MAX_MB", "1e9")) * (1 << 20))
        self._wred_pending = []        # FIFO of (desc, grad tensor, workspace, nbytes)
        self._wred_ws = [None, None, None]
        self._wred_flip = 0
        self.on_group_done = None
        self._group_first, o = {}, 0             # index of each layer's FIRST block -> layer number
        for k, layer in enumerate((net.layer1, net.layer2, net.layer3, net.layer4), start=1):
            self._group_first[o] = k
            o += len(layer)
        self.saved = None
        # training forward writes each ReLU's mask as bits (1 byte per 8 channels); the BatchNorm backward and the fused
        # data-gradient epilogue read that instead of re-reading the activation (CREID_RELU_BITMASK=0: read the activation)
        self.relu_bitmask = is16 and os.environ.get("CREID_RELU_BITMASK", "1") == "1"
        self.stem_fuse_pool = os.environ.get("CREID_STEM_FUSE", "1") == "1"    # bn1 + maxpool in one pass (38 vs 51 us)
        self.stem_pool_fused = os.environ.get("CREID_STEM_POOL", "1") == "1"   # eval: conv1 + bn1 + maxpool in one launch
        self.pair_fused = os.environ.get("CREID_PAIR_FUSE", "1") == "1"        # eval, layer1: conv3 of block i + conv1 of block i + 1
        # the backward counterpart (max-pool gradient gathered inside the BN backward passes) is correct but slower:
        # the gather is VALU-bound and runs twice (146 vs 116 us, tools/debug/stem_tail_probe.py) -- off by default
        self.stem_fuse_pool_bwd = os.environ.get("CREID_STEM_FUSE_BWD", "0") == "1"
        self.drop_gm = os.environ.get("CREID_DROP_GM", "1") == "1"     # A/B knob: 0 = bn3's backward still writes the masked copy
        self.fuse_bn_reduce = is16 and os.environ.get("CREID_FUSE_BN_REDUCE", "1") == "1" \
            and os.environ.get("CREID_IGEMM_DMA", "1") == "1"
        self._pending_steps = None     # device counter of training forwards not yet folded into num_batches_tracked
        # eval-mode forward: BatchNorm (running statistics = constants) folded into the producing convolution's epilogue --
        # one launch per conv instead of conv -> finalize -> apply (CREID_EVAL_FOLD=0: the three-launch schedule)
        # (the folded epilogue of the 16-bit types lives in the LDS-DMA kernels only: with CREID_IGEMM_DMA=0 a bf16 forward takes
        # the three-launch schedule instead of failing; float16 has no register-staged kernels at all)
        if dtype == torch.float16 and os.environ.get("CREID_IGEMM_DMA", "1") != "1":
            raise L.CreidError("compute dtype float16 needs the LDS-DMA convolution kernels: unset CREID_IGEMM_DMA=0 "
                               "(the register-staged fallback kernels exist for bfloat16 and float32 only)")
        self.eval_fold = os.environ.get("CREID_EVAL_FOLD", "1") == "1" and \
            (dtype == torch.float32 or os.environ.get("CREID_IGEMM_DMA", "1") == "1")
        self._fold_key = None
        # TIMING-ONLY ablation (results are WRONG): bit 0 skips the forward BatchNorm apply launches of bn1 / bn2 (no residual),
        # bit 1 their backward apply launches -- the upper bound of what fusing those passes into the consuming / producing
        # convolutions could save (profiles/r03_bn_fusion_bound.md)
        # training forward: bn2 + ReLU applied inside conv3's operand path (creid_conv1x1_bnrelu_fwd) for the bottlenecks whose
        # width (conv3's input channels, 64 or 128) is listed here -- the stand-alone apply pass of bn2 disappears; "" = never.
        # Measured (profiles/r06_bn_apply_in_conv3.md): layer2 (128) gains ~8 us per block, layer1 (64) loses ~3 us per block
        self.c3_axf = {int(v) for v in os.environ.get("CREID_C3_AXF", "64,128").split(",") if v.strip()}
        self.ds_reduce2 = os.environ.get("CREID_DS_REDUCE2", "1") == "1"
        self.dual_apply = os.environ.get("CREID_DUAL_APPLY", "1") == "1"     # A/B knob: 0 = separate downsample-BN apply launch
        # training forward: BatchNorm finalize + apply as ONE launch on layers with at most this many statistic rows (M <= 8192 by
        # default: 24 launches less per B = 64 step at the same step time -- captured and eager --, bit-identical:
        # profiles/r05_fin_apply.md; 0 = never);
        # CREID_FIN_APPLY_RB = row blocks of that launch (0 = library rule)
        self.fin_apply_rows = int(os.environ.get("CREID_FIN_APPLY", "64"))
        self.fin_apply_rb = int(os.environ.get("CREID_FIN_APPLY_RB", "64"))
        self._apply_dry = int(os.environ.get("CREID_BN_APPLY_DRY", "0"))
        if self._apply_dry:
            import sys
            print(f"[creid] WARNING: CREID_BN_APPLY_DRY={self._apply_dry} is a timing-only ablation switch -- results are WRONG",
                  file=sys.stderr)

    # ---- helpers
    @property
    def device(self):
        return self.net.conv1.weight.device

    def _empty(self, *shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._keep.append(self._ws)          # a launch on the side stream may still be reading the old one
            self._ws = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        return self._ws

    def _fork_side(self, *tensors):
        """Side-stream context for work that depends on everything enqueued so far on the current stream; the
        tensors it reads are kept alive until _join_side()."""
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        self._side.wait_stream(torch.cuda.current_stream())
        self._keep.extend(tensors)
        return torch.cuda.stream(self._side)

    def _join_side(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)
        self._keep.clear()

    def all_units(self):
        yield self.stem
        for b in self.blocks:
            for k in ("c1", "c2", "c3", "ds"):
                if b[k] is not None:
                    yield b[k]

    def prep_weights(self, side=False):
        """fp32 OIHW master weights -> compute-dtype [O][r][s][I] and [I][r][s][O] copies (ONE launch for all
        52 non-stem convolutions through a device descriptor table, rebuilt only if a pointer moved).  side=True: that launch
        goes to the side stream (the caller joins it before the first non-stem convolution); returns whether it did."""
        import numpy as np
        lib, st = L.lib(), L.stream()
        units = [u for u in self.all_units() if u is not self.stem]
        for u in units:
            if u.w_krsc is None:
                u.w_krsc = self._empty(u.cout, u.k, u.k, u.cin)
                u.w_crsk = self._empty(u.cin, u.k, u.k, u.cout)
        if self.stem.w_krsc is None:
            self.stem.w_krsc = self._empty(64, 8, 32)
        key = tuple(u.conv.weight.data_ptr() for u in units)
        if self._wprep_key != key:
            rec = np.zeros(len(units), dtype=np.dtype([("w", "<u8"), ("krsc", "<u8"), ("crsk", "<u8"), ("O", "<i4"),
                                                        ("I", "<i4"), ("kh", "<i4"), ("kw", "<i4"), ("start", "<i8")]))
            assert lib.creid_weight_prep_entry_bytes() == rec.dtype.itemsize == 48
            start, tiles, tstart = 0, 0, np.zeros(len(units), np.int32)
            for i, u in enumerate(units):
                rec[i] = (u.conv.weight.data_ptr(), u.w_krsc.data_ptr(), u.w_crsk.data_ptr(), u.cout, u.cin, u.k, u.k, start)
                start += u.cout * u.cin * u.k * u.k
                tstart[i] = tiles
                tiles += ((u.cout + 31) // 32) * ((u.cin + 31) // 32)
            self._wprep_tab = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
            self._wprep_tiles = torch.from_numpy(tstart).to(self.device)
            self._wprep_total = tiles
            self._wprep_key = key
        L.check(lib.creid_stem_weight_prep(L.ptr(self.stem.conv.weight), self.dt, L.ptr(self.stem.w_krsc), st),
                "stem_weight_prep")
        if side:
            # the 52 non-stem copies (~50 us at ResNet50 size) are needed by layer1 only: inside a captured graph they run as a
            # parallel branch beside the stem (image layout pass, 7 x 7 convolution, BatchNorm, max-pool: ~95 us that depend on the
            # stem copy alone); forward() joins the branch in front of the first bottleneck.  One fork / join costs ~6 us of graph
            # signalling (tools/probes/anyorder_probe.hip) against ~45 us hidden.
            with self._fork_side():
                L.check(lib.creid_weight_prep_multi(L.ptr(self._wprep_tab), L.ptr(self._wprep_tiles), len(units), self._wprep_total,
                                                    self.dt, L.stream()), "weight_prep_multi")
        else:
            L.check(lib.creid_weight_prep_multi(L.ptr(self._wprep_tab), L.ptr(self._wprep_tiles), len(units), self._wprep_total,
                                                self.dt, st), "weight_prep_multi")
        self.weights_dirty = False
        self._wsig = self._weight_signature()
        return side

    def _weight_signature(self):
        """Changes whenever torch-visible code rewrites or re-homes a convolution weight after the compute-dtype
        copies were made (load_state_dict, an external / PL optimiser, EMA, broadcast, .to()): in-place ops bump
        `_version`, re-homing changes `data_ptr`.  Writers torch cannot see set `weights_dirty` themselves."""
        return tuple((u.conv.weight.data_ptr(), u.conv.weight._version) for u in self.all_units())

    def fold_counters(self):
        """Fold the pending step count into every BatchNorm2d.num_batches_tracked (before a state_dict)."""
        if self._pending_steps is None:
            return
        for u in self.all_units():
            u.bn.num_batches_tracked += self._pending_steps.to(u.bn.num_batches_tracked.device)
        self._pending_steps.zero_()

    def fold_bn(self):
        """Eval mode: (scale, shift) of every plain BatchNorm2d from its running statistics, ONE launch over a device
        table (rebuilt only if a tensor moved); always re-evaluated, so statistics updated by a training step or a
        load_state_dict are picked up without any dirty tracking."""
        import numpy as np
        lib, st = L.lib(), L.stream()
        units = [u for u in self.all_units() if u.ibn is None]
        key = tuple((u.bn.weight.data_ptr(), u.bn.bias.data_ptr(), u.bn.running_mean.data_ptr(),
                     u.bn.running_var.data_ptr()) for u in units)
        if self._fold_key != key:
            total = sum(2 * u.cout for u in units)
            self._fold_buf = torch.empty(total, dtype=torch.float32, device=self.device)
            rec = np.zeros(len(units), dtype=np.dtype([("g", "<u8"), ("b", "<u8"), ("m", "<u8"), ("v", "<u8"), ("o", "<u8"),
                                                       ("C", "<i4"), ("eps", "<f4")]))
            assert lib.creid_bn2d_fold_entry_bytes() == rec.dtype.itemsize == 48
            o = 0
            for i, u in enumerate(units):
                u.fold = self._fold_buf[o:o + 2 * u.cout].view(2, u.cout)
                rec[i] = (u.bn.weight.data_ptr(), u.bn.bias.data_ptr(), u.bn.running_mean.data_ptr(),
                          u.bn.running_var.data_ptr(), u.fold.data_ptr(), u.cout, u.bn.eps)
                o += 2 * u.cout
            self._fold_tab = torch.from_numpy(rec.view(np.uint8).copy()).to(self.device)
            self._fold_n = len(units)
            self._fold_key = key
        L.check(lib.creid_bn2d_fold_multi(L.ptr(self._fold_tab), self._fold_n, st), "bn2d_fold_multi")

    # ---- layer steps
    def _conv_fold(self, u, a_in, B, H, W, relu, residual=None):
        """Eval mode: conv -> folded BatchNorm -> (+residual) -> (ReLU) in one launch.  Returns (a_out, oh, ow)."""
        lib, st = L.lib(), L.stream()
        d, oh, ow = _desc(B, H, W, u.cin, u.cout, u.k, u.stride, u.pad)
        a = self._empty(B * oh * ow, u.cout)
        L.check(lib.creid_conv2d_fwd_affine_nhwc(C.byref(d), L.ptr(a_in), L.ptr(u.w_krsc), L.ptr(a), L.ptr(u.fold),
                                                 L.ptr(residual), 1 if relu else 0, self.dt, st), "conv2d_fwd_affine")
        return a, oh, ow

    def _stem_operand(self, x, B, H, W):
        """The stem convolution's operand: zero-padded NHWC4 [B, H + 8, W + 6, 4] in the compute dtype.  An fp32 NCHW batch
        goes through the layout pass; a transforms.StemOperand (the device-side input transforms wrote that layout directly)
        is used as it is."""
        if not isinstance(x, torch.Tensor):
            assert x.xpad.dtype == self.dtype and tuple(x.xpad.shape) == (B, H + 8, W + 6, 4), "stem operand of another dtype / shape"
            return x.xpad
        xpad = self._empty(B, H + 8, W + 6, 4)
        L.check(L.lib().creid_image_to_nhwc4_pad(L.ptr(x), B, H, W, self.dt, L.ptr(xpad), L.stream()), "image_pad")
        return xpad

    def _forward_eval_folded(self, x_nchw, want_base_out):
        """validation_step / inference forward (modelling/bases.py:169-177, inference/inference_utils.py:104-113): 53
        convolutions with their BatchNorm folded in + pad, max-pool, GAP -- no statistics, no separate normalisation
        passes.  IBN blocks (resnet_ibn_a.py:27-32) keep their own pass: InstanceNorm needs per-image statistics."""
        lib, st = L.lib(), L.stream()
        B, _, H, W = x_nchw.shape
        self.fold_bn()
        xpad = self._stem_operand(x_nchw, B, H, W)
        H1, W1 = H // 2, W // 2
        H2, W2 = H1 // 2, W1 // 2
        a = self._empty(B * H2 * W2, 64)
        rc = -4
        if self.stem_pool_fused and self.dtype != torch.float32:
            # conv1 + folded bn1 (+ ReLU) + max-pool in one launch: the full-resolution tensor is never written (conv_stem.hip)
            rc = lib.creid_stem_conv_pool_fwd_affine(B, H, W, L.ptr(xpad), L.ptr(self.stem.w_krsc), L.ptr(a), L.ptr(self.stem.fold),
                                                     1 if self.net.stem_relu else 0, self.dt, st)
            if rc != -4:
                L.check(rc, "stem_conv_pool_fwd_affine")
        if rc == -4:                                     # image size outside the fused kernel's tiles (or fp32): two launches
            y0 = self._empty(B * H1 * W1, 64)
            L.check(lib.creid_stem_conv_fwd_affine(B, H, W, L.ptr(xpad), L.ptr(self.stem.w_krsc), L.ptr(y0), L.ptr(self.stem.fold),
                                                   1 if self.net.stem_relu else 0, self.dt, st), "stem_conv_fwd_affine")
            L.check(lib.creid_maxpool3x3s2_fwd(L.ptr(y0), B, H1, W1, 64, self.dt, L.ptr(a), None, st), "maxpool_fwd")   # no argmax taps
            del y0
        del xpad
        h, w = H2, W2
        a1_next = None                    # conv1 output of the CURRENT block when the previous block's conv3 launch produced it
        for bi, b in enumerate(self.blocks):
            a_in, hin, win = a, h, w
            if self.basic:                # resnet.py:33-48: two folded 3 x 3 convolutions, the residual added in the second one's epilogue
                a1, h1, w1 = self._conv_fold(b["c1"], a_in, B, hin, win, True)
                r = a_in if b["ds"] is None else self._conv_fold(b["ds"], a_in, B, hin, win, False)[0]
                a, h, w = self._conv_fold(b["c2"], a1, B, h1, w1, True, residual=r)
                continue
            side = b["ds"] is not None and self.eval_ds_side
            if side:
                # the downsample convolution depends on the block input only: it runs on a side stream (a parallel branch of a
                # captured graph) beside conv1 / conv2 and is joined before conv3, whose epilogue adds it
                with self._fork_side(a_in):
                    r = self._conv_fold(b["ds"], a_in, B, hin, win, False)[0]
            if a1_next is not None:
                a1, h1, w1, a1_next = a1_next, hin, win, None
            elif b["c1"].ibn is not None:
                _, a1, _, _, h1, w1 = self._conv_bn(b["c1"], a_in, B, hin, win, False, True)
            else:
                a1, h1, w1 = self._conv_fold(b["c1"], a_in, B, hin, win, True)
            a2, h2, w2 = self._conv_fold(b["c2"], a1, B, h1, w1, True)
            if side:
                self._join_side()
                r.record_stream(torch.cuda.current_stream())
            else:
                r = a_in if b["ds"] is None else self._conv_fold(b["ds"], a_in, B, hin, win, False)[0]
            nb = self.blocks[bi + 1] if bi + 1 < len(self.blocks) else None
            pair = (self.pair_fused and nb is not None and nb["ds"] is None and self.dtype != torch.float32
                    and (b["c3"].cin, b["c3"].cout, nb["c1"].cout) == (64, 256, 64) and nb["c1"].stride == 1)
            if pair and nb["c1"].ibn is not None:
                pair = (h2 * w2) % 128 == 0              # the statistics partials must be per-image row blocks
            if pair:
                # conv3 + bn3 + residual + ReLU of this block AND conv1 (+ bn1 + ReLU, or the statistics an IBN layer needs) of the
                # next one in one launch: the block output is written (it is the next residual) but never read back by conv1
                # (conv_pair.hip; layer1 only)
                M3 = B * h2 * w2
                a = self._empty(M3, 256)
                x1 = self._empty(M3, 64)
                if nb["c1"].ibn is None:
                    L.check(lib.creid_bottleneck_c3_c1_fwd_affine(M3, 64, 256, 64, L.ptr(a2), L.ptr(b["c3"].w_krsc), L.ptr(b["c3"].fold),
                                                                  L.ptr(r), L.ptr(a), L.ptr(nb["c1"].w_krsc), L.ptr(nb["c1"].fold),
                                                                  L.ptr(x1), self.dt, st), "bottleneck_c3_c1_fwd_affine")
                    a1_next = x1
                else:
                    part = self._empty((M3 // 128) * 2, 64, dtype=torch.float32)
                    L.check(lib.creid_bottleneck_c3_c1_fwd_stats(M3, 64, 256, 64, L.ptr(a2), L.ptr(b["c3"].w_krsc), L.ptr(b["c3"].fold),
                                                                 L.ptr(r), L.ptr(a), L.ptr(nb["c1"].w_krsc), L.ptr(x1), L.ptr(part),
                                                                 self.dt, st), "bottleneck_c3_c1_fwd_stats")
                    a1_next = self._ibn_tail(nb["c1"], x1, B, h2 * w2, False, True, part)[0]
                h, w = h2, w2
            else:
                a, h, w = self._conv_fold(b["c3"], a2, B, h2, w2, True, residual=r)
        feat = self._empty(B, self.cout, dtype=torch.float32)
        L.check(lib.creid_gap_fwd(L.ptr(a), B, h * w, self.cout, self.dt, L.ptr(feat), st), "gap_fwd")
        self.saved = None
        base_out = None
        if want_base_out:
            base_out = self._empty(B, self.cout, h, w, dtype=torch.float32)
            L.check(lib.creid_nhwc_to_nchw_f32(L.ptr(a), B, h * w, self.cout, self.dt, L.ptr(base_out), st), "nhwc_to_nchw")
        return base_out, feat

    def _conv_bn(self, u, a_in, B, H, W, training, relu, residual=None, residual_ss=None, apply=True):
        """conv -> BN(batch or running stats) -> (+residual) -> (ReLU).  Returns (x_raw, a_out, mean, invstd, oh, ow).
        apply=False: statistics only, a_out is the (scale, shift) pair instead (the downsample branch: its normalisation
        happens inside the block's bn3 pass, which takes the raw tensor as `residual` and that pair as `residual_ss`)."""
        lib, st = L.lib(), L.stream()
        d, oh, ow = _desc(B, H, W, u.cin, u.cout, u.k, u.stride, u.pad)
        M = B * oh * ow
        x = self._empty(M, u.cout)
        if u.ibn is not None:
            # the conv epilogue's per-128-row (sum, sumsq) partials are per-image row blocks when HW % 128 == 0:
            # IBN then needs no statistics pass of its own (InstanceNorm uses instance statistics in eval too)
            part = None
            if (oh * ow) % 128 == 0:
                part = self._empty(lib.creid_conv2d_bn_partial_rows(C.byref(d)) * 2, u.cout, dtype=torch.float32)
            L.check(lib.creid_conv2d_fwd_nhwc(C.byref(d), L.ptr(a_in), L.ptr(u.w_krsc), L.ptr(x), L.ptr(part), self.dt, st),
                    "conv2d_fwd")
            return (x,) + self._ibn_tail(u, x, B, oh * ow, training, relu, part) + (oh, ow)
        rows = lib.creid_conv2d_bn_partial_rows(C.byref(d)) if training else 0
        part = self._empty(rows * 2, u.cout, dtype=torch.float32) if training else None
        L.check(lib.creid_conv2d_fwd_nhwc(C.byref(d), L.ptr(a_in), L.ptr(u.w_krsc), L.ptr(x), L.ptr(part), self.dt, st),
                "conv2d_fwd")
        return (x,) + self._bn_tail(u, x, part, rows, M, training, relu, residual, residual_ss, apply) + (oh, ow)

    def _conv3_axf(self, u, x_raw, ss, B, H, W, residual, residual_ss):
        """conv3 of a bottleneck on conv2's RAW output: bn2 + ReLU on the operand path, the normalised tensor and its ReLU bits as
        side outputs (what the backward reads), then bn3 as usual.  Returns (x3, a2, a3, mean3, invstd3, oh, ow)."""
        lib, st = L.lib(), L.stream()
        d, oh, ow = _desc(B, H, W, u.cin, u.cout, 1, 1, 0)
        M = B * oh * ow
        x = self._empty(M, u.cout)
        rows = lib.creid_conv2d_bn_partial_rows(C.byref(d))
        part = self._empty(rows * 2, u.cout, dtype=torch.float32)
        a_in = self._empty(M, u.cin)
        mask = torch.empty(M * u.cin // 8, dtype=torch.uint8, device=self.device)
        L.check(lib.creid_conv1x1_bnrelu_fwd(L.ptr(x_raw), L.ptr(ss), L.ptr(u.w_krsc), M, u.cin, u.cout, L.ptr(x), L.ptr(part),
                                             L.ptr(a_in), L.ptr(mask), self.dt, st), "conv1x1_bnrelu_fwd")
        a_in._relu_mask = mask
        return (x, a_in) + self._bn_tail(u, x, part, rows, M, True, True, residual, residual_ss, True) + (oh, ow)

    def _ibn_tail(self, u, x, B, HW, training, relu, part=None):
        lib, st = L.lib(), L.stream()
        ibn, bn = u.ibn, u.bn
        rpi = lib.creid_ibn_rows_per_image(HW)
        ready = 1 if part is not None else 0
        if part is None:
            part = self._empty(B * rpi * 2, u.cout, dtype=torch.float32)
        mean = self._empty(B, u.cout, dtype=torch.float32)
        invstd = self._empty(B, u.cout, dtype=torch.float32)
        ss = self._empty(B * 2, u.cout, dtype=torch.float32)
        a = self._empty(B * HW, u.cout)
        mask = None
        if training and relu and self.relu_bitmask:
            mask = torch.empty(B * HW * u.cout // 8, dtype=torch.uint8, device=self.device)
        L.check(lib.creid_ibn_fwd_mask(L.ptr(x), B, HW, u.cout, ibn.half, L.ptr(ibn.IN.weight), L.ptr(ibn.IN.bias),
                                       L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                       1 if training else 0, bn.momentum, bn.eps, 1 if relu else 0, self.dt, L.ptr(part), ready,
                                       L.ptr(mean), L.ptr(invstd), L.ptr(ss), L.ptr(a), L.ptr(mask), st), "ibn_fwd")
        if mask is not None:
            a._relu_mask = mask
        return a, mean, invstd

    def _ibn_bwd(self, u, x, g, act, mean, invstd, B, HW, part=None):
        lib, st = L.lib(), L.stream()
        ibn, bn = u.ibn, u.bn
        rpi = lib.creid_ibn_rows_per_image(HW)
        ready = 1 if part is not None else 0
        if part is None:
            part = self._empty(B * rpi * 2, u.cout, dtype=torch.float32)
        coef = self._empty(B * 3, u.cout, dtype=torch.float32)
        per_img = self._empty(B * 2, ibn.half, dtype=torch.float32)
        dx = self._empty(B * HW, u.cout)
        mask = getattr(act, "_relu_mask", None) if act is not None else None
        L.check(lib.creid_ibn_bwd_mask(L.ptr(x), L.ptr(g), L.ptr(act), L.ptr(mask), L.ptr(mean), L.ptr(invstd), B, HW, u.cout, ibn.half,
                                  L.ptr(ibn.IN.weight), L.ptr(bn.weight), self.dt, L.ptr(part), ready, L.ptr(coef), L.ptr(per_img),
                                  L.ptr(self._grad_of(ibn.IN.weight)), L.ptr(self._grad_of(ibn.IN.bias)),
                                  L.ptr(self._grad_of(bn.weight)), L.ptr(self._grad_of(bn.bias)), L.ptr(dx), st), "ibn_bwd")
        return dx, None

    def _bn_tail(self, u, x, part, rows, M, training, relu, residual, residual_ss=None, apply=True):
        lib, st = L.lib(), L.stream()
        bn = u.bn
        mean = self._empty(u.cout, dtype=torch.float32)
        invstd = self._empty(u.cout, dtype=torch.float32)
        ss = self._empty(2, u.cout, dtype=torch.float32)
        if (self.fin_apply_rows and training and apply and residual_ss is None and rows <= self.fin_apply_rows
                and u.cout % (32 if self.dtype == torch.float32 else 64) == 0 and not self._apply_dry):
            # finalize + apply in ONE launch (CREID_FIN_APPLY=<max statistic rows>; profiles/r05_fin_apply.md)
            a = self._empty(M, u.cout)
            mask = None
            if relu and self.relu_bitmask:
                mask = torch.empty(M * u.cout // 8, dtype=torch.uint8, device=self.device)
            L.check(lib.creid_bn2d_finalize_apply_mask(L.ptr(part), rows, u.cout, M, L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                                       bn.momentum, bn.eps, L.ptr(bn.weight), L.ptr(bn.bias), L.ptr(mean), L.ptr(invstd),
                                                       L.ptr(ss), L.ptr(x), L.ptr(residual), 1 if relu else 0, self.dt, L.ptr(a),
                                                       L.ptr(mask), self.fin_apply_rb, st), "bn2d_finalize_apply")
            if mask is not None:
                a._relu_mask = mask
            return a, mean, invstd
        L.check(lib.creid_bn2d_finalize(L.ptr(part), rows, u.cout, M, L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                        1 if training else 0, bn.momentum, bn.eps, L.ptr(bn.weight), L.ptr(bn.bias),
                                        L.ptr(mean), L.ptr(invstd), L.ptr(ss), st), "bn2d_finalize")
        if not apply:
            return ss, mean, invstd
        a = self._empty(M, u.cout)
        mask = None
        if training and relu and self.relu_bitmask:
            mask = torch.empty(M * u.cout // 8, dtype=torch.uint8, device=self.device)
        if residual_ss is not None:
            L.check(lib.creid_bn2d_apply_dual_mask(L.ptr(x), L.ptr(ss), L.ptr(residual), L.ptr(residual_ss), 1 if relu else 0, M,
                                                   u.cout, self.dt, L.ptr(a), L.ptr(mask), st), "bn2d_apply_dual")
        elif not (self._apply_dry & 1 and training and relu and residual is None):
            L.check(lib.creid_bn2d_apply_mask(L.ptr(x), L.ptr(ss), L.ptr(residual), 1 if relu else 0, M, u.cout, self.dt,
                                              L.ptr(a), L.ptr(mask), st), "bn2d_apply")
        if mask is not None:
            a._relu_mask = mask            # travels with the saved activation to _bn_bwd / _dgrad
        return a, mean, invstd

    # ---- forward
    def forward(self, x_nchw, training: bool, want_base_out: bool = False):
        """x_nchw: fp32 [B, 3, H, W] on the GPU, or a transforms.StemOperand (same batch, already in the stem's layout)."""
        if isinstance(x_nchw, torch.Tensor):
            L.require_gpu(x_nchw)
            assert x_nchw.dtype == torch.float32 and x_nchw.dim() == 4 and x_nchw.shape[1] == 3
        else:
            L.require_gpu(x_nchw.xpad)
        wprep_pending = False
        if self.weights_dirty or self._wsig != self._weight_signature():
            # training steps replayed from a hipGraph re-derive the 16-bit weight copies every step (the optimiser has just rewritten
            # the masters): as a graph branch beside the stem when capturing (CREID_WPREP_SIDE=0: in line)
            side = (training and self.wprep_side and self._wprep_key is not None and torch.cuda.is_current_stream_capturing()
                    and not (self.wgrad_stream or self.reduce_stream or self.eval_ds_side))
            wprep_pending = self.prep_weights(side=side)
        lib, st = L.lib(), L.stream()
        B, _, H, W = x_nchw.shape
        if not training and self.eval_fold:
            return self._forward_eval_folded(x_nchw, want_base_out)
        # (f16 -- the reference's precision=16, utils/misc.py:111 -- trains like bf16; its gradients need the dynamic loss scale
        # that ModelBase.configure_optimizers attaches as `loss_scaler`: solver.LossScaler)
        sv = {"B": B, "H": H, "W": W, "training": training}
        if training:      # ONE device counter per step instead of 53 per-layer `num_batches_tracked += 1`; the increment itself
            if self._pending_steps is None:          # rides in the forward's last launch (creid_gap_fwd_count)
                self._pending_steps = torch.zeros((), dtype=torch.long, device=self.device)
            if self._side is None and not torch.cuda.is_current_stream_capturing():
                self._side = torch.cuda.Stream(device=self.device)     # (the weight-copy branch of captured steps forks onto it)
        # stem
        xpad = self._stem_operand(x_nchw, B, H, W)
        H1, W1 = H // 2, W // 2
        M0 = B * H1 * W1
        x0 = self._empty(M0, 64)
        rows = (M0 + 127) // 128 if training else 0
        part = self._empty(rows * 2, 64, dtype=torch.float32) if training else None
        L.check(lib.creid_stem_conv_fwd(B, H, W, L.ptr(xpad), L.ptr(self.stem.w_krsc), L.ptr(x0), L.ptr(part), self.dt, st),
                "stem_conv_fwd")
        H2, W2 = H1 // 2, W1 // 2
        p0 = self._empty(B * H2 * W2, 64)
        idx0 = self._empty(B * H2 * W2, 64, dtype=torch.uint8)
        if self.stem_fuse_pool and not self.net.stem_relu:
            # bn1 + maxpool in one pass over the raw conv output: the normalised 64-channel full-resolution tensor is read
            # by nothing else (no stem ReLU -> the backward needs no mask of it either) and is never written
            bn = self.stem.bn
            mean0 = self._empty(64, dtype=torch.float32); invstd0 = self._empty(64, dtype=torch.float32)
            ss = self._empty(2, 64, dtype=torch.float32)
            L.check(lib.creid_bn2d_finalize(L.ptr(part), rows, 64, M0, L.ptr(bn.running_mean), L.ptr(bn.running_var),
                                            1 if training else 0, bn.momentum, bn.eps, L.ptr(bn.weight), L.ptr(bn.bias),
                                            L.ptr(mean0), L.ptr(invstd0), L.ptr(ss), st), "bn2d_finalize")
            L.check(lib.creid_bn2d_apply_maxpool3x3s2(L.ptr(x0), L.ptr(ss), 0, B, H1, W1, 64, self.dt, L.ptr(p0), L.ptr(idx0),
                                                      st), "bn2d_apply_maxpool")
            y0 = None
        else:
            y0, mean0, invstd0 = self._bn_tail(self.stem, x0, part, rows, M0, training, self.net.stem_relu, None)
            L.check(lib.creid_maxpool3x3s2_fwd(L.ptr(y0), B, H1, W1, 64, self.dt, L.ptr(p0), L.ptr(idx0), st), "maxpool_fwd")
        sv["stem"] = (xpad, x0, y0, mean0, invstd0, idx0)
        a, h, w = p0, H2, W2
        sv["blocks"] = []
        if wprep_pending:
            self._join_side()                                  # the non-stem weight copies (a graph branch beside the stem)
        for b in self.blocks:
            a_in, hin, win = a, h, w
            if self.basic:
                # BasicBlock (resnet.py:33-48): conv1 -> bn1 -> relu -> conv2 -> bn2 (+ residual, through the downsample branch's
                # BatchNorm where there is one: applied inside bn2's pass like the bottleneck's bn3) -> relu
                x1, a1, m1, i1, h1, w1 = self._conv_bn(b["c1"], a_in, B, hin, win, training, True)
                xd = md = idd = rss = None
                r = a_in
                if b["ds"] is not None:
                    xd, rss, md, idd, _, _ = self._conv_bn(b["ds"], a_in, B, hin, win, training, False, apply=False)
                    r = xd
                x2, a2, m2, i2, h2, w2 = self._conv_bn(b["c2"], a1, B, h1, w1, training, True, residual=r, residual_ss=rss)
                if training:
                    sv["blocks"].append(dict(a_in=a_in, hin=hin, win=win, x1=x1, a1=a1, m1=m1, i1=i1, h1=h1, w1=w1,
                                             x2=x2, a2=a2, m2=m2, i2=i2, h2=h2, w2=w2, xd=xd, md=md, idd=idd))
                a, h, w = a2, h2, w2
                continue
            x1, a1, m1, i1, h1, w1 = self._conv_bn(b["c1"], a_in, B, hin, win, training, True)
            axf = (training and self.dtype != torch.float32 and self.relu_bitmask and not self._apply_dry and b["c3"].k == 1
                   and b["c3"].stride == 1 and b["c3"].cin in self.c3_axf and b["c3"].cin in (64, 128) and b["c2"].ibn is None)
            if axf:      # conv2 + statistics only: bn2's (scale, shift) go to conv3, which normalises its operand itself
                x2, ss2, m2, i2, h2, w2 = self._conv_bn(b["c2"], a1, B, h1, w1, training, True, apply=False)
            else:
                x2, a2, m2, i2, h2, w2 = self._conv_bn(b["c2"], a1, B, h1, w1, training, True)
            rss = None
            if b["ds"] is not None and self.dual_apply:
                # downsample branch: conv + statistics only; its normalisation rides in bn3's apply pass (one launch and
                # the write + read of the normalised branch tensor less per downsample block)
                xd, rss, md, idd, _, _ = self._conv_bn(b["ds"], a_in, B, hin, win, training, False, apply=False)
                r = xd
            elif b["ds"] is not None:
                xd, r, md, idd, _, _ = self._conv_bn(b["ds"], a_in, B, hin, win, training, False)
            else:
                xd, r, md, idd = None, a_in, None, None
            if axf:
                x3, a2, a3, m3, i3, h3, w3 = self._conv3_axf(b["c3"], x2, ss2, B, h2, w2, r, rss)
            else:
                x3, a3, m3, i3, h3, w3 = self._conv_bn(b["c3"], a2, B, h2, w2, training, True, residual=r, residual_ss=rss)
            if training:
                sv["blocks"].append(dict(a_in=a_in, hin=hin, win=win, x1=x1, a1=a1, m1=m1, i1=i1, h1=h1, w1=w1,
                                         x2=x2, a2=a2, m2=m2, i2=i2, h2=h2, w2=w2, xd=xd, md=md, idd=idd,
                                         x3=x3, a3=a3, m3=m3, i3=i3))
            a, h, w = a3, h3, w3
        feat = self._empty(B, self.cout, dtype=torch.float32)
        if training:
            L.check(lib.creid_gap_fwd_count(L.ptr(a), B, h * w, self.cout, self.dt, L.ptr(feat), L.ptr(self._pending_steps), st), "gap_fwd")
        else:
            L.check(lib.creid_gap_fwd(L.ptr(a), B, h * w, self.cout, self.dt, L.ptr(feat), st), "gap_fwd")
        sv["final"] = (h, w)
        self.saved = sv if training else None
        base_out = None
        if want_base_out:
            base_out = self._empty(B, self.cout, h, w, dtype=torch.float32)
            L.check(lib.creid_nhwc_to_nchw_f32(L.ptr(a), B, h * w, self.cout, self.dt, L.ptr(base_out), st), "nhwc_to_nchw")
        return base_out, feat

    # ---- backward
    def _grad_of(self, p):
        if p.grad is None:
            p.grad = torch.zeros_like(p)
        return p.grad

    def _bn_bwd(self, u, x, g, act, mean, invstd, M, want_gm=False, part=None, mask=None, dry=False, reduce2=None):
        """BN backward; `part` = column-reduction partials already produced by a fused dgrad epilogue; `mask` = ReLU bits
        to apply to g (default: the ones that travel with `act`).  reduce2 = (x2, mean2, invstd2): the apply pass also produces
        the column sums of a second BatchNorm backward over the same masked gradient (the downsample branch's); returned in
        place of gm."""
        lib, st = L.lib(), L.stream()
        rows = lib.creid_bn2d_bwd_rows(M)
        ready = 1 if part is not None else 0
        sums = self._bn_sums.pop(id(u), None)
        if sums is not None:
            ready = 2                      # finalize already issued (it rode in a weight-gradient launch): apply only
        elif part is None:
            part = self._empty(rows * 2, u.cout, dtype=torch.float32)
        if sums is None:
            sums = self._empty(3, u.cout, dtype=torch.float32)
        dx = self._empty(M, u.cout)
        gm = self._empty(M, u.cout) if want_gm else None
        bn = u.bn
        dgam = self._grad_of(bn.weight) if bn.weight.requires_grad else None
        dbet = self._grad_of(bn.bias) if bn.bias.requires_grad else None
        if ready == 1 and self.fin_with_wred and self._wred_pending:
            # the finalize and the oldest pending split reduction in ONE launch, then apply only
            rd, rgw, rws, rbytes = self._wred_pending.pop(0)
            L.check(lib.creid_bn2d_bwd_finalize_wred(L.ptr(part), rows, u.cout, M, L.ptr(mean), L.ptr(invstd), L.ptr(bn.weight),
                                                     L.ptr(sums), L.ptr(dgam), L.ptr(dbet), C.byref(rd), L.ptr(rgw), 1, L.ptr(rws),
                                                     rbytes, self.dt, st), "bn2d_bwd_finalize_wred")
            ready = 2
        if mask is None and act is not None:
            mask = getattr(act, "_relu_mask", None)
        if dry and ready == 2:
            return dx, gm
        if reduce2 is not None:
            x2, mean2, invstd2 = reduce2
            part2 = self._empty(rows * 2, u.cout, dtype=torch.float32)
            L.check(lib.creid_bn2d_bwd_mask_reduce2(L.ptr(x), L.ptr(g), L.ptr(mask), L.ptr(mean), L.ptr(invstd), L.ptr(bn.weight), M,
                                                    u.cout, self.dt, L.ptr(part), ready, L.ptr(sums), L.ptr(dgam), L.ptr(dbet),
                                                    L.ptr(dx), L.ptr(x2), L.ptr(mean2), L.ptr(invstd2), L.ptr(part2), st),
                    "bn2d_bwd_reduce2")
            return dx, part2
        L.check(lib.creid_bn2d_bwd_mask(L.ptr(x), L.ptr(g), L.ptr(act), L.ptr(mask), L.ptr(mean), L.ptr(invstd),
                                        L.ptr(bn.weight), M, u.cout, self.dt, L.ptr(part), ready, L.ptr(sums), L.ptr(dgam),
                                        L.ptr(dbet), L.ptr(dx), L.ptr(gm), st), "bn2d_bwd")
        return dx, gm

    def _wgrad(self, u, a_in, dy, B, H, W, fin=None):
        """Weight gradient.  fin = (unit, partials, M) of a BatchNorm whose backward finalize should ride in this launch
        (ignored -- the BatchNorm backward then runs its own finalize -- where the carrier does not apply)."""
        if not u.conv.weight.requires_grad:
            return
        if self.wgrad_stream:
            with self._fork_side(a_in, dy):
                self._wgrad_launch(u, a_in, dy, B, H, W)
        else:
            self._wgrad_launch(u, a_in, dy, B, H, W, fin)

    def _wgrad_launch(self, u, a_in, dy, B, H, W, fin=None):
        lib, st = L.lib(), L.stream()
        d, _, _ = _desc(B, H, W, u.cin, u.cout, u.k, u.stride, u.pad)
        nbytes = lib.creid_conv2d_wgrad_workspace_bytes(C.byref(d), self.dt)
        gw = self._grad_of(u.conv.weight)
        if self.wred_piggyback:
            # partial tiles now; the sum over the splits is carried by the next data-gradient launch.  At most two
            # jobs are ever pending (c1 and the downsample branch), so three rotating workspaces never collide.
            k = self._wred_flip = (self._wred_flip + 1) % 3
            if self._wred_ws[k] is None or self._wred_ws[k].numel() < nbytes:
                self._keep.append(self._wred_ws[k])
                self._wred_ws[k] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
            ws = self._wred_ws[k]
            if fin is not None and self.bnfin_piggyback and fin[1] is not None:
                fu, fpart, fM, mean, invstd = fin
                fbn = fu.bn
                sums = self._empty(3, fu.cout, dtype=torch.float32)
                dgam = self._grad_of(fbn.weight) if fbn.weight.requires_grad else None
                dbet = self._grad_of(fbn.bias) if fbn.bias.requires_grad else None
                L.check(lib.creid_conv2d_wgrad_partials_bnfin(C.byref(d), L.ptr(a_in), L.ptr(dy), L.ptr(ws), nbytes, self.dt,
                                                              L.ptr(fpart), lib.creid_bn2d_bwd_rows(fM), fu.cout, fM,
                                                              L.ptr(mean), L.ptr(invstd), L.ptr(fbn.weight), L.ptr(sums),
                                                              L.ptr(dgam), L.ptr(dbet), st), "conv2d_wgrad_partials_bnfin")
                self._bn_sums[id(fu)] = sums
            else:
                L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(a_in), L.ptr(dy), L.ptr(ws), nbytes, self.dt, st),
                        "conv2d_wgrad_partials")
            self._wred_pending.append((d, gw, ws, nbytes))
            assert len(self._wred_pending) <= 3
            return
        if not self.reduce_stream:
            ws = self._workspace(nbytes)
            L.check(lib.creid_conv2d_wgrad_nhwc(C.byref(d), L.ptr(a_in), L.ptr(dy), L.ptr(gw), 1, L.ptr(ws), nbytes, self.dt,
                                                st), "conv2d_wgrad")
            return
        # CREID_REDUCE_STREAM=1: partial tiles on the main stream, the short split reduce on a second stream beside
        # the data gradient that follows; two workspaces alternate so the next wgrad never waits for this reduce
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.device)
        k = self._ws_flip = 1 - getattr(self, "_ws_flip", 0)
        if self._ws2[k] is None or self._ws2[k].numel() < nbytes:
            self._keep.append(self._ws2[k])
            self._ws2[k] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=self.device)
        ws = self._ws2[k]
        main.wait_stream(self._side)            # the reduce that last used this workspace (two wgrads ago) is done
        L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(a_in), L.ptr(dy), L.ptr(ws), nbytes, self.dt, st),
                "conv2d_wgrad_partials")
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            L.check(lib.creid_conv2d_wgrad_reduce(C.byref(d), L.ptr(gw), 1, L.ptr(ws), nbytes, self.dt, L.stream()),
                    "conv2d_wgrad_reduce")

    def _dgrad(self, u, dy, B, H, W, add_src=None, bnred=None, stat_image_rows=0, add_src_stride=1, add_mask=None):
        """Data gradient.  bnred = (x, act, mean, invstd) of the BN layer that consumes the result: its column
        reduction is fused into the epilogue (bf16) and the partials are returned.  stat_image_rows = H*W when
        mean / invstd are per-(image, channel) (IBN), 0 for BatchNorm."""
        lib, st = L.lib(), L.stream()
        d, _, _ = _desc(B, H, W, u.cin, u.cout, u.k, u.stride, u.pad)
        M = B * H * W
        dx = self._empty(M, u.cin)
        fuse_bn = bnred is not None and self.fuse_bn_reduce
        carry = len(self._wred_pending) >= (2 if self.fin_with_wred else 1)     # "wred" mode keeps one job for the finalize launch
        if carry or fuse_bn or add_mask is not None:
            rd, rgw, rws, rbytes = self._wred_pending.pop(0) if carry else (None, None, None, 0)
            x, act, mean, invstd = bnred if fuse_bn else (None, None, None, None)
            mask = getattr(act, "_relu_mask", None) if act is not None else None
            part = self._empty(lib.creid_bn2d_bwd_rows(M) * 2, u.cin, dtype=torch.float32) if fuse_bn else None
            L.check(lib.creid_conv2d_dgrad_fused_nhwc(C.byref(d), L.ptr(dy), L.ptr(u.w_crsk), L.ptr(dx), L.ptr(add_src),
                                                      add_src_stride, L.ptr(add_mask), L.ptr(x), L.ptr(act), L.ptr(mask), L.ptr(mean),
                                                      L.ptr(invstd), L.ptr(part), stat_image_rows if fuse_bn else 0,
                                                      C.byref(rd) if rd is not None else None, L.ptr(rgw), 1,
                                                      L.ptr(rws), rbytes, self.dt, st), "conv2d_dgrad_fused")
            return dx, part
        L.check(lib.creid_conv2d_dgrad_nhwc(C.byref(d), L.ptr(dy), L.ptr(u.w_crsk), L.ptr(dx), L.ptr(add_src), self.dt, st),
                "conv2d_dgrad")
        return dx, None

    def first_bn_bwd_operands(self):
        """What creid_ctl_heads_fused needs to produce the column sums of the FIRST BatchNorm backward (the last bottleneck's bn3)
        while it writes g: (x3, ReLU bits of the block output, mean, invstd, partial-rows buffer), or None when that reduction
        keeps its own launch (fp32 engine, no mask bits, 128-row tiles that straddle images in the general case are fine)."""
        sv = self.saved
        if self.basic or sv is None or not sv.get("training") or not (self.fuse_bn_reduce and self.drop_gm and self.relu_bitmask):
            return None
        if os.environ.get("CREID_HEADS_BNRED", "1") != "1":
            return None
        s = sv["blocks"][-1]
        mask = getattr(s["a3"], "_relu_mask", None)
        if mask is None or s["x3"].shape[1] % 256 != 0:
            return None
        M = s["x3"].shape[0]
        part = self._empty(L.lib().creid_bn2d_bwd_rows(M) * 2, s["x3"].shape[1], dtype=torch.float32)
        return s["x3"], mask, s["m3"], s["i3"], part

    def backward(self, dfeat: torch.Tensor, g: torch.Tensor = None, part3: torch.Tensor = None):
        """Accumulates parameter gradients into `.grad` (fp32, reference layouts).  `self.on_group_done(k)`, if set,
        is called after the last kernel of layer k (4, 3, 2, 1) has been enqueued -- every gradient of that layer is
        then final in stream order (the data-parallel bucketed all-reduce hangs on it, parallel.py).
        g (optional, [B * h * w, 2048] in the compute dtype): the gradient of the final feature map, already pooled back and (f16)
        loss-scaled -- the last launch of creid_ctl_heads_fused writes it; `dfeat` is ignored then.  part3 (optional): the column
        sums of the deepest bn3 backward, produced by that same launch."""
        sv = self.saved
        assert sv is not None and sv["training"], "backward() needs a training-mode forward first"
        lib, st = L.lib(), L.stream()
        self._bn_sums.clear()
        B = sv["B"]
        h, w = sv["final"]
        if g is None:
            dfeat = dfeat.contiguous().float()
            if self.loss_scaler is not None and self.dtype == torch.float16:
                dfeat = self.loss_scaler.scale_(dfeat)        # every f16 gradient tensor / backbone parameter gradient below is scaled
            g = self._empty(B * h * w, self.cout)
            L.check(lib.creid_gap_bwd(L.ptr(dfeat), B, h * w, self.cout, self.dt, L.ptr(g), st), "gap_bwd")
        else:
            assert g.dtype == self.dtype and tuple(g.shape) == (B * h * w, self.cout) and g.is_contiguous()
        blocks = list(zip(self.blocks, sv["blocks"]))
        if self.basic:
            g = self._backward_basic(blocks, g, B)
            blocks = []
        # part3: bn3 partials of the CURRENT block, produced by the previous (deeper) block -- for the deepest block by the heads'
        # last launch when the caller hands them in (first_bn_bwd_operands)
        for bi in range(len(blocks) - 1, -1, -1):
            b, s = blocks[bi]
            prev = blocks[bi - 1] if bi > 0 else None          # the block whose output gradient we produce
            M3 = B * s["h2"] * s["w2"]
            # the block's incoming gradient g is masked by the final ReLU on three paths (bn3, the residual add, the
            # downsample BN).  With the mask as bits every consumer applies it itself and the masked copy `gm` is never
            # written; otherwise bn3's backward writes it once.
            m3 = getattr(s["a3"], "_relu_mask", None) if (self.fuse_bn_reduce and self.drop_gm) else None
            # downsample blocks: bn3's apply pass also sums the columns the downsample branch's BatchNorm backward needs (same
            # masked gradient, one read of it instead of two and one launch less; CREID_DS_REDUCE2=0: separate launches)
            ds_part = None
            if m3 is not None and b["ds"] is not None and self.ds_reduce2 and not self.fin_with_wred:
                dx3, ds_part = self._bn_bwd(b["c3"], s["x3"], g, s["a3"], s["m3"], s["i3"], M3, part=part3,
                                            reduce2=(s["xd"], s["md"], s["idd"]))
                gm = g
            else:
                dx3, gm = self._bn_bwd(b["c3"], s["x3"], g, s["a3"], s["m3"], s["i3"], M3, want_gm=m3 is None, part=part3)
                if m3 is not None:
                    gm = g
            # per convolution: data gradient first (it feeds the dependent chain dgrad -> BN finalize -> BN apply -> dgrad),
            # then the weight gradient, which is off that chain and carries the finalize of the BatchNorm whose column sums
            # the data gradient just produced (and, via the pending queue, gets its own split reduction carried by the
            # NEXT data gradient)
            wfirst = self.wgrad_first

            def wg(u, a, dy, h_, w_, fin=None, early=False, first=None):   # the call site that matches the launch order issues it
                f = wfirst if first is None else first
                if early == f:
                    self._wgrad(u, a, dy, B, h_, w_, fin=None if f else fin)

            wg(b["c3"], s["a2"], dx3, s["h2"], s["w2"], early=True)
            da2, p2 = self._dgrad(b["c3"], dx3, B, s["h2"], s["w2"], bnred=(s["x2"], s["a2"], s["m2"], s["i2"]))
            wg(b["c3"], s["a2"], dx3, s["h2"], s["w2"], fin=(b["c2"], p2, M3, s["m2"], s["i2"]))
            dx2, _ = self._bn_bwd(b["c2"], s["x2"], da2, s["a2"], s["m2"], s["i2"], M3, part=p2, dry=bool(self._apply_dry & 2))
            ibn1 = b["c1"].ibn is not None
            hw1 = s["h1"] * s["w1"]
            ibn_fused = ibn1 and hw1 % 128 == 0          # per-image statistics: the 128-row tiles must not straddle images
            wg(b["c2"], s["a1"], dx2, s["h1"], s["w1"], early=True)
            da1, p1 = self._dgrad(b["c2"], dx2, B, s["h1"], s["w1"],
                                  bnred=None if (ibn1 and not ibn_fused) else (s["x1"], s["a1"], s["m1"], s["i1"]),
                                  stat_image_rows=hw1 if ibn_fused else 0)
            M1 = B * s["h1"] * s["w1"]
            wg(b["c2"], s["a1"], dx2, s["h1"], s["w1"], fin=None if ibn1 else (b["c1"], p1, M1, s["m1"], s["i1"]))
            if ibn1:
                dx1, _ = self._ibn_bwd(b["c1"], s["x1"], da1, s["a1"], s["m1"], s["i1"], B, hw1, part=p1)
            else:
                dx1, _ = self._bn_bwd(b["c1"], s["x1"], da1, s["a1"], s["m1"], s["i1"], M1, part=p1, dry=bool(self._apply_dry & 2))
            nxt = None if prev is None else (prev[1]["x3"], prev[1]["a3"], prev[1]["m3"], prev[1]["i3"])
            # the c1 data gradient writes the block-input gradient (the widest tensor of the block) that the previous block's
            # bn3 apply reads next; when it is larger than the carrier threshold the weight gradient goes first (its operand
            # traffic would push that tensor out of the Infinity Cache) and bn3's finalize keeps its own launch
            c1_first = wfirst or B * s["hin"] * s["win"] * b["c1"].cin * 2 > self.carrier_max_bytes
            wg(b["c1"], s["a_in"], dx1, s["hin"], s["win"], early=True, first=c1_first)
            if b["ds"] is not None:
                dxd, _ = self._bn_bwd(b["ds"], s["xd"], gm, None, s["md"], s["idd"], M3, mask=m3, part=ds_part)
                dsu = b["ds"]
                if (dsu.stride == 2 and dsu.k == 1 and nxt is not None and self.fuse_bn_reduce
                        and s["hin"] % 2 == 0 and s["win"] % 2 == 0):
                    # stride-2 1x1 downsample: its data gradient lives on the even pixels only -- compute it as a
                    # plain 1x1 GEMM over the OUTPUT grid and let the c1 dgrad epilogue scatter-add it
                    from types import SimpleNamespace
                    shim = SimpleNamespace(cin=dsu.cin, cout=dsu.cout, k=1, stride=1, pad=0, w_crsk=dsu.w_crsk)
                    tmp, _ = self._dgrad(shim, dxd, B, s["h2"], s["w2"])        # (also carries a pending split reduction)
                    self._wgrad(b["ds"], s["a_in"], dxd, B, s["hin"], s["win"])
                    g, part3 = self._dgrad(b["c1"], dx1, B, s["hin"], s["win"], add_src=tmp, bnred=nxt, add_src_stride=2)
                else:
                    tmp, _ = self._dgrad(dsu, dxd, B, s["hin"], s["win"])
                    self._wgrad(b["ds"], s["a_in"], dxd, B, s["hin"], s["win"])
                    g, part3 = self._dgrad(b["c1"], dx1, B, s["hin"], s["win"], add_src=tmp, bnred=nxt)
            else:
                g, part3 = self._dgrad(b["c1"], dx1, B, s["hin"], s["win"], add_src=gm, bnred=nxt, add_mask=m3)
            Min = B * s["hin"] * s["win"]
            wg(b["c1"], s["a_in"], dx1, s["hin"], s["win"], first=c1_first,
               fin=None if (prev is None or part3 is None) else (prev[0]["c3"], part3, Min, prev[1]["m3"], prev[1]["i3"]))
            if self.on_group_done is not None and bi in self._group_first:
                self._flush_wred()                       # the layer's last split reduction: nothing left to carry it
                if self.wgrad_stream or self.reduce_stream:
                    self._join_side()                    # side-stream weight gradients of this group must be ordered before the hook
                self.on_group_done(self._group_first[bi])
        # stem
        xpad, x0, y0, mean0, invstd0, idx0 = sv["stem"]
        H, W = sv["H"], sv["W"]
        H1, W1 = H // 2, W // 2
        if self.stem_fuse_pool_bwd and not self.net.stem_relu:
            # the max-pool gradient is gathered inside the BatchNorm backward passes, the full-resolution dy0 is never written
            bn, M0 = self.stem.bn, B * H1 * W1
            part0 = self._empty(lib.creid_bn2d_bwd_rows(M0) * 2, 64, dtype=torch.float32)
            sums0 = self._empty(3, 64, dtype=torch.float32)
            dx0 = self._empty(M0, 64)
            dgam = self._grad_of(bn.weight) if bn.weight.requires_grad else None
            dbet = self._grad_of(bn.bias) if bn.bias.requires_grad else None
            L.check(lib.creid_bn2d_bwd_pooled(L.ptr(x0), L.ptr(g), L.ptr(idx0), B, H1, W1, None, L.ptr(mean0), L.ptr(invstd0),
                                              L.ptr(bn.weight), 64, self.dt, L.ptr(part0), L.ptr(sums0), L.ptr(dgam), L.ptr(dbet),
                                              L.ptr(dx0), st), "bn2d_bwd_pooled")
        else:
            dy0 = self._empty(B * H1 * W1, 64)
            L.check(lib.creid_maxpool3x3s2_bwd(L.ptr(g), L.ptr(idx0), B, H1, W1, 64, self.dt, L.ptr(dy0), st), "maxpool_bwd")
            dx0, _ = self._bn_bwd(self.stem, x0, dy0, y0 if self.net.stem_relu else None, mean0, invstd0, B * H1 * W1)
        if self.stem.conv.weight.requires_grad:
            with (self._fork_side(xpad, dx0) if self.wgrad_stream else contextlib.nullcontext()):
                nbytes = lib.creid_stem_conv_wgrad_workspace_bytes(B, H, W, self.dt)
                ws = self._workspace(nbytes)
                L.check(lib.creid_stem_conv_wgrad(B, H, W, L.ptr(xpad), L.ptr(dx0),
                                                  L.ptr(self._grad_of(self.stem.conv.weight)), 1, L.ptr(ws), nbytes,
                                                  self.dt, L.stream()), "stem_conv_wgrad")
        self._flush_wred()
        self._join_side()
        self.saved = None

    def _backward_basic(self, blocks, g, B):
        """Backward of the BasicBlock networks (resnet18 / resnet34), block by block in reverse: bn2 backward (it also writes the
        ReLU-masked gradient the shortcut needs), conv2's weight and data gradient, bn1 backward, conv1's weight gradient, the
        downsample branch (BatchNorm backward, weight gradient, data gradient) and conv1's data gradient with the shortcut's
        gradient added in its epilogue.  Plain launches of the same kernels as the bottleneck schedule; split reductions of the
        weight gradients ride in the following data gradients as there.  Returns the gradient w.r.t. the max-pool output."""
        for bi in range(len(blocks) - 1, -1, -1):
            b, s = blocks[bi]
            M2 = B * s["h2"] * s["w2"]
            dx2, gm = self._bn_bwd(b["c2"], s["x2"], g, s["a2"], s["m2"], s["i2"], M2, want_gm=True)
            self._wgrad(b["c2"], s["a1"], dx2, B, s["h1"], s["w1"])
            da1, _ = self._dgrad(b["c2"], dx2, B, s["h1"], s["w1"])
            dx1, _ = self._bn_bwd(b["c1"], s["x1"], da1, s["a1"], s["m1"], s["i1"], B * s["h1"] * s["w1"])
            self._wgrad(b["c1"], s["a_in"], dx1, B, s["hin"], s["win"])
            if b["ds"] is not None:
                dxd, _ = self._bn_bwd(b["ds"], s["xd"], gm, None, s["md"], s["idd"], M2)
                self._wgrad(b["ds"], s["a_in"], dxd, B, s["hin"], s["win"])
                short, _ = self._dgrad(b["ds"], dxd, B, s["hin"], s["win"])
            else:
                short = gm
            g, _ = self._dgrad(b["c1"], dx1, B, s["hin"], s["win"], add_src=short)
            if self.on_group_done is not None and bi in self._group_first:
                self._flush_wred()
                self.on_group_done(self._group_first[bi])
        return g

    def _flush_wred(self):
        """Split reductions that found no data-gradient launch to ride on (the last one of the backward pass and, with a
        group hook, of every layer group): the same job as the carried form, as its own launch."""
        lib, st = L.lib(), L.stream()
        while self._wred_pending:
            rd, rgw, rws, rbytes = self._wred_pending.pop(0)
            L.check(lib.creid_conv2d_wgrad_reduce_job(C.byref(rd), L.ptr(rgw), 1, L.ptr(rws), rbytes, self.dt, st),
                    "conv2d_wgrad_reduce_job")


class _BackboneFn(torch.autograd.Function):
    """Autograd boundary: the engine's backward writes parameter .grad directly."""

    @staticmethod
    def forward(ctx, x, trigger, engine, want_base_out):
        ctx.engine = engine
        base_out, feat = engine.forward(x, True, want_base_out)
        if base_out is None:
            base_out = feat.new_empty(0)
        ctx.mark_non_differentiable(base_out)
        return base_out, feat

    @staticmethod
    def backward(ctx, _g_base, g_feat):
        ctx.engine.backward(g_feat)
        return None, None, None, None
