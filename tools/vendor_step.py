"""Vendor yardstick, OUTSIDE the product: the same two workloads bench.py times, on stock PyTorch-ROCm
modules -- nn.Conv2d / nn.BatchNorm2d through MIOpen, nn.Linear through hipBLASLt, channels_last, bf16 autocast, torch.optim.Adam
(fused where available) -- on the same box.  Nothing here is imported by the package or the tests; bench.py runs this file in a
child process and prints the result as `extra.vendor_yardstick` when it finishes inside its time budget.

  step : ResNet50 (last_stride 1, no stem ReLU: modelling/backbones/resnet.py) 256 x 128, B = 64, forward + backward of
         sum(GAP features^2) stand-in head (the heads are ~2 % of the step), Adam over all parameters      -> images/s
  embed: the same backbone in eval mode, B = 128, forward only                                             -> images/s
  eval : the reference's distance stage as it is written (utils/reid_metric.py:25-33: pow-sum + addmm_, fp32) on
         2228 x 17661 x 2048 device features, plus torch.argsort of the matrix (what utils/eval_reid.py:36 does in
         numpy on the host) -- the stock-GPU route to ranked indices; no CMC / AP                           -> pairs/s

    python tools/vendor_step.py [--steps 20] [--no-graph]        -> one JSON line
"""
import argparse
import json
import sys
import time

import torch
from torch import nn


class Bottleneck(nn.Module):
    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False); self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        r = x if self.downsample is None else self.downsample(x)
        o = self.relu(self.bn1(self.conv1(x)))
        o = self.relu(self.bn2(self.conv2(o)))
        return self.relu(self.bn3(self.conv3(o)) + r)


class ResNet50(nn.Module):
    def __init__(self, last_stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False); self.bn1 = nn.BatchNorm2d(64)
        self.pool = nn.MaxPool2d(3, 2, 1)
        self.inplanes = 64
        self.layers = nn.Sequential(self._make(64, 3, 1), self._make(128, 4, 2), self._make(256, 6, 2), self._make(512, 3, last_stride))

    def _make(self, planes, n, stride):
        ds = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        blocks = [Bottleneck(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * 4
        blocks += [Bottleneck(self.inplanes, planes) for _ in range(1, n)]
        return nn.Sequential(*blocks)

    def forward(self, x):
        x = self.pool(self.bn1(self.conv1(x)))
        return self.layers(x).mean((2, 3))


def timed(fn, steps, warmup, graph):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if graph:
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        with torch.cuda.graph(g):
            fn()
        run = g.replay
    else:
        run = fn
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--no-graph", action="store_true")
    a = ap.parse_args()
    dev = "cuda"
    out = {"torch": torch.__version__, "hip": torch.version.hip, "device": torch.cuda.get_device_name(0),
           "note": "stock torch-ROCm modules (MIOpen convolutions / BatchNorm, channels_last, bf16 autocast, torch.optim.Adam); "
                   "head = sum of squared pooled features (the CTL heads are ~2 % of the product's step)"}
    torch.backends.cudnn.benchmark = True
    for graph in ([False] if a.no_graph else [True, False]):
        try:
            torch.manual_seed(0)
            net = ResNet50().to(dev).to(memory_format=torch.channels_last).train()
            try:
                opt = torch.optim.Adam(net.parameters(), lr=3.5e-4, weight_decay=5e-4, fused=True, capturable=graph)
            except (TypeError, RuntimeError):
                opt = torch.optim.Adam(net.parameters(), lr=3.5e-4, weight_decay=5e-4, capturable=graph)
            x = torch.randn(64, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)

            def step():
                opt.zero_grad(set_to_none=False)
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    f = net(x)
                    loss = f.float().square().mean()
                loss.backward()
                opt.step()
            dt = timed(step, a.steps, 5, graph)
            out["step"] = {"images_per_s": 64 / dt, "ms_per_step": dt * 1e3, "batch": 64, "hip_graph": graph}
            net.eval()
            xe = torch.randn(128, 3, 256, 128, device=dev).contiguous(memory_format=torch.channels_last)

            def fwd():
                with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                    return net(xe)
            dte = timed(fwd, a.steps, 5, graph)
            out["embed"] = {"images_per_s": 128 / dte, "ms_per_step": dte * 1e3, "batch": 128, "hip_graph": graph,
                            "note": "eval-mode BatchNorm NOT folded into the convolutions (what the stock modules do)"}
            break
        except Exception as e:  # noqa: BLE001  (a capture problem of the stock stack: fall back to eager launches)
            out.setdefault("errors", []).append(f"graph={graph}: {type(e).__name__}: {str(e)[:200]}")
            torch.cuda.synchronize()
    try:
        nq, ng, D = 2228, 17661, 2048
        gen = torch.Generator(device=dev).manual_seed(0)
        feats = torch.randn((nq + ng, D), generator=gen, device=dev)

        def dist_only():
            f = torch.nn.functional.normalize(feats, dim=1, p=2)
            qf, gf = f[:nq], f[nq:]
            d = torch.pow(qf, 2).sum(dim=1, keepdim=True).expand(nq, ng) + torch.pow(gf, 2).sum(dim=1, keepdim=True).expand(ng, nq).t()
            d.addmm_(qf, gf.t(), beta=1, alpha=-2)
            return d

        def dist_rank():
            return torch.argsort(dist_only(), dim=1)
        td = timed(dist_only, 10, 3, False)
        tr = timed(dist_rank, 5, 2, False)
        out["eval"] = {"pairs_per_s_to_ranked_indices": nq * ng / tr, "ms_normalise_and_distance_matrix": td * 1e3,
                       "ms_to_ranked_indices": tr * 1e3, "shape": [nq, ng, D],
                       "note": "normalize + pow-sum + addmm_ (hipBLASLt fp32) + torch.argsort; the reference ranks on the host in numpy"}
    except Exception as e:  # noqa: BLE001
        out.setdefault("errors", []).append(f"eval: {type(e).__name__}: {str(e)[:200]}")
    print(json.dumps(out))
    return 0 if "step" in out else 1


if __name__ == "__main__":
    sys.exit(main())
