"""Can a weight gradient hide under the data gradient of the same convolution?  For each layer shape: 10 data-gradient launches
and 10 weight-gradient (partials) launches captured (a) back to back on one stream, (b) as two parallel branches of one hipGraph
(one fork, one join: no per-launch cross-queue dependency).  If (b) ~ max(d, w) there is idle capacity a merged launch could
use; if (b) ~ d + w the two kernels already saturate what they contend for.
    python tools/debug/overlap_probe.py [batch]"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L   # noqa: E402
from centroids_reid_amd.bench_train import conv_shapes    # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
seen = {}
for sh in conv_shapes(B, 256, 128):
    seen[sh] = seen.get(sh, 0) + 1


def graph_time(build, reps=3):
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        build(side)
        with torch.cuda.graph(g, stream=side):
            build(side)
    torch.cuda.current_stream().wait_stream(side)
    best = 1e9
    for _ in range(reps):
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best * 1e3 / 10


tot = [0.0, 0.0, 0.0, 0.0]
print(f"B={B}: us per (dgrad, wgrad) pair: dgrad alone, wgrad alone, sequential, two graph branches")
for (cin, cout, k, s, h, w), cnt in seen.items():
    pad = k // 2
    x = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    wt = torch.randn((cout, cin, k, k), device="cuda") / (cin * k * k) ** 0.5
    krsc, crsk = ly.weight_prep(wt, torch.bfloat16)
    y = ly.conv2d_fwd(x, krsc, s, pad)
    d, _, _ = ly.conv_desc(B, h, w, cin, cout, k, s, pad)
    nbytes = lib.creid_conv2d_wgrad_workspace_bytes(C.byref(d), L.BF16)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    dx = torch.empty((B, h, w, cin), dtype=torch.bfloat16, device="cuda")

    def dgrad():
        L.check(lib.creid_conv2d_dgrad_nhwc(C.byref(d), L.ptr(y), L.ptr(crsk), L.ptr(dx), None, L.BF16, L.stream()), "d")

    def wgrad():
        L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(x), L.ptr(y), L.ptr(ws), nbytes, L.BF16, L.stream()), "w")

    def only_d(st):
        for _ in range(10): dgrad()

    def only_w(st):
        for _ in range(10): wgrad()

    def seq(st):
        for _ in range(10): dgrad(); wgrad()

    def par(st):
        other = torch.cuda.Stream()
        other.wait_stream(st)
        for _ in range(10): dgrad()
        with torch.cuda.stream(other):
            for _ in range(10): wgrad()
        st.wait_stream(other)

    td, tw, ts, tp = graph_time(only_d), graph_time(only_w), graph_time(seq), graph_time(par)
    for i, v in enumerate((td, tw, ts, tp)):
        tot[i] += v * cnt
    print(f"{cin:4d}->{cout:4d} k{k} s{s} {h:3d}x{w:<3d} x{cnt}  d {td:6.1f}  w {tw:6.1f}  seq {ts:6.1f}  par {tp:6.1f}  par/seq {tp / ts:4.2f}  max/seq {max(td, tw) / ts:4.2f}")
print(f"per step (us): dgrad {tot[0]:.0f}  wgrad {tot[1]:.0f}  sequential {tot[2]:.0f}  parallel branches {tot[3]:.0f}")
