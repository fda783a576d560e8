"""Can a convolution's weight gradient (MFMA / LDS-bound, off the dependent chain) hide the BatchNorm-backward apply of the
PREVIOUS layer (HBM-bound, no LDS), which is independent of it once the data gradient has run?  
For every (wgrad_i, bn_bwd_apply_{i-1}) pair of the ResNet50 backward at the training batch: 10 launches of each captured
(a) alone, (b) back to back on one stream, (c) as two PARALLEL branches of one hipGraph (one fork, one join) -- the upper
bound of what a fused launch with block-index role dispatch could reach.
    python tools/debug/bwd_overlap_probe.py [batch]  -> table for profiles/r05_bwd_overlap.md"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from centroids_reid_amd import layers as ly, _lib as L   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
lib = L.lib()
LAYERS, PLANES, STRIDES = (3, 4, 6, 3), (64, 128, 256, 512), (1, 2, 2, 1)


def pairs(H=256, W=128):
    """(label, wgrad conv (cin, cout, k, stride, h, w), apply (M, C)) in backward order, with multiplicities."""
    out = {}
    h, w, inpl = H // 4, W // 4, 64
    first = True
    for n, pl, st in zip(LAYERS, PLANES, STRIDES):
        for b in range(n):
            s = st if b == 0 else 1
            h2, w2 = h // s, w // s
            M3 = B * h2 * w2
            items = [("c3|bn2", (pl, pl * 4, 1, 1, h2, w2), (M3, pl)),
                     ("c2|bn1", (pl, pl, 3, s, h, w), (B * h * w, pl))]
            if not first:
                items.append(("c1|prev bn3", (inpl, pl, 1, 1, h, w), (B * h * w, inpl)))
            for it in items:
                out[it] = out.get(it, 0) + 1
            first = False
            h, w, inpl = h2, w2, pl * 4
    return out


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


tot = [0.0] * 4
print(f"B={B}: us per (wgrad_i, bn_bwd_apply_(i-1)) pair: wgrad alone, apply alone, sequential, two graph branches")
for (label, (cin, cout, k, s, h, w), (M, Cc)), cnt in pairs().items():
    pad = k // 2
    a_in = torch.randn((B, h, w, cin), device="cuda").to(torch.bfloat16)
    d, oh, ow = ly.conv_desc(B, h, w, cin, cout, k, s, pad)
    dy = torch.randn((B, oh, ow, cout), device="cuda").to(torch.bfloat16)
    nbytes = lib.creid_conv2d_wgrad_workspace_bytes(C.byref(d), L.BF16)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device="cuda")
    x = torch.randn((M, Cc), device="cuda").to(torch.bfloat16)
    g = torch.randn((M, Cc), device="cuda").to(torch.bfloat16)
    mask = torch.randint(0, 256, (M * Cc // 8,), dtype=torch.uint8, device="cuda")
    mean = torch.zeros(Cc, device="cuda"); invstd = torch.ones(Cc, device="cuda"); gamma = torch.ones(Cc, device="cuda")
    sums = torch.randn(3, Cc, device="cuda")
    dx = torch.empty_like(x)

    def wgrad():
        L.check(lib.creid_conv2d_wgrad_partials(C.byref(d), L.ptr(a_in), L.ptr(dy), L.ptr(ws), nbytes, L.BF16, L.stream()), "w")

    def apply():
        L.check(lib.creid_bn2d_bwd_mask(L.ptr(x), L.ptr(g), None, L.ptr(mask), L.ptr(mean), L.ptr(invstd), L.ptr(gamma), M, Cc,
                                        L.BF16, None, 2, L.ptr(sums), None, None, L.ptr(dx), None, L.stream()), "a")

    def only_w(st):
        for _ in range(10): wgrad()

    def only_a(st):
        for _ in range(10): apply()

    def seq(st):
        for _ in range(10): wgrad(); apply()

    def par(st):
        other = torch.cuda.Stream()
        other.wait_stream(st)
        for _ in range(10): wgrad()
        with torch.cuda.stream(other):
            for _ in range(10): apply()
        st.wait_stream(other)

    tw, ta, ts, tp = graph_time(only_w), graph_time(only_a), graph_time(seq), graph_time(par)
    for i, v in enumerate((tw, ta, ts, tp)):
        tot[i] += v * cnt
    print(f"{label:12s} w {cin:4d}->{cout:4d} k{k} s{s} {h:3d}x{w:<3d} | a {M:6d}x{Cc:<4d} x{cnt}  w {tw:6.1f}  a {ta:6.1f}  seq {ts:6.1f}"
          f"  par {tp:6.1f}  par/seq {tp / ts:4.2f}  max/seq {max(tw, ta) / ts:4.2f}", flush=True)
print(f"per step (us): wgrad {tot[0]:.0f}  apply {tot[1]:.0f}  sequential {tot[2]:.0f}  parallel branches {tot[3]:.0f}"
      f"  par/seq {tot[3] / tot[2]:.3f}")
