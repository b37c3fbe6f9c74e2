"""Does an MFMA-bound weight gradient share the chip with an HBM-bound pass?  Times (a) the two back to back on one stream,
(b) the same launches on two streams, for a 3x3 weight gradient beside bn_backward_apply / a 3x3 data gradient.
Eager launches (no graph), HIP events on the main stream around a join.  Diagnostics only."""
import torch
from neural_sound_generation_amd import ops

dev = torch.device("cuda:0")
B, H, W, D = 128, 20, 256, 128
torch.manual_seed(0)
x = torch.randn(B, H, W, D, device=dev).relu().to(torch.bfloat16)
dh = torch.randn(B, H, W, D, device=dev).to(torch.bfloat16)
h1 = torch.randn(B, H, W, D, device=dev).to(torch.bfloat16)
w = torch.randn(D, D, 3, 3, device=dev) * 0.05
d1 = ops.conv_desc(B, H, W, D, D, 3, 1, 1, dtype=torch.bfloat16)
wf, wd = ops.pack_weights(d1, w)
gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
mean = torch.zeros(D, device=dev); invstd = torch.ones(D, device=dev)
dg = torch.zeros(D, device=dev); db = torch.zeros(D, device=dev)
side = torch.cuda.Stream(device=dev)


def wgrad():
    ops.conv_wgrad(d1, x, dh, w.shape, want_bias=False)


def hbm_pass():
    ops.bn_backward_apply(h1, dh, mean, invstd, gamma, dg, db, relu_beta=beta)


def dgrad():
    ops.conv_dgrad(d1, dh, wd, add=dh, relu_x=x)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def both(main_fn, reps_main=1):
    def f():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            wgrad()
        for _ in range(reps_main):
            main_fn()
        cur.wait_stream(side)
    return f


with torch.cuda.stream(side):
    wgrad()
torch.cuda.synchronize()
tw = timeit(wgrad)
th = timeit(hbm_pass)
td = timeit(dgrad)
print(f"alone: wgrad {tw:.1f} us, bn_backward_apply {th:.1f} us, dgrad(add+mask) {td:.1f} us")
print(f"wgrad || 2 x bn_backward_apply: {timeit(both(hbm_pass, 2)):.1f} us  (serial {tw + 2 * th:.1f})")
print(f"wgrad || dgrad: {timeit(both(dgrad)):.1f} us  (serial {tw + td:.1f})")


def graphed(fn):
    gs = torch.cuda.Stream(device=dev)
    gs.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(gs):
        fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=gs):
        for _ in range(5):
            fn()
    return lambda: g.replay()


def serial2():
    wgrad(); hbm_pass(); hbm_pass()


print(f"graph of 5 x: serial {timeit(graphed(serial2)) / 5:.1f} us, two streams {timeit(graphed(both(hbm_pass, 2))) / 5:.1f} us")
