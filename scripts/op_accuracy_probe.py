"""Where does the fp32 step lose accuracy against an fp64 evaluation?  Op-level probes on identical inputs:
BatchNorm backward (training mode, ReLU in front) and the 3x3 conv's weight / data gradients, HIP vs CPU-ATen fp32,
both measured against CPU fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from neural_sound_generation_amd import ops

torch.manual_seed(0)
dev = "cuda:0"
def rel(a, b): return ((a.double().cpu() - b).norm() / b.norm()).item()

B, H, W, C = 1, 20, 256, 256
M = B * H * W
x = torch.randn(M, C)
gamma, beta = torch.rand(C) + 0.5, torch.randn(C) * 0.1
# a gradient that is mostly "mean + multiple of xhat" per channel: BatchNorm's backward cancels most of it
mean, var = x.mean(0), x.var(0, unbiased=False)
xhat = (x - mean) / torch.sqrt(var + 1e-5)
dy = 1.0 + 0.5 * xhat + 1e-3 * torch.randn(M, C)
def bn_ref(x, dy, gamma, beta, dt):
    x, dy, gamma, beta = (t.to(dt).clone().requires_grad_(t is not dy) for t in (x, dy, gamma, beta))
    y = F.relu(F.batch_norm(x.view(B, H, W, C).permute(0, 3, 1, 2), None, None, gamma, beta, True, 0.1, 1e-5))
    y.backward(dy.view(B, H, W, C).permute(0, 3, 1, 2))
    return x.grad, gamma.grad, beta.grad
dx64, dg64, db64 = bn_ref(x, dy, gamma, beta, torch.float64)
dx32, dg32, db32 = bn_ref(x, dy, gamma, beta, torch.float32)
xg = x.to(dev).view(B, H, W, C).contiguous()
m, i = ops.bn_stats(xg, C)
dxg, dgg, dbg = ops.bn_backward(xg, None, dy.to(dev).view(B, H, W, C).contiguous(), m, i, gamma.to(dev), relu_beta=beta.to(dev))
print("bn_backward  dx: hip %.2e cpu32 %.2e | dgamma hip %.2e cpu32 %.2e | dbeta hip %.2e cpu32 %.2e" % (
    rel(dxg.view(M, C), dx64), rel(dx32, dx64), rel(dgg, dg64), rel(dg32, dg64), rel(dbg, db64), rel(db32, db64)))
mean64 = x.double().mean(0); var64 = x.double().var(0, unbiased=False)
print("bn_stats mean: hip %.2e  invstd: hip %.2e" % (rel(m, mean64), rel(i, 1 / torch.sqrt(var64 + 1e-5))))

# 3x3 conv gradients with a zero-mean, xhat-orthogonal dy (what BatchNorm's backward hands the conv)
xin = torch.relu(torch.randn(B, C, H, W))
w = torch.randn(C, C, 3, 3) * 0.02
dh = dx64.float().view(B, H, W, C).permute(0, 3, 1, 2).contiguous()
def conv_ref(dt):
    xi, ww = xin.to(dt).clone().requires_grad_(True), w.to(dt).clone().requires_grad_(True)
    F.conv2d(xi, ww, None, 1, 1).backward(dh.to(dt))
    return xi.grad, ww.grad
gx64, gw64 = conv_ref(torch.float64)
gx32, gw32 = conv_ref(torch.float32)
d = ops.conv_desc(B, H, W, C, C, 3, 1, 1)
wf, wd = ops.pack_weights(d, w.to(dev))
xn = xin.permute(0, 2, 3, 1).contiguous().to(dev)
dhn = dh.permute(0, 2, 3, 1).contiguous().to(dev)
gw, _ = ops.conv_wgrad(d, xn, dhn, w.shape, want_bias=False)
gx = ops.conv_dgrad(d, dhn, wd)
print("conv3x3 wgrad: hip %.2e cpu32 %.2e | dgrad: hip %.2e cpu32 %.2e" % (
    rel(gw, gw64), rel(gw32, gw64), rel(gx.permute(0, 3, 1, 2), gx64), rel(gx32, gx64)))
y64 = F.conv2d(xin.double(), w.double(), None, 1, 1)
print("conv3x3 fwd: hip %.2e cpu32 %.2e" % (rel(ops.conv_forward(d, xn, wf, None).permute(0, 3, 1, 2), y64), rel(F.conv2d(xin, w, None, 1, 1), y64)))
