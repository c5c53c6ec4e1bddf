#!/usr/bin/env python3
"""Time the producer / consumer glue at cfg-B on the GPU: HIP kernels vs the
reference's aten expressions on the same device, forward and forward+backward, with the HBM
roofline of each (algorithmic bytes = every tensor read / written once)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import CFG_B as cfg
from vampire_amd.ops import HotPath

dev = torch.device("cuda:0")
hp = HotPath(cfg, dev)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
g = torch.Generator().manual_seed(0)
logits = (torch.randn(B * cfg.num_cams, cfg.D, cfg.fH, cfg.fW, generator=g) * 3).to(dev)
g_depth = torch.randn(logits.shape, generator=g).to(dev)
vo = torch.randn(B, cfg.mid_channels, cfg.oZ, cfg.oY, cfg.oX, generator=g).to(dev)
vd = (torch.rand(B, 1, cfg.oZ, cfg.oY, cfg.oX, generator=g) * 3).to(dev)
g_out = torch.randn(vo.shape, generator=g).to(dev)


def softmax_hip(bwd):
    x = logits.detach().requires_grad_(bwd)
    y = hp.depth_softmax(x)
    if bwd:
        y.backward(g_depth)


def softmax_aten(bwd):
    x = logits.detach().requires_grad_(bwd)
    y = x.softmax(dim=1)
    if bwd:
        y.backward(g_depth)


def gate_hip(bwd):
    a, b = vo.detach().requires_grad_(bwd), vd.detach().requires_grad_(bwd)
    y = hp.density_gate(a, b)
    if bwd:
        y.backward(g_out)


def gate_aten(bwd):
    a, b = vo.detach().requires_grad_(bwd), vd.detach().requires_grad_(bwd)
    y = a * b.tanh()
    if bwd:
        y.backward(g_out)


# ---- the fused forms (the module's default): softmax inside the lift, gate inside the 1x1 conv ----
from vampire_amd import synthetic
from vampire_amd.geometry import lift_matrices
COUT = 80
s2e, K, ida = synthetic.camera_rig(cfg, B, jitter=2.0, seed=11)
lm = lift_matrices(s2e, K, ida, synthetic.bda_matrix(B, rot_deg=-6.0, scale=1.02)).to(dev)
_, feat = synthetic.lift_inputs(cfg, B, seed=6, device=dev)
lg5 = logits.reshape(B, cfg.num_cams, cfg.D, cfg.fH, cfg.fW)
g_vox = torch.randn(B, cfg.mid_channels, cfg.vZ, cfg.vY, cfg.vX, generator=g).to(dev)
wconv = (torch.randn(COUT, cfg.mid_channels * cfg.oZ, 1, 1, generator=g) * 0.1).to(dev)
bconv = torch.randn(COUT, generator=g).to(dev)
g_bev = torch.randn(B, COUT, cfg.oY, cfg.oX, generator=g).to(dev)


def lift_fused(bwd):
    x, f = lg5.detach().requires_grad_(bwd), feat.detach().requires_grad_(bwd)
    y = hp.lift_logits(x, f, lm)
    if bwd:
        y.backward(g_vox)


def lift_unfused(bwd):
    x, f = lg5.detach().requires_grad_(bwd), feat.detach().requires_grad_(bwd)
    y = hp.lift(x.softmax(dim=2), f, lm)
    if bwd:
        y.backward(g_vox)


def conv_fused(bwd):
    a, b, w, bs = (t.detach().requires_grad_(bwd) for t in (vo, vd, wconv, bconv))
    y = hp.gate_conv1x1(a, b, w, bs)
    if bwd:
        y.backward(g_bev)


def conv_unfused(bwd):
    a, b, w, bs = (t.detach().requires_grad_(bwd) for t in (vo, vd, wconv, bconv))
    y = torch.nn.functional.conv2d((a * b.tanh()).reshape(B, -1, cfg.oY, cfg.oX), w, bs)
    if bwd:
        y.backward(g_bev)


def timeit(fn, *a, n=50):
    for _ in range(10):
        fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn(*a)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


nl, nv, nd = logits.numel() * 4, vo.numel() * 4, vd.numel() * 4
alg = {"softmax": (2 * nl, 2 * nl + 3 * nl), "gate": (2 * nv + nd, 2 * nv + nd + 3 * nv + 2 * nd)}
print("cfg-B, batch %d: depth logits %.1f MB, voxel_output %.1f MB" % (B, nl / 1e6, nv / 1e6))
for name, h, a in (("softmax", softmax_hip, softmax_aten), ("gate", gate_hip, gate_aten)):
    for impl, fn in (("hip", h), ("aten", a)):
        f, fb = timeit(fn, False), timeit(fn, True)
        print("%-8s %-5s forward %6.1f us (%.2f TB/s of %.0f MB)   forward+backward %6.1f us (%.2f TB/s of %.0f MB)" %
              (name, impl, f, alg[name][0] / f / 1e6, alg[name][0] / 1e6, fb, alg[name][1] / fb / 1e6,
               alg[name][1] / 1e6))

def graph_time(fn, *a, n=50):
    """Device time: the call captured into a HIP graph (no host gaps between the kernels)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            fn(*a)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        fn(*a)
    for _ in range(5):
        gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


torch.backends.cudnn.benchmark = True
print("fused forms (module default) against the reference's chain on the same device:")
for name, h, a in (("softmax+lift", lift_fused, lift_unfused), ("gate+conv1x1", conv_fused, conv_unfused)):
    for impl, fn in (("fused", h), ("chain", a)):
        f, fb = timeit(fn, False), timeit(fn, True)
        gf, gfb = graph_time(fn, False), graph_time(fn, True)
        print("%-13s %-6s eager: forward %7.1f us, forward+backward %7.1f us   HIP graph (device time): %7.1f / %7.1f us" %
              (name, impl, f, fb, gf, gfb))
