#!/usr/bin/env python3
"""Where the drop-in BaseVAMPIRE2 module spends its time at cfg-B (or another preset) on the GPU (SURVEY 8f N3
sizing): forward + backward of the whole module with the stand-in image encoder, HIP events around
the sections of _forward_single_sweep.  usage: tools/time_backbone.py [batch] [cfg]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vampire_amd.config import PRESETS
from vampire_amd.backbone import BaseVAMPIRE2
from vampire_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
cfg_name = sys.argv[2] if len(sys.argv) > 2 else "B"
c = PRESETS[cfg_name]
dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = os.environ.get("CUDNN_BENCHMARK", "0") == "1"
kw = dict(x_bound_seg=list(c.x_bound_seg), y_bound_seg=list(c.y_bound_seg), z_bound_seg=list(c.z_bound_seg),
          x_bound_det=list(c.x_bound_det), y_bound_det=list(c.y_bound_det), z_bound_det=list(c.z_bound_det),
          d_bound=list(c.d_bound), final_dim=c.final_dim, downsample_factor=4, upsample_factor=4,
          mid_channels=c.mid_channels, output_channels=80, img_backbone_conf=dict(),
          img_neck_conf=dict(out_channels=[128] * 4), num_classes=c.num_classes, density_mode="sdf",
          sdf_bias=-1.0, cat_pos=False, cat_seg=False)
torch.manual_seed(0)
mod = BaseVAMPIRE2(**kw).to(dev)
s2e, K, ida = synthetic.camera_rig(c, B, seed=0)
bda = synthetic.bda_matrix(B)
mats = {k: v.to(dev) for k, v in dict(sensor2ego_mats=s2e[:, None], intrin_mats=K[:, None], ida_mats=ida[:, None],
                                      sensor2sensor_mats=torch.eye(4).expand(B, 1, 6, 4, 4).contiguous(),
                                      bda_mat=bda).items()}
imgs = torch.randn(B, 1, 6, 3, *c.final_dim, device=dev)

# section timers: forward hooks on the submodules (events on the current stream)
marks = []
def hook(name):
    def pre(m, i):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, "b", e))
    def post(m, i, o):
        e = torch.cuda.Event(enable_timing=True); e.record(); marks.append((name, "e", e))
    return pre, post
for name in ("img_backbone", "img_neck", "mapping_along_depth", "channel_lower", "base_conv", "density_conv",
             "seg_conv", "rgb_conv", "voxel_output", "upsample2d"):
    pre, post = hook(name)
    getattr(mod, name).register_forward_pre_hook(pre)
    getattr(mod, name).register_forward_hook(post)


def run(train=True):
    mod.zero_grad(set_to_none=True)
    out = mod(imgs, mats)
    if train:
        loss = sum(o.float().mean() for o in out if torch.is_tensor(o) and o.requires_grad)
        loss.backward()


for _ in range(3):
    run()
torch.cuda.synchronize()
n = 20
tot = {}
e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
fwd = bwd = 0.0
fl, bl = [], []
for _ in range(n):
    marks.clear()
    mod.zero_grad(set_to_none=True)
    e0.record()
    out = mod(imgs, mats)
    e1.record()
    loss = sum(o.float().mean() for o in out if torch.is_tensor(o) and o.requires_grad)
    loss.backward()
    e2.record()
    torch.cuda.synchronize()
    fwd += e0.elapsed_time(e1); bwd += e1.elapsed_time(e2)
    fl.append(e0.elapsed_time(e1)); bl.append(e1.elapsed_time(e2))
    open_ = {}
    for name, kind, e in marks:
        if kind == "b":
            open_[name] = e
        else:
            tot[name] = tot.get(name, 0.0) + open_[name].elapsed_time(e)
fl.sort(); bl.sort()
print("BaseVAMPIRE2 at cfg-" + cfg_name + ", batch %d (stand-in image encoder): forward %.2f ms, backward %.2f ms (medians of %d steps; "
      "means %.2f / %.2f, forward min %.2f max %.2f)" % (B, fl[n // 2], bl[n // 2], n, fwd / n, bwd / n, fl[0], fl[-1]))
print("forward sections (ms): " + ", ".join("%s %.2f" % (k, v / n) for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
# in-library HIP-event timer over three more steps: the HIP kernels of the layers around the path
from vampire_amd import _capi
_capi.profile_select(None)
_capi.profile_enable(True)
for _ in range(3):
    run()
torch.cuda.synchronize()
_capi.profile_enable(False)
prof = _capi.profile_read()
print("HIP kernels per step (us): " + ", ".join("%s %.0f (%d x %.0f)" % (k, ms / 3 * 1e3, n // 3, ms / n * 1e3)
                                                for k, (n, ms) in sorted(prof.items(), key=lambda kv: -kv[1][1])[:12]))

