import sys, dataclasses, torch, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,R); sys.path.insert(0,os.path.join(R,'tests'))
from conftest import load_golden
from vampire_amd.config import CFG_TINY
from vampire_amd.ops import HotPath
dev=torch.device('cuda:0')
NAMES = ["rgb_preds", "seg_logits_preds", "depth_preds", "bev_rgb_preds","bev_seg_logits_preds", "bev_height_preds", "voxel_density", "voxel_output"]
for tag in ("nonaffine","smooth"):
    r=load_golden(f"tiny_render_{tag}.npz")
    cfg=dataclasses.replace(CFG_TINY,density_mode="sdf",cat_seg=False)
    hp=HotPath(cfg,dev)
    res={}
    for direct in (True,False):
        for ert in (True, False):
            hp.impl["cam_direct"]=direct; hp.impl["ert"]=ert
            vols=[r[k].to(dev) for k in ("density_feature","semantic_logits","base","rgb")]
            with torch.no_grad():
                outs=hp.render(*vols, r["beta"].reshape(()).to(dev), render_mats=r["render_mats"].to(dev))
            e=[float((o.cpu()-r[n]).abs().max()/r[n].abs().max()) for n,o in zip(NAMES[:3],outs[:3])]
            print(tag,"direct" if direct else "planned","ert" if ert else "noert",["%.2e"%x for x in e])
