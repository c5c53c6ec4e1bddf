#!/bin/bash
# Run ON THE GPU BOX: matrix-core utilisation of the bf16 / fp16 conv3d kernels under autocast (rocprofv3 --pmc, no trace
# domains): SQ_VALU_MFMA_BUSY_CYCLES against SQ_BUSY_CYCLES / GRBM_GUI_ACTIVE per kernel.
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/conv_pmc_bf16
rm -rf $OUT; mkdir -p $OUT
cat > /tmp/conv_run_bf16.py <<PY
import os, sys, torch
sys.path.insert(0, "$ROOT")
from vampire_amd.ops import conv3d_bf16
dev = torch.device("cuda:0")
for cin, cout in ((16, 16), (32, 16)):
    x = torch.randn(1, cin, 16, 200, 200, device=dev).requires_grad_(True)
    w = (torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05).requires_grad_(True)
    up = torch.randn(1, cout, 16, 200, 200, device=dev)
    for dt in (torch.bfloat16, torch.float16):
        xb = x.detach().to(dt).requires_grad_(True)
        wb = w.detach().to(dt).requires_grad_(True)
        for _ in range(3):
            conv3d_bf16(xb, wb).backward(up.to(dt))
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 /tmp/conv_run_bf16.py > /dev/null 2> $OUT/p$i.log
done
cd $ROOT
python3 - "$OUT" <<'PY'
import collections, csv, glob, os, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for f in glob.glob(os.path.join(sys.argv[1], "p*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
        if "conv3d" not in name: continue
        k = name.split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
print("# per launch; SQ_* summed over the 1024 SIMDs, GRBM_GUI_ACTIVE over the 8 XCDs")
print("# matrix-core utilisation = (SQ_VALU_MFMA_BUSY_CYCLES / 1024) / (GRBM_GUI_ACTIVE / 8)")
for k in sorted(acc):
    d = {c: v / n[k][c] for c, v in acc[k].items()}
    u = (d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024) / max(1.0, d.get("GRBM_GUI_ACTIVE", 8) / 8)
    print("%-44s mfma util %4.1f %%  %s" % (k, 100 * u, {c: int(v) for c, v in sorted(d.items())}))
PY
