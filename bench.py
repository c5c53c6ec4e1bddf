#!/usr/bin/env python3
"""Benchmark of the lift + render hot path (driver contract: see the task statement).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--cfg B] [--batch 1]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = lift forward -> render forward (camera + BEV) -> backward of both, for
`--batch` 6-camera samples per GPU of synthetic data already resident in HBM.
Metric = BASELINE.json's "6-cam samples/sec (lift+render fwd+bwd)"; the N=1 workload is
BASELINE.json configs[1] (R50 256x704 features, 200x200x16 voxel grid, bs=1 per GPU).
Ranks shard samples (data parallel); the only collective is the DDP all-reduce of the
path's one parameter gradient (density beta) over RCCL.

Extra objects in the JSON line:
  roofline      dominant kernel of the step: algorithmic bytes per launch / HIP-event
                duration, against the 8 TB/s HBM peak (in-library event timer)
  fwd_roofline  the fused forward (north_star's 70 % target) timed as ONE unit: a HIP-event pair
                around lift + render, 20 warm-up + 100 timed iterations, median
  extra_configs cfg-A, cfg-B batch 8 and cfg-D bf16 measured in the same run (N = 1 only)
  cpu_baseline  the oracle (torch-CPU port of the reference's op sequence) timed on this
                box's host cores on a bounded sample; a reported baseline, not a target
"""
import argparse
import json
import os
import sys
import time

# The layered and multi-task legs run MIOpen convolutions (image encoder, 3-D UNet, BEV head -- none of it
# the hot path).  With torch.backends.cudnn.benchmark MIOpen's find tunes ~150 of them: 6.5 minutes on a
# fresh box (layered 57 s, multi-task 340 s; MIOPEN_FIND_MODE=FAST changes nothing -- what made one run
# look fast was a warm kernel cache on a reused box).  The default run therefore uses MIOpen's immediate
# mode for these two legs (no find; slower convolutions, said so in the line) and `--miopen-find` asks for
# the tuned numbers.
ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def _use_shipped_miopen_db():
    """profiles/miopen_db holds MIOpen's user find-db, performance db and kernel cache for the ~150 convolutions of the
    layered and the multi-task leg, made on an MI355X of this image by tools/miopen_tune.py (6.5 minutes of find).
    Pointing MIOpen at a private copy (it writes there) gives a default run the TUNED convolutions -- layered step 12
    instead of 130 ms -- without the find.  Must happen before MIOpen loads, i.e. before torch is imported; an
    environment that already names a db wins.  The copy lives in a directory named after a hash of the shipped files
    (a stale copy of another checkout is never reused), created privately (mode 0700, owned by this user: MIOpen loads
    compiled code objects from it).  Returns the directory used, or None."""
    src = os.path.join(ROOT, "profiles", "miopen_db")
    if "MIOPEN_USER_DB_PATH" in os.environ or "MIOPEN_CUSTOM_CACHE_DIR" in os.environ or not os.path.isdir(src):
        return None
    import hashlib
    import shutil
    import tempfile
    h = hashlib.sha256()
    for root, _dirs, files in sorted(os.walk(src)):
        for f in sorted(files):
            path = os.path.join(root, f)
            h.update(os.path.relpath(path, src).encode())
            h.update(str(os.path.getsize(path)).encode())
            with open(path, "rb") as fh:
                h.update(fh.read(1 << 16))
    dst = os.path.join(tempfile.gettempdir(),
                       f"vampire_miopen_db_{os.getuid()}_{os.environ.get('LOCAL_RANK', '0')}_{h.hexdigest()[:16]}")
    try:
        if os.path.isdir(dst):
            st = os.stat(dst)
            if st.st_uid != os.getuid():
                print(f"[bench] {dst} belongs to another user: MIOpen db not used", file=sys.stderr)
                return None                      # somebody else's directory under our name: do not load code from it
            if st.st_mode & 0o077:
                os.chmod(dst, 0o700)             # (ours, made by an older checkout with the source's mode bits)
        else:
            tmp = tempfile.mkdtemp(prefix="vampire_miopen_db_")          # 0700, ours
            for root, _dirs, files in os.walk(src):                       # (contents only: no permission bits, no times)
                rel = os.path.relpath(root, src)
                os.makedirs(os.path.join(tmp, rel), exist_ok=True)
                for f in files:
                    shutil.copyfile(os.path.join(root, f), os.path.join(tmp, rel, f))
            try:
                os.rename(tmp, dst)
            except OSError:                      # another rank of this user won the race: use theirs
                shutil.rmtree(tmp, ignore_errors=True)
                if not os.path.isdir(dst):
                    raise
    except OSError as e:
        print(f"[bench] shipped MIOpen db not usable ({type(e).__name__}: {e})", file=sys.stderr)
        return None
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ["MIOPEN_CUSTOM_CACHE_DIR"] = dst
    return dst


def _miopen_db_matches():
    """Do the shipped db files belong to THIS MIOpen build and device?  Their names carry both
    (gfx950100.HIP.<version>.ufdb.txt): on another build the find-db is not consulted and `find` would be a real,
    6.5-minute find -- the caller then stays in immediate mode and says so."""
    if MIOPEN_DB is None:
        return False
    try:
        ver = torch.backends.cudnn.version()                 # MIOpen's version as major * 1e6 + minor * 1e3 + patch
        want = f"{ver // 1000000}_{(ver // 1000) % 1000}_{ver % 1000}"
        names = [f for f in os.listdir(MIOPEN_DB) if f.endswith(".ufdb.txt")]
        arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
        return any(f.startswith(arch) and f".HIP.{want}" in f for f in names)
    except Exception:                                        # noqa: BLE001
        return False


MIOPEN_DB = _use_shipped_miopen_db()

import torch                                    # noqa: E402  (after the MIOpen environment)
import torch.distributed as dist                # noqa: E402

HBM_PEAK_GBS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: 8 TB/s spec


def kernel_algorithmic_bytes(cfg, B):
    """Compulsory HBM bytes per launch of each kernel: the reference-layout
    tensors the kernel logically consumes are read once and the ones it produces are written
    once, fp32; scratch traffic (packed copies, sample records, cell lists) is NOT counted, so
    `achieved` is a lower bound on what the kernel really moves.  Kernels with no compulsory
    traffic of their own (count / rank passes, scans, the heavy-voxel pass) have no entry and cannot be `dominant`."""
    P = cfg.num_cams * cfg.fH * cfg.fW
    V = cfg.vZ * cfg.vY * cfg.vX
    C, K, D = cfg.mid_channels, cfg.num_classes, cfg.D
    YX, oZ = cfg.oY * cfg.oX, cfg.oZ
    CO = C + (K if cfg.cat_seg else 0)
    cam = 1 + K + 3
    zf = min(1.0, (oZ + 1) / cfg.vZ)                 # fraction of volume planes the det grid touches
    return {
        "lift_fwd": B * (4 * P * (D + C) + 4 * C * V),
        "lift_bwd_fill": B * (4 * C * V + 8 * V),                 # grad_out, hit words in (round 4: no projection, no depth taps)
        "lift_bwd_gather": B * (2 * 4 * P * (D + C)),             # depth, feat in; grad_depth / grad_feat out
        "pack_volume": B * (4 * cam * V),                         # the three volumes in
        "render_cam_fwd": B * (4 * cam * V + 4 * P * (K + 4)),
        "render_cam_term": B * (4 * V + 4 * P),                   # density volume in, one int per ray out
        "render_bev_fwd": B * (4 * V * zf + 4 * YX * (oZ + 1)),
        # (since round 3 one kernel: density, weights and all channels -- the density planes and voxel_density /
        # bev_height are its traffic too)
        "render_bev_fwd_channels": B * (4 * (1 + K + 3 + C) * V * zf + 4 * YX * (K + 3) + 4 * oZ * YX * CO + 4 * YX * (oZ + 1)),
        # round 6: camera tiles + BEV column blocks in one launch (no-grad forwards with early termination): both bodies' bytes
        "render_fwd_merged": B * (4 * cam * V + 4 * P * (K + 4)
                                  + 4 * (1 + K + 3 + C) * V * zf + 4 * YX * (K + 3) + 4 * oZ * YX * CO + 4 * YX * (oZ + 1)),
        "render_cam_bwd_ray": B * (4 * cam * V + 4 * P * (K + 4)),  # volumes + upstream gradients in
        # (render_cam_bwd_gather has no entry: in the default schedule it ADDS the camera branch's contributions onto
        # the gradient volumes the BEV gather has written -- and is credited with -- and touches them only where rays
        # left records: 28 MB by the counters at cfg-B, none of it compulsory traffic of its own.  Round 4 credited it
        # 2 * 4 * cam * V = 112.6 MB, which overstated its fraction.)
        "render_bev_bwd_q": B * (4 * (K + 3) * V * zf + 4 * YX * (K + 3)),
        "render_bev_bwd_scan": B * (4 * V * zf + 4 * YX * (oZ + 2)),
        # the four volume gradients out (every plane: it overwrites), upstream voxel_output + Wb/DS0 in
        "render_bev_bwd_gather": B * (4 * (cam + C) * V + 4 * oZ * YX * (2 + CO) + 4 * YX * (K + 3)),
    }


STAGES = {   # SURVEY.md section 8(d) stage names -> kernels of this build
    "lift_fwd": ["feat_to_channel_last", "lift_fwd"],
    "render_fwd": ["pack_volume", "render_cam_term", "render_cam_fwd", "render_bev_fwd", "render_bev_fwd_channels", "render_fwd_merged"],
    # (round 4: the count pass lives in lift_fwd; "lift_bwd_count" only appears when a backward runs without it)
    "lift_bwd": ["lift_bwd_count", "lift_bwd_fill", "lift_bwd_gather", "lift_bwd_v1", "feat_to_channel_first"],
    "render_bwd": ["render_cam_bwd_ray", "render_cam_bwd_rank", "render_cam_bwd_fill", "render_cam_bwd_gather",
                   "render_cam_bwd_heavy", "render_cam_bwd_v1", "render_bev_bwd_q", "render_bev_bwd_scan",
                   "render_bev_bwd_gather", "unpack_grad", "memset"],
}


def measured_traffic(cfg_name, kernel, batch, run_kernels):
    """HBM bytes per launch from a committed rocprofv3 PMC run (profiles/traffic_*.json, made on
    the GPU box by tools/collect_profiles.sh + tools/summarize_profiles.py), or None when no run matches.
    The files are tried newest first by the round tag in their NAME (file times are checkout times after a
    clone), and one is used only if its kernel list is this run's (the profiler slots of the eager
    one-stream step): a profile made before a kernel was added, removed or renamed is stale and is
    refused rather than quoted."""
    import glob
    import re

    def tag(path):           # traffic_r03.json, traffic_r02d.json, traffic_r04_noert.json -> (round, suffix): newest first
        m = re.search(r"traffic_r(\d+)([a-z]*)", os.path.basename(path))
        return (int(m.group(1)), m.group(2)) if m else (-1, "")
    files = sorted((f for f in glob.glob(os.path.join(ROOT, "profiles", "traffic_*.json")) if "noert" not in f),
                   key=tag, reverse=True)
    for f in files:
        try:
            with open(f) as fh:
                t = json.load(fh)
        except (OSError, ValueError):
            continue
        ks = set(t.get("kernels", {}))
        # (small helper kernels share the "aux" slot of the in-library timer; the heavy-cell sums -- before round 6's
        # cell splat: the heavy-voxel drain -- are not matched, so that earlier rounds' files stay comparable)
        ignore = {"aux", "memset", "scan", "cell_scan", "cam_heavy_list", "bev_axis_table", "render_cam_bwd_heavy"}
        if (t.get("cfg") == cfg_name and t.get("batch") == batch and kernel in ks
                and (set(run_kernels) - ignore) == (ks - ignore)):
            return t["kernels"][kernel]["hbm_bytes_per_launch"]
    return None


def _cpu_run(cfg, backward):
    """One oracle pass (forward, or forward + backward) of one 6-camera sample on the host."""
    from oracle import aten_oracle as O
    from vampire_amd.geometry import PathGeometry
    from vampire_amd import synthetic
    geo = PathGeometry(cfg)
    s2e, K, ida = synthetic.camera_rig(cfg, 1)
    bda = synthetic.bda_matrix(1)
    depth, feat = synthetic.lift_inputs(cfg, 1)
    vols = list(synthetic.render_inputs(cfg, 1))
    beta = torch.tensor(0.1, requires_grad=backward)
    if backward:
        depth.requires_grad_(True); feat.requires_grad_(True)
        for v in vols:
            v.requires_grad_(True)
    t0 = time.perf_counter()
    with torch.set_grad_enabled(backward):
        vox, outs = O.lift_render_forward(cfg, geo, depth, feat, vols, (s2e, K, ida, bda), beta)
        if backward:
            loss = vox.sum() * 1e-3 + sum(o.sum() for o in outs) * 1e-3
            loss.backward()
    return time.perf_counter() - t0


def cpu_baseline(cfg, cfg_name):
    """The oracle (torch-CPU restatement of the reference's op sequence) on the host cores, bounded
    sample: one 6-camera sample of the bench workload per run, median of 3 runs for
    forward + backward (the metric) and for the forward alone; one run each of the reference's
    default configuration (cfg-A) beside it."""
    from vampire_amd.config import CFG_A
    import statistics
    fb = [_cpu_run(cfg, True) for _ in range(3)]
    fw = [_cpu_run(cfg, False) for _ in range(3)]
    med_fb, med_fw = statistics.median(fb), statistics.median(fw)
    out = {"value": 1.0 / med_fb, "unit": "samples/s", "cores": torch.get_num_threads(),
           "kind": "port",
           "sample": f"1 sample of the bench workload (cfg-{cfg_name}) per run, oracle/aten_oracle.py; fwd+bwd median of 3 = "
                     f"{med_fb:.2f} s, fwd alone median of 3 = {med_fw:.2f} s; {torch.get_num_threads()} torch threads on "
                     f"{os.cpu_count()} logical CPUs",
           "fwd_samples_per_s": 1.0 / med_fw, "runs_s": {"fwd_bwd": [round(t, 3) for t in fb], "fwd": [round(t, 3) for t in fw]}}
    if cfg_name != "A":
        out["cfg_A"] = {"fwd_bwd_s": round(_cpu_run(CFG_A, True), 3), "fwd_s": round(_cpu_run(CFG_A, False), 3)}
    return out


def forward_pair_us(model, batch, iters=100, warm=20):
    """the fused forward timed as ONE unit -- lift kernels then render kernels on
    pre-generated volumes, one HIP-event pair around the pair per iteration (recorded on the stream
    the kernels are launched on; the side stream's work is joined before the second event), >= 20
    warm-up + >= 100 timed iterations, median."""
    with torch.no_grad():
        def fwd():
            model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
        for _ in range(warm):
            fwd()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); fwd(); b.record()
            b.synchronize()
            ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    return ts[len(ts) // 2], ts[len(ts) // 10], ts[(9 * len(ts)) // 10]


def forward_pair_graph_us(model, batch, iters=100, warm=20):
    """The same forward pair replayed from a HIP graph, where the launches cost no host time (the lift is NOT
    overlapped with the renderer: in the model the 3-D UNet sits between them).  Two captures -- the renderer's
    camera and BEV branch, which share only their inputs, on ONE stream and side by side on two
    (HotPath.impl["fwd_overlap"]) -- and the faster one is reported: since round 5 that is the one-stream graph
    (a replayed fork costs the camera kernel a 6 - 13 us later start and, beside the BEV kernel, 15 us of its own
    run time: 136 against 124 us).  One HIP-event pair around each replay, median.
    Returns (median, p10, p90, "one stream" | "two streams"), or None when no capture reproduces the eager outputs."""
    best = None
    for overlap in (False, True):
        r = _forward_pair_graph_us(model, batch, overlap, iters, warm)
        if r is not None and (best is None or r[0] < best[0]):
            best = r + ("two streams" if overlap else "one stream",)
    return best


def _forward_pair_graph_us(model, batch, overlap, iters, warm):
    hp = model.hp
    keep = hp.impl["fwd_overlap"]
    try:
        with torch.no_grad():
            def fwd():
                return model(batch.depth, batch.feat, batch.vols, batch.lift_mats, batch.render_mats)
            vox0, outs0 = fwd()
            ref = [vox0.clone()] + [o.clone() for o in outs0]
            hp.impl["fwd_overlap"] = overlap
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                for _ in range(3):
                    fwd()
            cur.wait_stream(side)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                vox, outs = fwd()
            torch.cuda.synchronize()
            g.replay()
            torch.cuda.synchronize()
            for got, want in zip([vox] + list(outs), ref):
                if not (torch.equal(got, want) or float((got - want).abs().max()) <= 1e-6 * float(want.abs().max())):
                    raise RuntimeError("graph replay does not reproduce the eager forward")
            for _ in range(warm):
                g.replay()
            torch.cuda.synchronize()
            ts = []
            for _ in range(iters):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(); g.replay(); b.record()
                b.synchronize()
                ts.append(a.elapsed_time(b) * 1e3)
            # throughput view of the same graph: `iters` replays behind each other between ONE event pair (the protocol
            # of the headline step: a replay's launch latency, ~8 us, hides behind the previous replay's kernels)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                g.replay()
            b.record()
            b.synchronize()
            back_to_back = a.elapsed_time(b) * 1e3 / iters
        ts.sort()
        return ts[len(ts) // 2], ts[len(ts) // 10], ts[(9 * len(ts)) // 10], back_to_back
    except Exception as e:                          # noqa: BLE001 -- report the eager number only
        print(f"[bench] forward graph unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        return None
    finally:
        hp.impl["fwd_overlap"] = keep


def extra_config(cfg_name, batch, dtype, steps=10, warm=3, ert=True, density_mode=None):
    """A secondary configuration, measured the same way as the headline (fwd+bwd step time with a
    barrier-free single-rank loop) plus its forward pair: driver-observed rather than README prose.
    `ert=False` switches the camera branch's early ray termination off (every sample marched);
    `density_mode="naive"` is the sigmoid density, under which no ray saturates on this data (the
    worst case for that optimisation)."""
    import dataclasses
    from vampire_amd.config import PRESETS
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step
    cfg = PRESETS[cfg_name]
    if density_mode is not None:
        cfg = dataclasses.replace(cfg, density_mode=density_mode)
    dev = torch.device("cuda", torch.cuda.current_device())
    model = LiftRenderStep(cfg, dev)
    model.hp.impl["ert"] = ert
    data = SyntheticBatch(cfg, batch, dev, seed=1, dtype=dtype)
    for _ in range(warm):
        model.zero_grad(set_to_none=True); train_step(model, data)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.zero_grad(set_to_none=True); train_step(model, data)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    # the same step replayed from a HIP graph, as the headline is timed (None when capture is unavailable)
    graph_ms = None
    if os.environ.get("VAMP_BENCH_GRAPH", "1") == "1":
        try:
            g = capture_step(model, data, train_step)
            for _ in range(3):
                g.replay()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                g.replay()
            torch.cuda.synchronize()
            graph_ms = (time.perf_counter() - t0) / steps * 1e3
            del g
        except Exception as e:                      # noqa: BLE001
            print(f"[bench] extra config graph capture unavailable ({type(e).__name__}: {e})", file=sys.stderr)
    fwd_us, _, _ = forward_pair_us(model, data, iters=30, warm=5)
    ab = cfg.algorithmic_bytes(4 if dtype == torch.float32 else 2)
    return {"workload": f"cfg-{cfg_name}, {batch} sample(s)/GPU/step, {'f32' if dtype == torch.float32 else 'bf16'} inputs"
                        + ("" if ert else ", early ray termination OFF")
                        + ("" if density_mode is None else f", density_mode={density_mode}"),
            "samples_per_s": batch / (ms * 1e-3), "ms_per_step": ms, "launch": "eager (samples_per_s, ms_per_step); graph_* = HIP-graph replay like the headline",
            "graph_ms_per_step": graph_ms, "graph_samples_per_s": None if graph_ms is None else batch / (graph_ms * 1e-3),
            "fwd_us": fwd_us,
            "fwd_frac_of_hbm_peak": ab["fwd"] * batch / (fwd_us * 1e-6) / 1e9 / HBM_PEAK_GBS}


def layered_measure(cfg, dev, batch_per_gpu, rank, world, steps=6, warm=3, find=False):
    """the operators between the reference backbone's own layers (step.LayeredStep: 777 111
    parameters, a 3.1 MB gradient bucket) under the bucketed GradSync (VAMP_GRAD_SYNC=ddp:
    DistributedDataParallel), so that a multi-GPU run exercises a real all-reduce overlapped with the
    backward.  Reported beside the headline (whose only parameter is the density beta); the step is
    dominated by the 3-D UNet's convolutions, which are not part of the hot path."""
    from vampire_amd import dist as vdist
    from vampire_amd.step import LayeredStep, LayeredBatch, layered_step
    torch.backends.cudnn.benchmark = find      # MIOpen find mode for the UNet's remaining layers, or immediate mode
    model = LayeredStep(cfg, dev)
    wrapped = vdist.wrap_ddp(model, dev)
    data = LayeredBatch(cfg, batch_per_gpu, dev, seed=vdist.shard_seed(0, rank))

    def step():
        model.zero_grad(set_to_none=True)
        layered_step(wrapped, data)

    tw = time.perf_counter()
    for _ in range(warm):
        step()
    vdist.barrier(); torch.cuda.synchronize()
    warm_s = time.perf_counter() - tw            # (a real MIOpen find shows here: minutes instead of a second)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    vdist.barrier(); torch.cuda.synchronize()
    el = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    nparam = sum(p.numel() for p in model.parameters())
    sync = ("none (1 rank)" if wrapped is model else
            ("GradSync, %d buckets of <= 1 MiB, all-reduce launched as each bucket completes" % len(wrapped._buckets)
             if isinstance(wrapped, vdist.GradSync) else "DistributedDataParallel"))
    return {"ms_per_step": el / steps * 1e3, "samples_per_s": batch_per_gpu * world * steps / el,
            "parameters": nparam, "gradient_bytes": 4 * nparam, "grad_sync": sync, "steps": steps,
            "warmup_s": round(warm_s, 2),
            "what": "mapping_along_depth, channel_lower, depth softmax, lift, Unet3D, heads, render, occupancy queries, "
                    "voxel_output: forward + backward from synthetic neck features"}


def multitask_measure(dev, batch_per_gpu, rank, world, steps=6, warm=3, find=False):
    """BASELINE.json configs[4]: the full multi-task model -- ResNet-50 + SECOND FPN image encoder, the
    backbone with the HIP lift / render / query / gate operators, the CenterPoint-style BEV head -- on the
    reference's own configuration (cfg-A: 256 x 704 images, 256 x 256 x 20 seg grid, base_exp.py:40-252),
    a collate_fn-shaped synthetic batch, the nine losses, AdamW step, under bf16 autocast; data-parallel
    over the ranks (gradient all-reduce as in layered_measure).  Encoders / head / losses are this
    build's own stand-ins for the absent mmdet / mmdet3d / torchmetrics (vampire_amd/multitask.py)."""
    from vampire_amd import dist as vdist
    from vampire_amd import multitask as M
    from vampire_amd.config import CFG_A
    torch.backends.cudnn.benchmark = find
    torch.manual_seed(0)
    bb, hd = M.reference_confs(CFG_A)
    model = M.VAMPIRE2(bb, hd).to(dev)
    with torch.no_grad():
        model.backbone.density_conv.bias.fill_(CFG_A.sdf_bias)     # informative densities (the init value saturates every ray)
    wrapped = vdist.wrap_ddp(model, dev)
    loss_fn = M.MultiTaskLoss(model, downsample_factor=4, upsample_factor=4, sdf_bias=CFG_A.sdf_bias)
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4)
    data = M.synthetic_batch(CFG_A, batch_per_gpu, seed=vdist.shard_seed(0, rank), device=dev, num_points=30000, num_boxes=30)
    loss = None
    for _ in range(warm):
        loss = M.multitask_step(wrapped, loss_fn, data, optimizer=opt)
    vdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = M.multitask_step(wrapped, loss_fn, data, optimizer=opt)
    vdist.barrier(); torch.cuda.synchronize()
    el = vdist.max_over_ranks(time.perf_counter() - t0, dev)
    nparam = sum(p.numel() for p in model.parameters())
    return {"ms_per_step": el / steps * 1e3, "samples_per_s": batch_per_gpu * world * steps / el, "steps": steps,
            "parameters": nparam, "loss": float(loss.detach()), "amp": "bf16", "batch_per_gpu": batch_per_gpu,
            "loss_terms": {k: float(v) for k, v in loss_fn.last.items()},
            "what": "configs[4]: R50 + SECONDFPN + BaseVAMPIRE2 (HIP operators) + BEVDepthHead stand-ins, 9 losses, "
                    "AdamW; cfg-A, synthetic collate_fn-shaped batch; forward + backward + optimizer step"}


def capture_step(model, batch, train_step):
    """One training step (without the DDP wrapper) captured into a HIP graph; the replay must
    reproduce the eager gradients."""
    def raw_step():
        model.zero_grad(set_to_none=True)
        train_step(model, batch)

    cur = torch.cuda.current_stream()
    side = torch.cuda.Stream()
    side.wait_stream(cur)
    with torch.cuda.stream(side):
        for _ in range(3):
            raw_step()
    cur.wait_stream(side)
    torch.cuda.synchronize()
    tensors = lambda: [t for t in [batch.depth.grad, batch.feat.grad] + [v.grad for v in batch.vols] + [model.beta.grad]
                       if t is not None]                    # (no beta gradient under density_mode="naive")
    ref = [t.detach().clone() for t in tensors()]
    g = torch.cuda.CUDAGraph()
    # thread_local: calls of other threads (the RCCL watchdog polls its events) must not fail the capture
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        raw_step()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    for got, want in zip(tensors(), ref):
        # fp32 sums with LDS float atomics differ in the last bits from run to run; a gradient handed back in
        # bf16 (bf16 inputs) can flip one rounding step on such a difference
        tol = 1e-5 if got.dtype == torch.float32 else 2.0 ** -7
        if not (torch.equal(got, want) or float((got.float() - want.float()).abs().max()) <= tol * float(want.float().abs().max())):
            raise RuntimeError("graph replay does not reproduce the eager gradients")
    return g


class OverlappedBetaSync:
    """The DDP all-reduce of the path's one parameter gradient around a REPLAYED step.  The captured step
    holds no collective; after replay k the gradient is copied to a side buffer (4 bytes) and its
    all-reduce is launched asynchronously on RCCL's stream, where it runs under replay k + 1 (whose
    backward rewrites beta.grad); it is joined -- a stream wait, the host does not block -- before the
    next copy, and once more at the end of the timed region.  The headline therefore times replays with
    an overlapped collective, not replay + a serialized 4-byte RCCL latency."""

    def __init__(self, model, world):
        self.model, self.world, self.work, self.buf = model, world, None, None

    def after_replay(self):
        if self.world <= 1 or self.model.beta.grad is None:
            return
        if self.work is not None:
            self.work.wait()
        if self.buf is None:
            self.buf = torch.empty_like(self.model.beta.grad)
        self.buf.copy_(self.model.beta.grad)
        self.work = dist.all_reduce(self.buf, async_op=True)      # sum (gloo has no AVG); the mean at the join

    def join(self):
        if self.work is not None:
            self.work.wait()
            self.model.beta.grad.copy_(self.buf / self.world)
            self.work = None


def replay_rate(model, batch, train_step, steps, world, dev, vdist, sync_wrapper=None):
    """`steps` steps of `batch` under the headline's protocol -- the step replayed from a HIP graph (all ranks,
    or none), overlapped beta all-reduce, barrier + synchronize on both sides, max over ranks -- for a
    secondary batch size.  `sync_wrapper`: the GradSync around `model` on a multi-rank run -- its hook must stay
    silent while the step is captured (the capture must not hold the collective: found with two ranks in round 6,
    "capturing stream has unjoined work").  Returns (seconds, launch mode)."""
    graph = None
    if os.environ.get("VAMP_BENCH_GRAPH", "1") == "1":
        hook = sync_wrapper is not None and hasattr(sync_wrapper, "enabled")
        try:
            if hook:
                sync_wrapper.enabled = False
            graph = capture_step(model, batch, train_step)
        except Exception as e:                          # noqa: BLE001
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); timing eager steps", file=sys.stderr)
            model.hp._side = None                       # (a stream a failed capture forked stays in capture mode: a fresh one)
        finally:
            if hook:
                sync_wrapper.enabled = True
    if world > 1:
        ok = torch.tensor([1.0 if graph is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            graph = None
    # (eager fallback: the wrapper's own hook + join do the all-reduce; replays use the overlapped one)
    eager_model = sync_wrapper if sync_wrapper is not None else model
    sync = OverlappedBetaSync(model, world if graph is not None or sync_wrapper is None else 1)

    def step():
        if graph is not None:
            graph.replay()
        else:
            model.zero_grad(set_to_none=True)
            train_step(eager_model, batch)
        sync.after_replay()

    for _ in range(3):
        step()
    sync.join()
    vdist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync.join()
    vdist.barrier(); torch.cuda.synchronize()
    return vdist.max_over_ranks(time.perf_counter() - t0, dev), ("hip_graph" if graph is not None else "eager")


_T0 = time.perf_counter()


def _mark(what):
    """progress on stderr: where the wall time of a bench run goes"""
    print(f"[bench] +{time.perf_counter() - _T0:6.1f} s  {what}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cfg", default="B", help="A | B | D (vampire_amd.config.PRESETS)")
    ap.add_argument("--batch", type=int, default=1, help="samples per GPU per step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary configurations (extra_configs)")
    ap.add_argument("--no-matrix", action="store_true",
                    help="skip the {eager, graph} x {early termination on, off} step matrix (profile runs: only default-path kernels)")
    ap.add_argument("--multitask", action="store_true", help="(accepted for compatibility: the multi-task step is always timed unless --no-extra)")
    ap.add_argument("--layers", action="store_true", help="(accepted for compatibility: the layered step is always timed unless --no-extra)")
    ap.add_argument("--miopen-find", action="store_true",
                    help="let MIOpen tune the convolutions of the layered / multi-task legs (6.5 minutes on a fresh box)")
    a = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world == 1:
        # convenience: re-launch ourselves under torch.distributed.run as a child process
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
               f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", "29533",
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    from vampire_amd.config import PRESETS
    from vampire_amd import _capi
    from vampire_amd.step import LiftRenderStep, SyntheticBatch, train_step

    cfg = PRESETS[a.cfg]
    # one process per GPU; (the modulo only matters for the 2-ranks-on-1-GPU gloo self-test)
    dev_index = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    from vampire_amd import dist as vdist
    # "nccl" == RCCL over xGMI on ROCm; VAMP_DIST_BACKEND=gloo exists for self-tests only
    vdist.init(os.environ.get("VAMP_DIST_BACKEND", "nccl"), dev)
    dtype = torch.float32 if a.dtype == "f32" else torch.bfloat16

    model = LiftRenderStep(cfg, dev)
    step_model = vdist.wrap_ddp(model, dev)
    batch = SyntheticBatch(cfg, a.batch, dev, seed=vdist.shard_seed(0, rank), dtype=dtype)

    def one_step():
        model.zero_grad(set_to_none=True)
        train_step(step_model, batch)

    def fence():
        vdist.barrier()
        torch.cuda.synchronize()

    # Warm-up: the first steps untimed, the last ones with HIP events around EVERY kernel (the
    # per-kernel / per-stage table).  Timed region: events only around the dominant kernel found
    # in the warm-up -- one event pair per step instead of ~40, so the timer does not distort the
    # step time it is measuring.
    n_prof = min(a.warmup, 3)
    for _ in range(a.warmup - n_prof):
        one_step()
    fence()
    # (per-kernel timings are taken with both render branches on ONE stream: side by side on two
    # streams the kernels stretch each other and an event pair no longer times one kernel)
    overlap = model.hp.impl["overlap"]
    model.hp.impl["overlap"] = False
    one_step()
    fence()
    _capi.profile_select(None)
    _capi.profile_enable(True)
    for _ in range(n_prof):
        one_step()
    fence()
    _capi.profile_enable(False)
    model.hp.impl["overlap"] = overlap
    warm = {k: (n, ms, n_prof) for k, (n, ms) in _capi.profile_read().items()}
    alg = kernel_algorithmic_bytes(cfg, a.batch)
    dom = None
    if warm:
        dom = max((k for k in warm if k in alg), key=lambda k: warm[k][1])
        _capi.profile_select(dom)
    fence()
    # Timed region.  Default: the step replayed from a HIP graph (shapes are static; the capture holds
    # both streams of the step, the zero fills and the workspaces -- tests/test_hip_parity.py::
    # test_step_is_graph_capturable); the DDP all-reduce of the one parameter gradient stays outside
    # the graph and is issued after every replay.  VAMP_BENCH_GRAPH=0, or a capture that fails or
    # does not reproduce the eager gradients, times eager steps instead.  Either way the dominant
    # kernel's launch duration comes from HIP events around it in K eager steps of this same run.
    graph, launch_mode = None, "eager"
    hook_sync = isinstance(step_model, vdist.GradSync)
    if os.environ.get("VAMP_BENCH_GRAPH", "1") == "1" and (step_model is model or hook_sync):
        try:
            if hook_sync:
                step_model.enabled = False          # the capture must not hold the collective
            graph = capture_step(model, batch, train_step)
            launch_mode = "hip_graph"
        except Exception as e:                      # noqa: BLE001 -- any failure: measure eagerly
            print(f"[bench] graph capture unavailable ({type(e).__name__}: {e}); timing eager steps", file=sys.stderr)
            graph = None
        if hook_sync:
            step_model.enabled = True
    if world > 1:
        # all ranks replay, or all launch eagerly: one rank falling back alone would pair its averaging
        # all-reduce with the others' sums
        ok = torch.tensor([1.0 if graph is not None else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if float(ok.item()) == 0.0:
            graph, launch_mode = None, "eager"

    _mark("warm-up + capture done")
    beta_sync = OverlappedBetaSync(model, world)

    def graph_step():
        graph.replay()
        beta_sync.after_replay()                    # asynchronous: runs under the next replay

    timed_step = graph_step if graph is not None else one_step
    for _ in range(3):
        timed_step()
    beta_sync.join()
    fence()
    if graph is None:
        _capi.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        timed_step()
    beta_sync.join()
    fence()
    elapsed = time.perf_counter() - t0
    _capi.profile_enable(False)
    elapsed = vdist.max_over_ranks(elapsed, dev)
    if graph is not None:
        # the dominant kernel's launches, timed with HIP events in K eager one-stream steps
        model.hp.impl["overlap"] = False
        one_step()
        fence()
        _capi.profile_enable(True)
        for _ in range(a.steps):
            one_step()
        fence()
        _capi.profile_enable(False)
        model.hp.impl["overlap"] = overlap
    _capi.profile_select(None)
    _mark("timed region + kernel profile done")
    fwd_med, fwd_p10, fwd_p90 = forward_pair_us(model, batch) if rank == 0 else (0.0, 0.0, 0.0)
    fwd_graph = forward_pair_graph_us(model, batch) if (rank == 0 and os.environ.get("VAMP_BENCH_GRAPH", "1") == "1") else None
    # the same pair with early ray termination OFF: the data-independent forward (every inside sample marched)
    fwd_off, fwd_off_graph = None, None
    if rank == 0 and not a.no_matrix:
        ert_default = model.hp.impl["ert"]
        model.hp.impl["ert"] = False
        fwd_off = forward_pair_us(model, batch, iters=50, warm=10)
        if os.environ.get("VAMP_BENCH_GRAPH", "1") == "1":
            fwd_off_graph = forward_pair_graph_us(model, batch, iters=50, warm=10)
        model.hp.impl["ert"] = ert_default

    # {HIP graph, eager launches} x {early ray termination on, off}: the headline is the best case on
    # both axes (the termination gain is data-dependent), so the line carries all four (N = 1 only:
    # no collective inside)
    _mark("forward pairs done")
    step_matrix, ert_stats = None, None
    if world == 1 and not a.no_matrix:
        def time_loop(fn, n):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t) / n * 1e3
        step_matrix = {}
        ert_default = model.hp.impl["ert"]
        for ert in (True, False):
            model.hp.impl["ert"] = ert
            tag = "ert_on" if ert else "ert_off"
            step_matrix["eager_" + tag + "_ms"] = round(time_loop(one_step, a.steps), 4)
            if ert == ert_default and graph is not None:
                step_matrix["graph_" + tag + "_ms"] = round(elapsed / a.steps * 1e3, 4)     # the headline itself
            elif os.environ.get("VAMP_BENCH_GRAPH", "1") == "1":
                try:
                    g2 = capture_step(model, batch, train_step)
                    step_matrix["graph_" + tag + "_ms"] = round(time_loop(g2.replay, a.steps), 4)
                    del g2
                except Exception as e:              # noqa: BLE001
                    step_matrix["graph_" + tag + "_ms"] = None
                    print(f"[bench] graph capture ({tag}) unavailable ({type(e).__name__}: {e})", file=sys.stderr)
        model.hp.impl["ert"] = ert_default
        inside, kept = model.hp.ert_statistics(batch.vols[0].detach(), model.beta if cfg.density_mode == "sdf" else None,
                                               batch.render_mats)
        ert_stats = {"inside_samples": inside, "kept_samples": kept,
                     "terminated_fraction": round(1.0 - kept / max(1, inside), 4)}

    # the same workload at 8 samples per GPU per step under the headline's protocol, so that the
    # driver's N = 1, 2, 4, 8 runs also carry a weak-scaling curve that is not bounded by the latency of a
    # collective against a half-millisecond step (configs[2] itself is bs = 1 per GPU: the headline)
    _mark("step matrix done")
    bs8 = None
    if not a.no_extra and a.batch != 8:
        batch8 = SyntheticBatch(cfg, 8, dev, seed=vdist.shard_seed(1, rank), dtype=dtype)
        el8, mode8 = replay_rate(model, batch8, train_step, a.steps, world, dev, vdist,
                                 sync_wrapper=step_model if step_model is not model else None)
        bs8 = {"value": 8 * world * a.steps / el8, "unit": "samples/s", "ms_per_step": el8 / a.steps * 1e3,
               "per_gpu_batch": 8, "global_batch": 8 * world, "n_gpus": world, "steps": a.steps, "launch": mode8,
               "scaling": "weak"}
        del batch8
    # SURVEY 8(e) / BASELINE configs[4]: the step with a real gradient bucket and the full multi-task step,
    # always in the line (a few seconds; --no-extra skips them for profile runs)
    _mark("bs8 done")
    # (with the shipped find-db the "find" of these two legs is a look-up: tuned convolutions at no cost)
    db_ok = _miopen_db_matches()
    tuned = a.miopen_find or db_ok
    layered = layered_measure(cfg, dev, a.batch, rank, world, find=tuned) if not a.no_extra else None
    _mark("layered step done")
    multitask = multitask_measure(dev, a.batch, rank, world, find=tuned) if not a.no_extra else None
    _mark("multi-task step done")

    prof = dict(warm)
    # the dominant kernel: measured over the timed region
    prof.update({k: (n, ms, a.steps) for k, (n, ms) in _capi.profile_read().items()})
    if rank == 0:
        kern = {k: {"launches": n, "avg_us": ms / n * 1e3, "us_per_step": ms / st * 1e3,
                    "launches_per_step": n / st}
                for k, (n, ms, st) in prof.items()}
        if dom is None:
            dom = max((k for k in kern if k in alg), key=lambda k: kern[k]["us_per_step"])
        # `alg` holds the bytes of all launches of a kernel name in one step (the BEV gather runs
        # once per channel kind): per launch = alg / launches per step, over the average launch
        dom_alg_launch = alg[dom] / kern[dom]["launches_per_step"]
        dom_gbs = dom_alg_launch / (kern[dom]["avg_us"] * 1e-6) / 1e9
        # stage view with SURVEY.md section 8(d)'s algorithmic bytes: kernel time per step, summed
        # (with the two render branches on two streams the stage's wall time is below this sum)
        sb = cfg.algorithmic_bytes(4 if a.dtype == "f32" else 2)
        stages = {}
        for st, names in STAGES.items():
            us = sum(kern[k]["us_per_step"] for k in names if k in kern)
            stages[st] = {"us": round(us, 1), "algorithmic_bytes": sb[st] * a.batch,
                          "frac_of_hbm_peak": round(sb[st] * a.batch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)
                          if us > 0 else None}
        fwd_kernel_sum_us = stages["lift_fwd"]["us"] + stages["render_fwd"]["us"]
        # the pair timed as one unit: replayed from a HIP graph (camera and BEV branch side by side) when
        # the capture works, else the eager one-stream launches; both are reported
        fwd_us = fwd_graph[0] if fwd_graph else fwd_med
        fwd_bytes = sb["fwd"] * a.batch
        fwd_gbs = fwd_bytes / (fwd_us * 1e-6) / 1e9
        line = {
            "metric": "6-cam samples/sec (lift+render fwd+bwd)",
            "value": a.batch * world * a.steps / elapsed,
            "unit": "samples/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": f"cfg-{a.cfg}: 6x{cfg.final_dim[0]}x{cfg.final_dim[1]} images, "
                                   f"D={cfg.D} C={cfg.mid_channels} K={cfg.num_classes}, voxel grid "
                                   f"{cfg.vX}x{cfg.vY}x{cfg.vZ}, det grid {cfg.oX}x{cfg.oY}x{cfg.oZ}, "
                                   f"{a.batch} sample(s)/GPU/step, lift+render fwd+bwd, density_mode={cfg.density_mode} "
                                   f"(the reference's default), camera-branch early ray termination "
                                   f"{'on' if model.hp.impl['ert'] else 'off'} (T < 1.5e-8)"
                                   + (f": on this synthetic density (0.5 randn - 1) {100 * ert_stats['terminated_fraction']:.0f} % of the "
                                      f"inside samples lie behind a saturated ray and are skipped -- data-dependent, see step_matrix "
                                      f"for the same step with termination off" if ert_stats else ""),
                       "per_gpu_batch": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}",
                       "launch": ("the step replayed from a HIP graph (DDP all-reduce issued after each replay); "
                                  "kernel timings from HIP events in eager one-stream steps of the same run"
                                  if launch_mode == "hip_graph" else "eager launches")},
            "rccl_ranks": (dist.get_world_size() if dist.is_initialized() else 1),
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": dom_gbs, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": dom_gbs / HBM_PEAK_GBS,
                         "traffic": measured_traffic(a.cfg, dom, a.batch, list(kern)),
                         "algorithmic_bytes_per_launch": dom_alg_launch,
                         "launches_per_step": kern[dom]["launches_per_step"],
                         "avg_launch_us": kern[dom]["avg_us"],
                         # the same figure for every kernel that has compulsory bytes of its own (the kernel the
                         # round-2 review named, lift_bwd_fill, is in here)
                         "per_kernel": {k: {"avg_launch_us": round(kern[k]["avg_us"], 2),
                                            "frac": round(alg[k] / kern[k]["launches_per_step"] / (kern[k]["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                            "traffic": measured_traffic(a.cfg, k, a.batch, list(kern))}
                                        for k in sorted(kern) if k in alg and kern[k]["avg_us"] > 0}},
            "fwd_roofline": {"bound": "hbm", "achieved": fwd_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": fwd_gbs / HBM_PEAK_GBS, "fused_fwd_us": fwd_us,
                             "p10_us": fwd_graph[1] if fwd_graph else fwd_p10, "p90_us": fwd_graph[2] if fwd_graph else fwd_p90,
                             "iters": 100, "warmup": 20,
                             "timing": (f"one HIP-event pair around each replay of the captured forward (lift, then the renderer's "
                                        f"camera and BEV branch on {fwd_graph[4]}: the faster of the two captures), median"
                                        if fwd_graph else "one HIP-event pair around lift + render per iteration, median"),
                             "back_to_back_us": fwd_graph[3] if fwd_graph else None,
                             "back_to_back_frac": (fwd_bytes / (fwd_graph[3] * 1e-6) / 1e9 / HBM_PEAK_GBS) if fwd_graph else None,
                             "back_to_back_note": "the same graph replayed `iters` times between one event pair (the headline step's protocol); `frac` above stays on the per-replay median",
                             "eager_one_stream_us": fwd_med, "eager_p10_us": fwd_p10, "eager_p90_us": fwd_p90,
                             "eager_frac": fwd_bytes / (fwd_med * 1e-6) / 1e9 / HBM_PEAK_GBS,
                             "kernel_sum_us": fwd_kernel_sum_us, "algorithmic_bytes": fwd_bytes},
            "fwd_roofline_ert_off": (None if fwd_off is None else {
                "bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes": fwd_bytes,
                "fused_fwd_us": (fwd_off_graph or fwd_off)[0],
                "achieved": fwd_bytes / ((fwd_off_graph or fwd_off)[0] * 1e-6) / 1e9,
                "frac": fwd_bytes / ((fwd_off_graph or fwd_off)[0] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "eager_one_stream_us": fwd_off[0], "eager_frac": fwd_bytes / (fwd_off[0] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                "iters": 50, "warmup": 10,
                "what": "the same forward pair with early ray termination OFF: every inside sample is marched (data-independent)"}),
            "stages": stages,
            "step_matrix": step_matrix,
            "step_matrix_note": "graph_* = device-bound (replayed launches); eager_* depend on the HOST's launch rate "
                                "(~35 launches and two stream joins per step) and vary between boxes",
            "weak_scaling_bs8": bs8,
            "miopen": ("find (tuned convolutions)" if a.miopen_find else
                       "find served from the shipped user find-db + kernel cache (profiles/miopen_db, made by tools/miopen_tune.py; "
                       "its file names match this MIOpen build and device): tuned convolutions for the layered / multi-task legs "
                       "without the 6.5 minutes of find" if db_ok else
                       "immediate mode (no find) for the layered / multi-task legs: their convolutions are untuned "
                       + ("(the shipped find-db is for another MIOpen build / device); " if MIOPEN_DB is not None else "; ")
                       + "--miopen-find tunes them"),
            "layered_step": layered,
            "multitask_step": multitask,
            "early_ray_termination": ert_stats,
            "kernels_avg_us": {k: round(v["avg_us"], 2) for k, v in sorted(kern.items())},
            "kernels_us_per_step": {k: round(v["us_per_step"], 2) for k, v in sorted(kern.items())},
        }
        if world == 1 and not a.no_extra:
            # secondary configurations of BASELINE.json / SURVEY 8(d), measured in this same run
            line["extra_configs"] = [extra_config("A", 1, torch.float32), extra_config("B", 8, torch.float32),
                                     extra_config("D", 1, torch.bfloat16),
                                     extra_config(a.cfg, a.batch, dtype, ert=False),
                                     extra_config(a.cfg, a.batch, dtype, density_mode="naive")]
        _mark("extra configs done")
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cfg, a.cfg)
            _mark("cpu baseline done")
        print(json.dumps(line), flush=True)
    vdist.shutdown()


if __name__ == "__main__":
    main()
