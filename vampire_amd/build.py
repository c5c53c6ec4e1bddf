"""Build recipe for the C-ABI HIP library (gfx950 only).

    python -m vampire_amd.build            # -> vampire_amd/_lib/libvampire_hip.so

hipcc cross-compiles without a GPU.  The library is kept in-tree (git-ignored) so
that it travels to the GPU box with the repo snapshot.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "_lib")
LIBNAME = "libvampire_hip.so"
SOURCES = ["runtime.hip", "lift.hip", "lift_bwd_cell.hip", "render_fwd.hip", "render_cam_direct.hip", "render_bwd.hip", "render_bwd_ray.hip", "render_bwd_cell.hip", "render_bev.hip", "render_bev_fused.hip", "render_fwd_merged.hip", "sample_points.hip", "glue.hip", "gate_conv.hip", "voxel_pooling.hip", "upsample.hip", "conv3d.hip", "conv3d_bf16.hip"]
# -ffp-contract=off: the projection chains must round like the reference's fp32 ops
FLAGS = ["-O3", "--offload-arch=gfx950", "-ffp-contract=off", "-fPIC", "-std=c++17",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"] + os.environ.get("VAMP_EXTRA_FLAGS", "").split()


# per-file flags.  -fno-slp-vectorize: hipcc pairs the per-channel fma chains of the gathers into
# v_pk_* instructions and spends as many v_mov on building the register pairs as it saves
FILE_FLAGS = {}


def lib_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "vampire_hip.h"))
    objs, jobs = [], []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    out = lib_path()
    if jobs or force or _stale(out, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv))
