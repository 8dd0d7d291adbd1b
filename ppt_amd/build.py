"""Build libppt_hip.so (every HIP kernel of the package, gfx950 only) in-tree with hipcc.

    python -m ppt_amd.build [--force]

hipcc cross-compiles without a GPU; the .so sits next to the sources (git-ignored, but it travels
with the gpurun snapshot).  FPS / kNN are built with -ffp-contract=off on top of their explicit
_rn intrinsics: their results must be bit-exact.
"""
import concurrent.futures
import os
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
OUT = os.path.join(CSRC, "libppt_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-slp-vectorize: no compiler-formed packed-fp32 VALU ops (v_pk_add_f32 / v_pk_fma_f32).  A dependent chain of them
# (the BatchNorm chunk sums of the GEMM epilogue: v_pk_add_f32 x8, s_nop 0 between, horizontal op_sel add) gave wrong
# lanes 16-31 in ~1e-4 of the chunks, different ones on every run, on gfx950 with ROCm 7.2 (tools/gemm_determinism.py);
# the scalar chain is bit-stable.  PPT_SLP=1 re-enables it for A/B timing.
COMMON = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-fno-gpu-rdc"] + ([] if os.environ.get("PPT_SLP") == "1" else ["-fno-slp-vectorize"])
PER_FILE = {"fps.hip": ["-ffp-contract=off"], "knn_group.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ppt_hip.h"))
    jobs, objs = [], []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(bdir, f[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([HIPCC] + COMMON + PER_FILE.get(f, []) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if jobs or force or _stale(OUT, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
