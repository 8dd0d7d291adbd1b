"""Build libppt_hip.so (every HIP kernel of the package, gfx950 only) in-tree with hipcc.

    python -m ppt_amd.build [--force]

hipcc cross-compiles without a GPU; the .so sits next to the sources (git-ignored, but it travels
with the gpurun snapshot).  FPS / kNN are built with -ffp-contract=off on top of their explicit
_rn intrinsics: their results must be bit-exact.
"""
import concurrent.futures
import hashlib
import json
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
COMMON = ["-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-Wno-inline-asm",
          "-fno-gpu-rdc"] + ([] if os.environ.get("PPT_SLP") == "1" else ["-fno-slp-vectorize"])
PER_FILE = {"fps.hip": ["-ffp-contract=off"], "knn_group.hip": ["-ffp-contract=off"]}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths, extra=()):
    """sha256 over the CONTENT of the files (and the flags): the prebuilt objects travel with the gpurun snapshot, where
    modification times mean nothing -- an object is reused only if the bytes it was compiled from are the bytes in the tree."""
    h = hashlib.sha256()
    for x in extra:
        h.update(x.encode() + b"\0")
    for p in paths:
        h.update(os.path.basename(p).encode() + b"\0")
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


LAST_BUILD = {"compiled": [], "reused": [], "linked": False}       # what the last build() call did (__graft_entry__.build reports it)


def build(force=False, verbose=True):
    bdir = os.path.join(CSRC, "_build")
    os.makedirs(bdir, exist_ok=True)
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    headers.append(os.path.join(os.path.dirname(os.path.dirname(CSRC)), "include", "ppt_hip.h"))
    stamp_path = os.path.join(bdir, "stamps.json")
    try:
        with open(stamp_path) as fh:
            stamps = json.load(fh)
    except (OSError, ValueError):
        stamps = {}
    jobs, objs, new_stamps = [], [], {}
    LAST_BUILD.update(compiled=[], reused=[], linked=False)
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(bdir, f[:-4] + ".o")
        objs.append(obj)
        flags = COMMON + PER_FILE.get(f, [])
        new_stamps[f] = _digest([src] + headers, flags)
        if force or not os.path.exists(obj) or stamps.get(f) != new_stamps[f]:
            jobs.append([HIPCC] + flags + ["-c", src, "-o", obj])
            LAST_BUILD["compiled"].append(f)
        else:
            LAST_BUILD["reused"].append(f)

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
    new_stamps["__link__"] = hashlib.sha256("".join(new_stamps[f] for f in sorted(new_stamps)).encode()).hexdigest()
    if jobs or force or not os.path.exists(OUT) or stamps.get("__link__") != new_stamps["__link__"]:
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs)
        LAST_BUILD["linked"] = True
    with open(stamp_path, "w") as fh:
        json.dump(new_stamps, fh, indent=0, sort_keys=True)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
