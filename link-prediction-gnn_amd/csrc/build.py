#!/usr/bin/env python3
"""Build liblpg_hip.so in-tree:  hipcc --offload-arch=gfx950, one object per .hip, then link.

Cross-compiles without a GPU.  Objects are rebuilt only when the source (or a header) is newer.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liblpg_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
         "-fvisibility=hidden", "-Wall", "-Wno-unused-function", "-DNDEBUG"]


# per-file additions.  tuple_wl.hip: hipcc's SLP vectoriser pairs the products of two rows into v_pk_fma_f32, which wants
# every weight duplicated in a register pair -- 64 registers for 32 weights, 187 VGPRs and 60 spills at the kernel's
# 128-register bound; without it 90-103 VGPRs, no scratch.
EXTRA = {"tuple_wl.hip": ["-fno-slp-vectorize"]}


def hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def sources():
    return sorted(f for f in os.listdir(HERE) if f.endswith(".hip"))


def _newer(a, b):
    return not os.path.exists(b) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, jobs=None, tuning=False):
    """tuning=True builds liblpg_hip_tuning.so with -DLPG_TUNING (the sweep scripts under profiles/ load it through
    LPG_LIB_VARIANT=tuning): the options table is seeded from LPG_* environment variables and the measurement-only
    kernel variants that return wrong results on purpose exist.  The product library has neither."""
    LIB = os.path.join(HERE, "liblpg_hip_tuning.so" if tuning else "liblpg_hip.so")
    OBJ = os.path.join(HERE, "build_tuning" if tuning else "build")
    FLAGS = globals()["FLAGS"] + (["-DLPG_TUNING"] if tuning else [])
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "..", "include", "lpg_hip.h"))
    todo, objs = [], []
    for s in sources():
        src = os.path.join(HERE, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj) or any(_newer(h, obj) for h in headers):
            todo.append((src, obj))

    def cc(pair):
        src, obj = pair
        cmd = [hipcc()] + FLAGS + EXTRA.get(os.path.basename(src), []) + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        return src, r.returncode, r.stdout + r.stderr

    if todo:
        with cf.ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 4)) as ex:
            for src, rc, out in ex.map(cc, todo):
                if verbose and out.strip():
                    print(out, file=sys.stderr)
                if rc != 0:
                    raise RuntimeError(f"hipcc failed on {src}:\n{out}")
                if verbose:
                    print(f"[lpg build] compiled {os.path.basename(src)}")
    if todo or not os.path.exists(LIB):
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[lpg build] linked {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, tuning="--tuning" in sys.argv)
