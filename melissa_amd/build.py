"""Build libmelissa_hip.so (hand-written HIP for gfx950) in-tree with hipcc.

``python -m melissa_amd.build`` or ``melissa_amd.build.build_library()``; hipcc cross-compiles
without a GPU.  The .so is git-ignored but travels to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libmelissa_hip.so")
SOURCES = ["fwd.hip", "env.hip", "grad.hip"]
HEADERS = ["common.hpp", "gemm_f32.hpp", "gemm_bf16.hpp", "gemm_split.hpp", "gemm_ring.hpp", "plan.hpp", "plan_masks.hpp", "attention.hpp", "heads.hpp", "episode_stream.hpp", os.path.join("..", "..", "include", "melissa_hip.h")]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return "hipcc"


HASH_PATH = LIB_PATH + ".srchash"


def source_hash() -> str:
    h = hashlib.sha256()
    h.update(os.environ.get("MEL_HIPCC_FLAGS", "").encode())
    for rel in SOURCES + HEADERS:
        path = os.path.join(CSRC, rel)
        if os.path.exists(path):
            h.update(rel.encode())
            h.update(open(path, "rb").read())
    return h.hexdigest()


def is_stale() -> bool:
    """Content based (a snapshot copy scrambles mtimes): stale iff the sources' hash differs from the one
    recorded next to the library when it was built."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    return open(HASH_PATH).read().strip() != source_hash()


def build_library(force: bool = False, verbose: bool = False) -> str:
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    if not force and not is_stale():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    # -ffp-contract=off: the env's float64 arithmetic and the fp32 radius rule must round exactly like
    # the reference's Python/NumPy expressions (no fused multiply-add unless written as fmaf).
    tmp = f"{LIB_PATH}.tmp.{os.getpid()}"          # atomic publish: concurrent ranks never see a partial file
    extra = os.environ.get("MEL_HIPCC_FLAGS", "").split()         # tuning experiments (-DMEL_ATT_G=8 ...)
    cmd = [_hipcc(), "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared",
           "-ffp-contract=off", *extra, *srcs, "-o", tmp]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(tmp):
            os.remove(tmp)
        raise RuntimeError(f"hipcc failed:\n{res.stdout}\n{res.stderr}")
    os.replace(tmp, LIB_PATH)
    with open(HASH_PATH + f".tmp.{os.getpid()}", "w") as f:
        f.write(source_hash())
    os.replace(HASH_PATH + f".tmp.{os.getpid()}", HASH_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
