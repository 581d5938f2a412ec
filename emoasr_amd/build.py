"""Build libemoasr_hip.so (hipcc, gfx950) and the C oracle helpers in-tree.

Usage: python -m emoasr_amd.build [--force]
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libemoasr_hip.so")
SOURCES = ["api.hip", "gemm.hip", "layernorm.hip", "elementwise.hip", "convmodule.hip",
           "subsample.hip", "ctc.hip", "attention.hip", "optim.hip", "feats.hip", "decoder.hip", "rnnt.hip", "layer.hip", "decode_rt.hip", "distill.hip", "gemm_big.hip", "convfused.hip", "rowlin.hip", "decode_coop.hip", "lstm_coop.hip", "rnnt_greedy.hip", "rnnt_beam.hip", "ctc_beam_host.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result", "-Wno-unused-value", "-Wno-comment",
         "-ffp-contract=off"]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def build(force=False, verbose=True, variant=None, defines=()):
    """variant / defines: a second library build/libemoasr_hip_<variant>.so compiled with extra -D flags (compile-time A/B on one
    box: EMOASR_HIP_LIB selects it); the default build is untouched by it."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build") if not variant else os.path.join(HERE, "build", "variant_" + variant)
    LIB = globals()["LIB"] if not variant else os.path.join(HERE, "build", f"libemoasr_hip_{variant}.so")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, h) for h in ("common.h", "mma.h")]
    headers.append(os.path.join(HERE, "..", "include", "emoasr_hip.h"))
    sources = SOURCES
    flags = FLAGS + ["-D" + d for d in defines]
    if variant:   # compiler-option A/Bs: extra hipcc flags for a VARIANT build only (EMOASR_HIPCC_FLAGS="-mllvm -amdgpu-...")
        flags = flags + os.environ.get("EMOASR_HIPCC_FLAGS", "").split()
    obj_of = lambda src: os.path.join(objdir, src.replace("/", "_").replace(".hip", ".o"))
    jobs = []
    for src in sources:
        s = os.path.join(CSRC, src)
        o = obj_of(src)
        if force or _newer(s, o) or any(_newer(h, o) for h in headers):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        cmd = [hipcc] + flags + ["-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [obj_of(s) for s in sources]
    if jobs or not os.path.exists(LIB):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-z,defs", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    # python -m emoasr_amd.build [--force] [--variant NAME -DMACRO ...]
    _v = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    build(force="--force" in sys.argv, variant=_v, defines=[a[2:] for a in sys.argv if a.startswith("-D")])
