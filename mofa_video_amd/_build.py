"""Build libmofa_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

One object per translation unit under ``mofa_video_amd/build/`` (git-ignored), compiled in parallel, then one link:
``build(force=True)`` is a real recompile of every kernel in about 20 s; ``force=False`` recompiles only the units whose
sources (or shared headers) are newer than their object.  ``probe=True`` builds ``tools/libmofa_hip_probe.so`` instead:
the same library plus the K-loop timing variants and the cycle-trace hook of tools/igemm8_probe.py (-DMOFA_PROBE), which
the product library does not carry."""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmofa_hip.so")
PROBE_LIB = os.path.join(HERE, "..", "tools", "libmofa_hip_probe.so")
SOURCES = ["igemm.hip", "igemm8.hip", "igemm320.hip", "ff320.hip", "lin320.hip", "attention.hip", "norm.hip", "elementwise.hip", "softsplat.hip", "output.hip",
           "cmp_ops.hip", "frontend.hip"]
HEADERS = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "igemm_common.h"), os.path.join(CSRC, "igemm_pipe.h"),
           os.path.join(HERE, "..", "include", "mofa_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build(force=False, verbose=False, probe=False, variant=None, defines=()):
    """variant / defines: an A/B build ``tools/libmofa_hip_<variant>.so`` with extra -D flags (bench.py / the tools take it with
    ``--lib``); experiments only, the product library is always built without them"""
    objdir = os.path.join(HERE, "build", variant or ("probe" if probe else "lib"))
    os.makedirs(objdir, exist_ok=True)
    lib = os.path.join(HERE, "..", "tools", f"libmofa_hip_{variant}.so") if variant else (PROBE_LIB if probe else LIB)
    hdr_t = max(os.path.getmtime(h) for h in HEADERS if os.path.exists(h))
    flags = FLAGS + (["-DMOFA_PROBE"] if probe else []) + [f"-D{d}" for d in defines]
    jobs = []
    for s in SOURCES:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s.replace(".hip", ".o"))
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([_hipcc(), *flags, "-c", src, "-o", obj])
    if not jobs and os.path.exists(lib) and all(os.path.getmtime(lib) >= os.path.getmtime(os.path.join(objdir, s.replace(".hip", ".o")))
                                                for s in SOURCES):
        return lib

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *[os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES],
         "-o", lib])
    return lib


if __name__ == "__main__":
    import sys
    var = sys.argv[sys.argv.index("--variant") + 1] if "--variant" in sys.argv else None
    print(build(force="--incremental" not in sys.argv, verbose=True, probe="--probe" in sys.argv, variant=var,
                defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
