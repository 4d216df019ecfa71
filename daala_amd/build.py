"""Build libdaalahip.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m daala_amd.build` or `daala_amd.build.build()`.  hipcc cross-compiles
without a GPU; the resulting daala_amd/lib/libdaalahip.so is git-ignored but
travels to the GPU box with the gpurun snapshot.

Two libraries come out of the same sources: lib/libdaalahip.so, the DEFAULT build (the product:
three environment switches, od_ctx.cuh), and lib/libdaalahip_exp.so, the same with
-DODHIP_EXPERIMENTS: superseded kernel generations, ablations and tuning knobs behind their
ODHIP_* environment switches, for tools/ and the cross-check tests (selected with
ODHIP_LIB=daala_amd/lib/libdaalahip_exp.so, daala_amd.EXPERIMENTS_LIB).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libdaalahip.so")
EXP_LIB = os.path.join(LIBDIR, "libdaalahip_exp.so")
EXP_OBJDIR = os.path.join(LIBDIR, "exp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
SOURCES = ["dct_kernels.hip", "lapped_kernels.hip", "pvq_kernels.hip", "pvq_bands.hip", "pvq_ref.hip", "pvq_refbands.hip", "image_kernels.hip", "dering_kernels.hip", "dering_cache.hip", "frame_cache.hip",
           "odhip_host.hip", "ctx.hip", "quant.hip", "pipeline.hip", "y4m.hip", "dist_kernels.hip", "rate_host.hip", "export_kernels.hip"]
# -ffp-contract=off is MANDATORY for the fp64 PVQ search (bit-exactness with
# gcc -O2 on x86-64, which emits no FMA); harmless for the integer kernels.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
         "-ffp-contract=off", "-fno-fast-math", "-Wall",
         "-Wno-unused-function"]


def _deps():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in files:
            out.append(os.path.join(root, f))
    out.append(os.path.join(os.path.dirname(HERE), "include", "daala_hip.h"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, experiments=True):
    """The product library first, compiled AND linked before anything of the experiments build is touched: a
    compile error in experiments-only code cannot break it; the experiments variant follows and
    is reported, not raised, when `experiments` is "optional"."""
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(EXP_OBJDIR, exist_ok=True)
    deps = _deps()
    variants = [(LIB, LIBDIR, [])]
    if experiments:
        variants.append((EXP_LIB, EXP_OBJDIR, ["-DODHIP_EXPERIMENTS"]))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
        return r.stderr

    for lib, objdir, extra in variants:
        objs = []
        jobs = []
        for src in SOURCES:
            obj = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(obj)
            if force or _stale(obj, deps):
                jobs.append([HIPCC] + FLAGS + extra + ["-c", os.path.join(CSRC, src), "-o", obj])
        try:
            with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
                for warn in ex.map(run, jobs):
                    if verbose and warn:
                        print(warn)
            if jobs or force or _stale(lib, objs):
                run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        except RuntimeError:
            if lib == LIB or experiments != "optional":
                raise
            print("daala_amd.build: the experiments variant failed to build (the product library is intact)",
                  file=sys.stderr)
    return LIB


def build_variant(name, extra_flags, verbose=False):
    """A/B builds (tools/ab_bench.sh): the default sources with extra compiler flags (-DODHIP_...=n) into
    daala_amd/lib_<name>/libdaalahip.so, selected at run time with ODHIP_LIB."""
    outdir = os.path.join(HERE, "lib_" + name)
    os.makedirs(outdir, exist_ok=True)
    lib = os.path.join(outdir, "libdaalahip.so")
    stamp = os.path.join(outdir, "flags.txt")
    flags_changed = not os.path.exists(stamp) or open(stamp).read() != " ".join(extra_flags)
    deps = _deps()
    jobs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(outdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if flags_changed or _stale(obj, deps):
            jobs.append([HIPCC] + FLAGS + list(extra_flags) + ["-c", os.path.join(CSRC, src), "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr))

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 4)) as ex:
        list(ex.map(run, jobs))
    if jobs or not os.path.exists(lib):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    with open(stamp, "w") as f:
        f.write(" ".join(extra_flags))
    return lib


if __name__ == "__main__":
    if "--variant" in sys.argv:
        i = sys.argv.index("--variant")
        print(build_variant(sys.argv[i + 1], sys.argv[i + 2:], verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
