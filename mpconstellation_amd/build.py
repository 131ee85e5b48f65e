"""In-tree build of libmpcx.so for gfx950 (hipcc cross-compiles without a GPU)."""
import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmpcx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -O2: measured 1.7 % faster than -O3 on solve_kernel (A/B on one box, profiles/tools/ab_timing.py); -Os is 20 % slower
# -ffp-contract=on: a*b+c is fused where the SOURCE writes it in one expression, never across statements.  hipcc's default
# (fast) lets the optimiser fuse across statements, and it did so differently in the two compilations of solve.hip
# (solve_kernel / solve_kernel2w: same expressions, results 1e-13 apart); with `on` the two kernels are bit-identical
# (profiles/r04/two_wave_check.txt) at the same speed (A/B on one box: 1.362 vs 1.382 ms at S64, 5.675 vs 5.678 at S4096).
FLAGS = ["--offload-arch=gfx950", "-O2", "-ffp-contract=on", "-std=c++17", "-fPIC", "-shared", "-fno-gpu-rdc", "-pthread"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + \
        glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jobs=None):
    """hipcc every translation unit to an object file side by side (the solver's headers are compiled four times -- solve.hip,
    solve2w.hip, solve_lds.hip, solve_tp.hip: one after the other they were a 46 s build), then one link."""
    if not force and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    cflags = [f for f in FLAGS if f not in ("-shared",)]
    srcs = sources()
    objs = [os.path.join(objdir, os.path.basename(src)[:-4] + ".o") for src in srcs]

    def compile_one(pair):
        src, obj = pair
        cmd = [HIPCC] + cflags + ["-c", "-o", obj, src]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=jobs or min(len(srcs), os.cpu_count() or 1)) as pool:
        list(pool.map(compile_one, zip(srcs, objs)))
    cmd = [HIPCC] + FLAGS + ["-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force=True, verbose=True)
