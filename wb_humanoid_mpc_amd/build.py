"""Build the HIP library in-tree (hipcc cross-compiles gfx950 without a GPU)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libhsqp_hip.so")


def _hipcc():
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def build_hip(force=False, verbose=False, extra=(), phase_profile=False):
    """phase_profile=True builds libhsqp_hip_prof.so with -DHSQP_PHASE_PROFILE (tools/phase_profile.py)."""
    global LIB
    lib = LIB.replace(".so", "_prof.so") if phase_profile else LIB
    if phase_profile:
        extra = tuple(extra) + ("-DHSQP_PHASE_PROFILE",)
    return _build(lib, force, verbose, extra)


def _build(LIB, force, verbose, extra):
    srcs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC))] + [os.path.join(_HERE, "..", "include", "hsqp.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(s) for s in srcs):
        return LIB
    # -amdgpu-sched-strategy=iterative-ilp: the kernels are chains of short dependent phases at an occupancy fixed by LDS and
    # launch bounds, so scheduling for ILP instead of for occupancy pays (measured A/B in one run, tools/gpu_variants.sh:
    # 9.53 vs 9.88 ms per iteration; max-ilp: no gain; iterative-minreg: 11.2 ms; any -unroll-threshold change: Riccati 2.5x slower)
    sched = os.environ.get("HSQP_SCHED_STRATEGY", "iterative-ilp")   # (tuning builds: "" = the compiler's default)
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", *(["-mllvm", "-amdgpu-sched-strategy=" + sched] if sched else []), "-std=c++17", "-shared", "-fPIC",
           os.path.join(CSRC, "hsqp_capi.hip"), os.path.join(CSRC, "hsqp_comm.hip"), "-ldl", "-o", LIB, *extra]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    import sys
    print(build_hip(force="-f" in sys.argv, verbose=True,
                    extra=("-Rpass-analysis=kernel-resource-usage",) if "-r" in sys.argv else ()))
