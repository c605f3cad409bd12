"""TEST INFRASTRUCTURE: builds (on the machine it runs on: g++ -O3 -march=native -fopenmp) and drives oracle/cpu_baseline.cpp, the
timed CPU baseline of bench.py.  Only tests/ and bench.py's cpu_baseline leg import this."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "cpu_baseline.cpp")
LIB = os.path.join(_HERE, "_native", "libcpu_baseline.so")     # built per machine, never shipped (git- and gpurun-ignored)
FLAGS = ["-O3", "-march=native", "-std=c++17", "-fPIC", "-fopenmp", "-Wno-unknown-pragmas", "-shared"]
_dp = C.POINTER(C.c_double)


def build(force=False):
    csrc = os.path.join(_HERE, "..", "wb_humanoid_mpc_amd", "csrc")
    deps = [SRC] + [os.path.join(csrc, f) for f in os.listdir(csrc)]
    if force or not os.path.exists(LIB) or any(os.path.getmtime(d) > os.path.getmtime(LIB) for d in deps):
        os.makedirs(os.path.dirname(LIB), exist_ok=True)
        subprocess.check_call(["g++", *FLAGS, SRC, "-o", LIB])
    return LIB


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cores():
    """Hardware threads this process may actually use: the affinity mask, capped by the cgroup CPU quota (a container with
    `cpu.max = 1600000 100000` gets 16 CPUs' worth of time however many threads os.cpu_count() reports)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
            if q != "max":
                quota = float(q) / float(period)
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q, period = float(f.read()), float(g.read())
                if q > 0:
                    quota = q / period
        except (OSError, ValueError):
            pass
    return max(1, min(n, int(quota + 0.5))) if quota else n, n, quota


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo among the CPUs this process may run on: SMT siblings count once."""
    try:
        allowed = os.sched_getaffinity(0)
        seen, cpu, pkg = set(), None, None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k = k.strip()
            if k == "processor":
                cpu = int(v)
            elif k == "physical id":
                pkg = int(v)
            elif k == "core id" and cpu in allowed:
                seen.add((pkg, int(v)))
        return len(seen) or None
    except (OSError, ValueError):
        return None


class CpuBaseline:
    def __init__(self, model):
        self.lib = C.CDLL(build())
        self.lib.cpub_create.restype = C.c_void_p
        self.h = C.c_void_p(self.lib.cpub_create(C.byref(model.desc)))
        if not self.h.value:
            raise RuntimeError("cpub_create failed")

    def __del__(self):
        try:
            self.lib.cpub_destroy(self.h)
        except Exception:
            pass

    def iterate(self, x_init, x, u, par, dt, outer=1, inner=1, iterations=1):
        """(dx[B,N+1,58], du[B,N,35], perf[B,6] = cost, dyn, eq before / after) of the last of `iterations` iterations."""
        x_init, x, u, par = (np.ascontiguousarray(a, dtype=np.float64) for a in (x_init, x, u, par))
        B, N = u.shape[0], u.shape[1]
        dx, du, perf = np.zeros_like(x), np.zeros_like(u), np.zeros((B, 6))
        rc = self.lib.cpub_iterate(self.h, B, N, C.c_double(dt), x_init.ctypes.data_as(_dp), x.ctypes.data_as(_dp), u.ctypes.data_as(_dp),
                                   par.ctypes.data_as(_dp), int(outer), int(inner), int(iterations), dx.ctypes.data_as(_dp), du.ctypes.data_as(_dp),
                                   perf.ctypes.data_as(_dp))
        if rc != 0:
            raise RuntimeError(f"cpu baseline failed: {rc}")
        return dx, du, perf
