"""The kernels hand their workgroup size to the device functions as a compile-time constant (Ctx::nthreads; wb_humanoid_mpc_amd/csrc/hsqp_capi.hip), not as
blockDim.x: a launch with another block size would run the item loops with the wrong stride without any error.  This reads the source: every kernel
whose Ctx carries a constant is launched with exactly that constant everywhere, and its __launch_bounds__ names it."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "wb_humanoid_mpc_amd", "csrc", "hsqp_capi.hip")
LAUNCH = r"(?:hipLaunchKernelGGL|HSQP_LAUNCH)\("   # HSQP_LAUNCH: hipLaunchKernelGGL behind the LDS-poisoning test aid (same arguments)


def _kernels():
    lines = open(SRC).read().split("\n")
    out = {}
    for i, l in enumerate(lines):
        m = re.search(r"const Ctx ctx\{\(int\)threadIdx\.x, ([^,]+),", l)
        if not m:
            continue
        j = i
        while j >= 0 and "__global__" not in lines[j]:
            j -= 1
        head = lines[j] + " " + lines[j + 1]
        name = re.search(r"void (\w+)\s*\(", head).group(1)
        bounds = re.search(r"__launch_bounds__\(([^,)]+(?:\([^)]*\))?[^,)]*)", lines[j])
        out[name] = (m.group(1).strip(), bounds.group(1).strip() if bounds else None)
    return out


def test_constant_workgroup_sizes_match_the_launches():
    src = open(SRC).read()
    kernels = _kernels()
    assert len(kernels) >= 20
    constant = {k: v for k, v in kernels.items() if "blockDim" not in v[0]}
    assert {"k_riccati", "k_project", "k_scan_combine", "k_lq_chain", "k_lq_limb", "k_lq_rows", "k_value_quad"} <= set(constant)
    for name, (size, bounds) in constant.items():
        launches = re.findall(LAUNCH + name + r"(?:<[^>]*>)?,\s*dim3\((?:[^;]*?)\),\s*dim3\(([^)]*)\)", src)
        # launches through dim3 variables: `const dim3 qblock(EXPR);` ... HSQP_LAUNCH(k, qgrid, qblock, ...)
        for var in re.findall(LAUNCH + name + r"(?:<[^>]*>)?,\s*\w+,\s*(\w+),", src):
            decl = re.search(r"dim3 (?:\w+\([^;]*\),\s*)?" + var + r"\(([^;]*?)\);", src)
            assert decl, (name, var)
            launches.append(decl.group(1))
        assert launches, name
        for block in launches:
            assert block.strip() == size, (name, block, size)
        assert bounds is not None and bounds.replace(" ", "") == size.replace(" ", ""), (name, bounds, size)
