"""The committed problem images (wb_humanoid_mpc_amd/data/g1_*.json) are what tools/export_g1_model.py derives from the reference's
URDF, task.info, reference.info and gait.info — re-derived here whenever /root/reference is mounted (the build container; skipped on
the GPU box, which only has the committed files)."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _same(a, b, path=""):
    if isinstance(a, dict):
        assert isinstance(b, dict) and set(a) == set(b), f"{path}: keys differ: {set(a) ^ set(b)}"
        for k in a:
            _same(a[k], b[k], f"{path}.{k}")
    elif isinstance(a, list):
        assert isinstance(b, list) and len(a) == len(b), f"{path}: length {len(a)} vs {len(b)}"
        for i, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{i}]")
    elif isinstance(a, float) or isinstance(b, float):
        assert abs(a - b) <= 1e-12 * max(1.0, abs(a)), f"{path}: {a} vs {b}"
    else:
        assert a == b, f"{path}: {a!r} vs {b!r}"


@pytest.mark.parametrize("formulation,name", [("wb", "g1_wb.json"), ("centroidal", "g1_centroidal.json")])
def test_committed_problem_image_is_the_export_of_the_reference_files(tmp_path, formulation, name):
    if not os.path.isdir("/root/reference/robot_models/unitree_g1"):
        pytest.skip("/root/reference is not mounted")
    spec = importlib.util.spec_from_file_location("export_g1_model", os.path.join(ROOT, "tools", "export_g1_model.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    out = tmp_path / name
    mod.main(str(out), formulation)
    fresh = json.load(open(out))
    committed = json.load(open(os.path.join(ROOT, "wb_humanoid_mpc_amd", "data", name)))
    fresh.pop("_generated_by", None); committed.pop("_generated_by", None)
    _same(committed, fresh, name)
