"""pytest configuration: the `gpu` marker, package loading, golden-fixture helpers."""
from __future__ import annotations

import importlib.util
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
if os.path.join(ROOT, "oracle") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "layouts(*names): the solve-kernel layouts (TINYMPC_LAYOUT) a test of a `kernel_layout` module runs against")


def pytest_generate_tests(metafunc):
    """Modules with a `kernel_layout` fixture run every test once per solve-kernel layout: the module's LAYOUTS, or what the test's
    own `@pytest.mark.layouts(...)` names. Only combinations that apply are generated -- a skip in the GPU run is then a real
    condition (one GPU visible ...), not a parametrisation that does not apply."""
    if "kernel_layout" in metafunc.fixturenames:
        marker = metafunc.definition.get_closest_marker("layouts")
        layouts = marker.args if marker else getattr(metafunc.module, "LAYOUTS")
        metafunc.parametrize("kernel_layout", list(layouts), indirect=True)


def load_pkg():
    """Import the package directory `tinympc-matlab_amd/` (hyphenated, so not a plain import)."""
    import __graft_entry__ as ge

    return ge.load_package()


@pytest.fixture(scope="session")
def pkg():
    return load_pkg()


@pytest.fixture(scope="session")
def problems(pkg):
    return pkg.problems


def golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def rel_err(a, b, eps: float = 1e-300) -> float:
    """Per-array relative error ||a-b||_inf / max(||b||_inf, eps) (parity traps)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.max(np.abs(a - b)) / max(float(np.max(np.abs(b))), eps))


def settings_from_golden(g: dict) -> dict:
    out = {}
    for k, v in g.items():
        if k.startswith("set_"):
            out[k[4:]] = v.item() if hasattr(v, "item") else v
    return out


def problem_from_golden(pkg, g: dict):
    """Rebuild the Problem a fixture was generated from, using only the fixture's own inputs."""
    P = pkg.problems
    p = P.Problem("golden", g["A"], g["B"], g["Q"], g["R"], int(g["N"]), float(g["rho"]), g["x0"])
    if int(g["has_bounds"]):
        p.x_min, p.x_max, p.u_min, p.u_max = g["x_min"], g["x_max"], g["u_min"], g["u_max"]
    p.x_ref, p.u_ref = g["Xref"], g["Uref"]
    return p
