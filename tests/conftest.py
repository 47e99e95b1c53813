import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not errored) on a box without a GPU: plain `pytest tests` stays green on CPU."""
    import torch

    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs a real MI355X (torch.cuda.is_available() is False)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---- parity margins: every comparison whose bound may exceed 1e-4 goes through `margins`, which records
# {case, key, err, floor, limit} and asserts err <= max(tol, 2 x floor).  The floor is the REFERENCE's own fp32-vs-fp64
# difference for that quantity (tests/golden/*_f64.npz, harness_reference.npz).  The table is written at session end.
_MARGIN_ROWS = []


def _rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    return ((a - b).abs().max() / (b.abs().max() + 1e-30)).item()


@pytest.fixture(scope="session")
def margins():
    def check(case, key, got, ref32, ref64=None, tol=1e-4, floor=None, self_noise=()):
        """assert rel(got, ref32) <= max(tol, 2 * floor).  floor = the largest of rel(ref32, ref64) (the reference's fp32 run
        against its float64 run) and the `self_noise` figures: how far the reference's OWN fp32 output moves when its
        inputs move by what no fp32 implementation can avoid (one ulp on the rays; 1e-5 on the encoder features)."""
        err = _rel(got, ref32)
        if floor is None:
            floor = _rel(ref32, ref64) if ref64 is not None else 0.0
        floor64 = floor
        for f in self_noise:
            floor = max(floor, float(f))
        limit = max(tol, 2.0 * floor)
        _MARGIN_ROWS.append({"case": case, "key": key, "err": float(f"{err:.3e}"), "floor": float(f"{floor:.3e}"),
                             "floor_fp64": float(f"{floor64:.3e}"), "limit": float(f"{limit:.3e}"), "needs_floor": bool(err > tol),
                             "ok": bool(err <= limit)})
        assert err <= limit, {"case": case, "key": key, "err": err, "floor": floor, "limit": limit}
        return err

    def record(case, rows):
        for r in rows:
            _MARGIN_ROWS.append({"case": case, **r})

    check.record = record
    return check


def pytest_sessionfinish(session, exitstatus):
    if not _MARGIN_ROWS:
        return
    import json

    out_dir = os.path.join(ROOT, "gpurun_out") if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else os.path.join(ROOT, "profiles")
    path = os.environ.get("NJF_MARGINS_OUT", os.path.join(out_dir, "r02_parity_margins.json"))
    summary = {"rule": "err <= max(1e-4, 2 x floor); err, floor = max|a-b| / max|b| (norm-wise); floor = the reference's own "
                       "fp32-vs-fp64 difference for that quantity (floor_fp64) or, for outputs downstream of sample placement / "
                       "the encoder, the larger of it and the movement of the reference's fp32 output under a one-ulp "
                       "perturbation of the rays / a 1e-5 perturbation of the encoder features (tests/golden/make_golden_r02.py)",
               "rows": len(_MARGIN_ROWS), "rows_over_1e-4": sum(r["needs_floor"] for r in _MARGIN_ROWS),
               "failed": sum(not r["ok"] for r in _MARGIN_ROWS), "table": _MARGIN_ROWS}
    with open(path, "w") as f:
        json.dump(summary, f, indent=0)


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    import torch

    def load(name):
        with np.load(os.path.join(GOLDEN, name + ".npz")) as f:
            return {k: torch.from_numpy(f[k]) for k in f.files}

    return load
