"""Error metrics shared by the GPU parity tests.

Two numbers per comparison, both relative to the reference tensor:
  * ``rel_l2``  = ||got - ref||_2 / ||ref||_2      (a tensor that is wrong everywhere by a few % of its peak FAILS this)
  * ``max_rel`` = max|got - ref| / max|ref|        (one bad element FAILS this)
Every comparison is also appended to ``gpurun_out/parity_metrics.jsonl`` (when that directory can be created) so the
measured values behind the tolerance table in DESIGN.md §4 come from the test run itself.
"""
import json
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LOG = os.path.join(ROOT, "gpurun_out", "parity_metrics.jsonl")


def metrics(got, ref):
    got = torch.as_tensor(got).detach().to(torch.float64).cpu()
    ref = torch.as_tensor(ref).detach().to(torch.float64).cpu()
    assert got.shape == ref.shape, f"{tuple(got.shape)} vs {tuple(ref.shape)}"
    d = got - ref
    return {"rel_l2": float(d.norm() / max(float(ref.norm()), 1e-30)),
            "max_rel": float(d.abs().max() / max(float(ref.abs().max()), 1e-30)),
            "max_abs": float(d.abs().max()), "ref_max": float(ref.abs().max()), "finite": bool(torch.isfinite(got).all())}


def record(what, m, **extra):
    try:
        os.makedirs(os.path.dirname(_LOG), exist_ok=True)
        with open(_LOG, "a") as f:
            f.write(json.dumps({"what": what, **m, **extra}) + "\n")
    except OSError:
        pass


def check(got, ref, what, l2_tol, max_tol, **extra):
    """assert both metrics; returns them"""
    m = metrics(got, ref)
    record(what, m, l2_tol=l2_tol, max_tol=max_tol, **extra)
    assert m["finite"], f"{what}: non-finite values"
    assert m["rel_l2"] <= l2_tol, f"{what}: rel-L2 {m['rel_l2']:.3e} > {l2_tol:.1e}"
    assert m["max_rel"] <= max_tol, f"{what}: max|err|/max|ref| {m['max_rel']:.3e} > {max_tol:.1e}"
    return m
