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


def ddim_net_terms(coeffs, x_prev, x_next):
    """(eps, x0), fp64, of the epsilon-prediction DDIM step (eta = 0) that took ``x_prev`` to ``x_next``:
    x_next = A x_prev + B eps with (alpha_bar_t, alpha_bar_prev) = ``coeffs``.  Recovers what the NETWORK contributed at that step from
    two consecutive history rows of either chain — the latents themselves grow ~14x over a 50-step chain with random-init weights, so
    their rel-L2 mostly measures that common mode; eps / x0 on each chain's OWN trajectory show a compounding divergence if there is one."""
    a_t, a_prev = [float(v) for v in coeffs]
    A = (a_prev / a_t) ** 0.5
    B = (1 - a_prev) ** 0.5 - A * (1 - a_t) ** 0.5
    eps = (x_next.double() - A * x_prev.double()) / B
    return eps, (x_prev.double() - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5


def oracle_chain_metrics(cfg, sd_r, hist, lat0, enc2, dtype, steps, marks, guidance_scale=7.5, ip_scale=0.4, num_tokens=4, log=None):
    """The fp32 CPU oracle loop (oracle/unet.py + oracle/ddim.py; reference models/pipelines.py:406-453) for ONE image against the HIP
    engine's history of that image: ``hist`` fp32 [steps + 1, 4, h, w], ``lat0`` its initial latents, ``enc2`` [2, L, ctx] =
    (negative, positive) conditioning rows.  -> {step: metrics of the latents + "eps" / "x0" sub-metrics}"""
    from oracle import ddim as oddim
    from oracle import unet as ou
    osch = oddim.DDIMSchedule()
    osch.set_timesteps(steps)
    ref, encr, curve = lat0.clone(), enc2.to(dtype).float(), {}
    for i, t in enumerate(osch.timesteps.tolist()):
        mi = torch.cat([ref] * 2).to(dtype).float()                     # `.half()` of pipelines.py:414 in the storage dtype
        prev = ref
        ref = oddim.step_epilogue(osch, ou.unet_forward(cfg, sd_r, mi, t, encr, ip_scale=ip_scale, num_tokens=num_tokens), t, ref, guidance_scale)
        if i + 1 in marks:
            m = metrics(hist[i + 1], ref)
            eps_r, x0_r = ddim_net_terms(osch.coeffs(t), prev, ref)
            eps_h, x0_h = ddim_net_terms(osch.coeffs(t), hist[i], hist[i + 1])
            m["eps"], m["x0"] = metrics(eps_h, eps_r), metrics(x0_h, x0_r)
            curve[i + 1] = m
            if log is not None:
                log(i + 1, m)
    return curve
