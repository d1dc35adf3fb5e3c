"""CPU oracle for the TheaterGen per-character denoising hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and there only as the checker / the CPU baseline — never as
the thing measured or shipped.  The product package (``theatergen_amd``) never
imports this package and fails loudly when its HIP library is missing.

What it is: a plain PyTorch-CPU fp32 restatement of the reference algorithm,
op for op in the reference's own order (``baddbmm -> softmax -> bmm`` attention,
two independent softmaxes for the decoupled IP cross-attention, ...).  Every
function cites the reference ``file:line`` it follows (paths relative to
``/root/reference``).

Parity pinning (see DESIGN.md "Oracle"):
  * attention processors, Resampler, ImageProjModel, GEGLU/FeedForward, guidance
    losses, latent utilities, schedule: PINNED against the *imported* reference
    modules in the build container; the resulting input/output vectors are
    committed under ``tests/golden/`` together with ``tests/golden/make_golden.py``.
  * ResnetBlock2D / Downsample2D / Upsample2D / Timesteps / TimestepEmbedding /
    DDIMScheduler.step: third-party ``diffusers==0.21.4`` arithmetic whose
    source is NOT under /root/reference and is not installed here: restated
    from the documented 0.21.4 semantics -> "parity unpinned" for those rows.
"""
