"""GPU: ``DenoiseEngine``'s per-schedule time-projection table (round 4).

The UNet's time path (models/unet_2d_condition.py:829-856 + every ResnetBlock2D's ``time_emb_proj``) depends on the timestep only; the engine computes one row per step of
its schedule with the launches ``unet.time_embed`` issues per step and gathers the row by the device step counter.  The rows are the per-step path's bits, so the whole
denoising history must be identical with the table on and off, eager and graph-replayed, and a weight change must rebuild it.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _engine_history(unet, cfg, lat, enc, use_graph, table):
    from theatergen_amd.pipelines import DenoiseEngine
    old = os.environ.get("TG_TPROJ_TABLE")
    os.environ["TG_TPROJ_TABLE"] = "1" if table else "0"
    try:
        eng = DenoiseEngine(unet, None, n_img=lat.shape[0], height=128, width=128, num_inference_steps=4, guidance_scale=7.5, enc_len=81, use_graph=use_graph)
        eng.set_conditioning(enc)
        h = eng.run(lat).clone()
        assert (eng._tproj_table is not None) == table
        return h, eng
    finally:
        if old is None:
            del os.environ["TG_TPROJ_TABLE"]
        else:
            os.environ["TG_TPROJ_TABLE"] = old


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_time_projection_table_is_bit_identical(dtype):
    from tests.test_hotpath_gpu import _build
    from theatergen_amd import config
    cfg = config.tiny()
    unet, _ = _build(cfg, dtype)
    g = torch.Generator().manual_seed(21)
    lat = torch.randn(2, 4, 16, 16, generator=g)
    enc = (torch.randn(4, 81, cfg.cross_attention_dim, generator=g) * 0.5).to(DEV, dtype)
    ref, _ = _engine_history(unet, cfg, lat, enc, use_graph=False, table=False)
    for use_graph in (False, True):
        h, eng = _engine_history(unet, cfg, lat, enc, use_graph, table=True)
        assert torch.equal(h, ref), f"time-projection table changed the history (graph={use_graph})"
        assert eng._tproj_table.shape[:2] == (4, 4)
    # a weight of the time path changes: the table (and the captured graph) must follow
    with torch.no_grad():
        unet.time_embedding.linear_2.bias.add_(0.25)
    ref2, _ = _engine_history(unet, cfg, lat, enc, use_graph=False, table=False)
    assert not torch.equal(ref2, ref)
    old_table = eng._tproj_table
    h2 = eng.run(lat).clone()
    assert eng._tproj_table is not old_table and torch.equal(h2, ref2), "stale time-projection table after a weight update"
    with torch.no_grad():
        unet.time_embedding.linear_2.bias.sub_(0.25)
