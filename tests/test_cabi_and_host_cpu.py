"""CPU (no GPU): the C-ABI library loads and exports every symbol include/theatergen_hip.h declares (no compute
calls), and the host-side logic (scheduler tables, box geometry, phrase indices, parameter tables, processor
wiring, workload generation, sharding) matches the oracle / the golden vectors from the reference."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def test_header_symbols_are_exported_and_bound():
    from theatergen_amd import _lib, build
    header = open(os.path.join(ROOT, "include", "theatergen_hip.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t|const char\*)\s+(tg_[a-z0-9_]+)\s*\(", header, flags=re.M))
    assert len(declared) >= 20
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    h = _lib.lib()          # binds every symbol: AttributeError if one is missing
    for name in declared:
        assert getattr(h, name) is not None
    assert h.tg_version() == _lib.ABI_VERSION
    assert h.tg_last_error() is not None


def test_argument_errors_map_to_runtime_error():
    """Validation happens on the host side of the ABI, before any launch: works without a GPU."""
    import ctypes as C
    from theatergen_amd import _lib
    h = _lib.lib()
    d = _lib.GemmDesc()
    rc = h.tg_gemm(C.byref(d), None)
    assert rc == -1 and b"tg_gemm" in h.tg_last_error()
    with pytest.raises(RuntimeError, match="theatergen_hip error"):
        _lib.check(rc)
    a = _lib.AttnDesc()
    assert h.tg_attention(C.byref(a), None) == -1


def test_no_cpu_fallback():
    from theatergen_amd import ops
    x = torch.zeros(8, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.linear(x, torch.zeros(64, 64, dtype=torch.bfloat16))
    with pytest.raises(RuntimeError, match="GPU"):
        ops.layernorm(x, None, None)


def test_ddim_tables_match_oracle():
    from oracle.ddim import DDIMSchedule
    from theatergen_amd.scheduler import DDIMScheduler
    for steps in (50, 20, 30):
        o, s = DDIMSchedule(), DDIMScheduler()
        o.set_timesteps(steps)
        s.set_timesteps(steps)
        assert torch.equal(o.timesteps, s.timesteps)
        coef = s.coef_table()
        for i, t in enumerate(o.timesteps.tolist()):
            a_t, a_prev = o.coeffs(t)
            want = torch.stack([a_t ** 0.5, (1 - a_t) ** 0.5, a_prev ** 0.5, (1 - a_prev) ** 0.5])
            assert torch.equal(coef[i], want.float())
    s50 = DDIMScheduler()
    s50.set_timesteps(50)
    assert s50.timesteps[0].item() == 981 and s50.timesteps[-1].item() == 1     # steps_offset = 1 (generate.py:68-76)


def test_geometry_host_functions_vs_reference_golden():
    from theatergen_amd import utils as U
    gold = _load("geometry_latents")
    boxes = gold["geo.boxes"].tolist()
    sp = []
    for b in boxes:
        for (H, W) in ((64, 64), (16, 16), (8, 8), (96, 96)):
            sp.append(list(U.scale_proportion(b, H, W)) + list(U.scale_proportion(b, H, W, use_legacy=True)))
    assert np.array_equal(np.array(sp), gold["geo.scale_proportion"])
    assert np.array_equal(np.stack([U.proportion_to_mask(b, 64, 64, device="cpu").numpy() for b in boxes]), gold["geo.mask64"])
    cen = [U.get_centered_box(b) for b in boxes[:3]] + [U.get_centered_box(b, horizontal_center_only=False) for b in boxes[:3]] \
        + [U.get_centered_box(b, horizontal_center_only=False, vertical_placement="floor_padding", floor_padding=0.05) for b in boxes[:3]]
    np.testing.assert_allclose(np.array(cen), gold["geo.centered"], rtol=0, atol=0)
    masks = [torch.from_numpy(m) for m in gold["geo.masks"]]
    assert np.array_equal(np.array([U.binary_mask_to_box(m) for m in masks]), gold["geo.mask_box"])
    assert np.array_equal(torch.stack([U.binary_mask_to_box_mask(m, to_device=False) for m in masks]).numpy(), gold["geo.mask_box_mask"])
    np.testing.assert_allclose(np.array([U.binary_mask_to_center(m, normalize=True) for m in masks]), gold["geo.mask_center"], rtol=0, atol=0)
    # host path of shift_tensor (bool masks) and the quantisation of normalised offsets
    for (xo, yo), want in zip(gold["geo.shifts"].tolist(), gold["lat.align_masks"][:0]):
        pass
    m = masks[0]
    from oracle import box_geometry as geo
    for xo, yo in gold["geo.shifts"].tolist():
        assert torch.equal(U.shift_tensor(m, xo, yo, offset_normalized=True), geo.shift_tensor(m, xo, yo, offset_normalized=True))
    with pytest.raises(ValueError):
        U.binary_mask_to_box(torch.zeros(8, 8))
    from theatergen_amd.schedule import get_fast_schedule
    ts = torch.arange(981, 0, -20)
    assert np.array_equal(get_fast_schedule(ts, 10, 2).numpy(), gold["sched.fast_10_2"])
    assert np.array_equal(get_fast_schedule(ts, 49, 2).numpy(), gold["sched.fast_49_2"])


def test_prepare_mid_image_vs_reference_golden(tmp_path, monkeypatch):
    """the stage-1 -> stage-2 pixel paste (reference utils/latents.py:48-135; host-side integer work, so it runs here): bit-exact mask and canvas for two
    separate characters, overlapping boxes (the reference's uint8 mask sum wraps) and a box that leaves the canvas; the two PNGs land where the reference
    writes them; an empty mask fails the way the reference does"""
    from PIL import Image
    from tests.golden import gen_common as gc
    from theatergen_amd import latents as L
    gold = _load("mid_image")
    monkeypatch.chdir(tmp_path)
    for case in (0, 1, 2):
        masks, images, boxes = gc.mid_image_case(case)
        mask_img, canvas = L.prepare_mid_image("1.5", 3, masks, [Image.fromarray(i) for i in images], boxes)
        assert mask_img.mode == "L" and canvas.mode == "RGB" and canvas.size == (512, 512)
        assert np.array_equal(np.array(mask_img), gold[f"mid{case}.mask"]), f"case {case}: mask"
        assert np.array_equal(np.array(canvas), gold[f"mid{case}.image"]), f"case {case}: canvas"
    assert (gold["mid1.mask"] == 1).any(), "case 1 must exercise the wrapped overlap (mask value 1)"
    assert (tmp_path / "visualization" / "3vis_image.png").exists() and (tmp_path / "visualization" / "3vis_mask.png").exists()
    # numpy images work like PIL images (theatergen.py:177 appends arrays of the VAE output); the side effect can be switched off
    monkeypatch.setattr(L, "MID_IMAGE_DIR", None)
    masks, images, boxes = gc.mid_image_case(0)
    mask_img, canvas = L.prepare_mid_image("1.5", 9, masks, images, boxes)
    assert np.array_equal(np.array(canvas), gold["mid0.image"]) and not (tmp_path / "visualization" / "9vis_image.png").exists()
    with pytest.raises(TypeError):
        L.prepare_mid_image("1.5", 0, [torch.zeros(512, 512, dtype=torch.bool)], images[:1], boxes[:1])


def test_phrase_indices_vs_reference_golden():
    from tests.golden.gen_common import FakeTokenizer
    from theatergen_amd import guidance as G
    gold = _load("guidance")
    tok = FakeTokenizer()
    prompt = "a photo of a red cat and a small brown dog , park"
    pos, wti = G.get_phrase_indices(tok, prompt, ["a red cat", "a small brown dog"], words=["cat", "dog"], return_word_token_indices=True)
    assert pos[0] == gold["phrase.pos0"].tolist() and pos[1] == gold["phrase.pos1"].tolist() and wti == gold["phrase.wti"].tolist()
    pos2, newp = G.get_phrase_indices(tok, "a street", ["a blue car"], add_suffix_if_not_found=True)
    assert pos2[0] == gold["phrase.suffix.pos0"].tolist() and newp == str(gold["phrase.suffix.prompt"])


def test_parameter_tables_and_module_keys():
    import math
    from theatergen_amd import config, weights
    from theatergen_amd.unet import UNet2DConditionModel, install_ip_processors
    n15 = sum(math.prod(s) for k, s in weights.unet_param_shapes(config.sd15()).items() if "_ip" not in k)
    nxl = sum(math.prod(s) for k, s in weights.unet_param_shapes(config.sdxl()).items() if "_ip" not in k)
    assert n15 == 859_520_964 and nxl == 2_567_463_684          # the public parameter counts of SD-1.5 / SDXL UNets
    for cfg in (config.tiny(), config.tiny(linear=True), config.tiny(xl=True)):
        m = UNet2DConditionModel(cfg)
        install_ip_processors(m, num_tokens=4)
        shapes = weights.unet_param_shapes(cfg)
        sd = m.state_dict()
        assert set(sd.keys()) == set(shapes.keys())
        for k, v in sd.items():
            assert tuple(v.shape) == tuple(shapes[k]), k
    # processor table: names / order as diffusers' attn_processors (reference ip_adapter.py:95-119, 139-140)
    m = UNet2DConditionModel(config.tiny())
    names = list(m.attn_processors.keys())
    assert names[0] == "down_blocks.0.attentions.0.transformer_blocks.0.attn1.processor"
    assert names[1] == "down_blocks.0.attentions.0.transformer_blocks.0.attn2.processor"
    assert sum(n.startswith("mid_block") for n in names) == 2 and len(names) == 32
    # diffusers order = down (0-11), up (12-29), mid (30-31): the reference registers both block lists before mid_block
    # (models/unet_2d_condition.py:430-431), and IP-Adapter checkpoints index `ip_adapter.{1,3,..,31}` by that order
    assert all(n.startswith("down_blocks") for n in names[:12])
    assert names[12].startswith("up_blocks.1.attentions.0") and all(n.startswith("up_blocks") for n in names[12:30])
    assert names[30] == "mid_block.attentions.0.transformer_blocks.0.attn1.processor"
    assert names[31] == "mid_block.attentions.0.transformer_blocks.0.attn2.processor"


def test_ip_adapter_checkpoint_keys_in_diffusers_order():
    """A state dict keyed like a released IP-Adapter checkpoint (`{2i+1}.to_k_ip.weight` in the reference's processor
    order, hidden sizes of the SD-1.5 plan: reference ip_adapter/ip_adapter.py:98-116, 139-140) must load through
    `ModuleList(unet.attn_processors.values()).load_state_dict` and land on the layer it was made for."""
    import torch
    from theatergen_amd import config
    from theatergen_amd.attention_processor import IPAttnProcessor
    from theatergen_amd.ip_adapter import IPAdapter
    from theatergen_amd.pipelines import SDPipe
    from theatergen_amd.unet import UNet2DConditionModel
    cfg = config.sd15()
    with torch.device("meta"):
        unet = UNet2DConditionModel(cfg)
    # the reference's table, written out independently of the module tree: down 320,320,640,640,1280,1280 ; up-1 1280 x3,
    # up-2 640 x3, up-3 320 x3 ; mid 1280
    hidden = [320, 320, 640, 640, 1280, 1280] + [1280] * 3 + [640] * 3 + [320] * 3 + [1280]
    ad = IPAdapter.__new__(IPAdapter)
    ad.device, ad.num_tokens, ad.pipe, ad.dtype = "meta", 4, SDPipe.__new__(SDPipe), torch.float32
    ad.pipe.unet, ad.pipe.controlnet = unet, None
    ad.set_ip_adapter()
    procs = list(unet.attn_processors.values())
    assert len(procs) == 32
    sd = {}
    for i, h in enumerate(hidden):
        sd[f"{2 * i + 1}.to_k_ip.weight"] = torch.empty(h, cfg.cross_attention_dim, device="meta")
        sd[f"{2 * i + 1}.to_v_ip.weight"] = torch.empty(h, cfg.cross_attention_dim, device="meta")
    torch.nn.ModuleList(procs).load_state_dict(sd, assign=True)      # raises on any size / key mismatch
    for i, h in enumerate(hidden):
        p = procs[2 * i + 1]
        assert isinstance(p, IPAttnProcessor) and p.hidden_size == h and tuple(p.to_k_ip.weight.shape) == (h, cfg.cross_attention_dim)


def test_controlnet_tables_and_oracle_cpu():
    """Stage-2 ControlNet: parameter table == the public SD-1.5 ControlNet count, module keys == diffusers names, CN
    processors installed on the cross-attention only; the oracle runs on CPU and a freshly initialised ControlNet
    (zero convs) contributes exact zeros, with guess-mode / conditioning scales applied to random heads."""
    import math
    import torch
    from oracle import controlnet as oc
    from theatergen_amd import config, weights
    from theatergen_amd.attention_processor import AttnProcessor, CNAttnProcessor
    from theatergen_amd.controlnet import ControlNetModel, install_cn_processors
    assert sum(math.prod(s) for s in weights.controlnet_param_shapes(config.sd15()).values()) == 361_279_120
    cfg = config.tiny()
    m = ControlNetModel(cfg)
    shapes = weights.controlnet_param_shapes(cfg)
    sd = m.state_dict()
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    assert len(m.controlnet_down_blocks) == 12
    procs = install_cn_processors(m, num_tokens=4)
    assert all(isinstance(p, CNAttnProcessor) == n.endswith("attn2.processor") for n, p in procs.items())
    assert all(isinstance(p, AttnProcessor) for n, p in procs.items() if n.endswith("attn1.processor"))
    g = torch.Generator().manual_seed(0)
    x, enc, cond = torch.randn(2, 4, 16, 16, generator=g), torch.randn(2, 81, cfg.cross_attention_dim, generator=g), torch.rand(2, 3, 128, 128, generator=g)
    fresh = {k: v.float() for k, v in sd.items()}
    d, mid = oc.controlnet_forward(cfg, fresh, x, 500, enc, cond)
    assert len(d) == 12 and all(float(t.abs().max()) == 0.0 for t in d) and float(mid.abs().max()) == 0.0
    rnd = weights.random_controlnet_state_dict(cfg, seed=1)
    d1, m1 = oc.controlnet_forward(cfg, rnd, x, 500, enc, cond, conditioning_scale=1.0)
    d2, m2 = oc.controlnet_forward(cfg, rnd, x, 500, enc, cond, conditioning_scale=0.5)
    assert torch.allclose(d2[3], 0.5 * d1[3]) and torch.allclose(m2, 0.5 * m1)
    dg, mg = oc.controlnet_forward(cfg, rnd, x, 500, enc, cond, conditioning_scale=1.0, guess_mode=True)
    scales = torch.logspace(-1, 0, 13)
    assert torch.allclose(dg[0], d1[0] * scales[0]) and torch.allclose(mg, m1 * scales[-1])
    assert float(d1[0].abs().max()) > 0


def test_vae_decoder_tables_and_oracle_cpu():
    """VAE decode: parameter table == the public SD VAE decoder count (+ post_quant_conv), module keys
    == diffusers names, oracle decode shape / scaling-factor convention on CPU."""
    import math
    import torch
    from oracle import vae as ov
    from theatergen_amd import weights
    from theatergen_amd.vae import AutoencoderKL, sd_vae_config, tiny_vae_config
    assert sum(math.prod(s) for s in weights.vae_decoder_param_shapes(sd_vae_config()).values()) == 49_490_199
    cfg = tiny_vae_config()
    m = AutoencoderKL(cfg)
    shapes = dict(weights.vae_decoder_param_shapes(cfg))
    shapes.update(weights.vae_encoder_param_shapes(cfg))          # the module now carries the encoder half as well
    sd = m.state_dict()
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    rnd = weights.random_vae_decoder_state_dict(cfg, seed=2)
    lat = torch.randn(1, 4, 8, 8, generator=torch.Generator().manual_seed(0))
    img = ov.decode(cfg, rnd, lat)
    assert img.shape == (1, 3, 64, 64) and torch.isfinite(img).all()
    assert torch.allclose(ov.decode(cfg, rnd, lat * 2.0, scaling_factor=2 * cfg.scaling_factor), img, atol=1e-5)


def test_vae_encoder_tables_and_oracle_cpu():
    """VAE encode (second half): parameter table == the public SD VAE encoder count, module keys ==
    diffusers names, the oracle's bottom / right padded downsample halves even sizes, sampling convention."""
    import math
    import torch
    from oracle import vae as ov
    from theatergen_amd import weights
    from theatergen_amd.vae import AutoencoderKL, sd_vae_config, tiny_vae_config
    enc = weights.vae_encoder_param_shapes(sd_vae_config())
    assert sum(math.prod(s) for k, s in enc.items() if k.startswith("encoder.")) == 34_163_592
    full = sum(math.prod(s) for s in enc.values()) + sum(math.prod(s) for s in weights.vae_decoder_param_shapes(sd_vae_config()).values())
    assert full == 83_653_863                                      # the public parameter count of the SD VAE
    cfg = tiny_vae_config()
    m = AutoencoderKL(cfg)
    shapes = dict(weights.vae_encoder_param_shapes(cfg))
    shapes.update(weights.vae_decoder_param_shapes(cfg))
    sd = m.state_dict()
    assert set(sd.keys()) == set(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    rnd = weights.random_vae_state_dict(cfg, seed=2)
    img = torch.rand(1, 3, 64, 64, generator=torch.Generator().manual_seed(0)) * 2 - 1
    mo = ov.encode_moments(cfg, rnd, img)
    assert mo.shape == (1, 8, 8, 8) and torch.isfinite(mo).all()
    z = ov.sample_latents(cfg, mo, torch.zeros(1, 4, 8, 8))
    assert torch.allclose(z, cfg.scaling_factor * mo[:, :4])


def test_story_workload_and_sharding():
    from theatergen_amd import distributed as D
    from theatergen_amd import story
    jobs = story.story_jobs(3)
    assert len(jobs) == 8 and {j.turn for j in jobs} == {1, 2, 3, 4} and {j.char for j in jobs} == {0, 1}
    assert len({j.char_id for j in jobs}) == 2                     # two characters persist across the four turns
    assert len({j.text_seed for j in jobs}) == 8
    assert story.box_xyxy(0) == [40 / 512, 150 / 512, 230 / 512, 450 / 512]
    sh = story.shared_conditioning(64, 4, torch.float32, "cpu")
    tok = story.character_image_tokens([jobs[0].char_id, jobs[1].char_id], 64, 4, torch.float32, "cpu")
    cidx = {jobs[0].char_id: 0, jobs[1].char_id: 1}
    enc = story.job_conditioning(jobs, sh, tok, cidx, 64, torch.float32, "cpu")
    assert enc.shape == (16, 81, 64)
    assert torch.equal(enc[0, :77], sh["neg_text"][0]) and torch.equal(enc[8, 77:], tok[0]) and torch.equal(enc[9, 77:], tok[1])
    assert torch.equal(enc[8 + 2, 77:], tok[0])                     # same character, next turn -> same image tokens
    items = list(range(10))
    parts = [D.shard(items, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == items and parts[1] == [1, 5, 9]


def test_static_slots_identity_is_the_object_not_the_address():
    """The K / V^T and ControlNet-embedding caches (address + `_version` keys alias when the allocator reuses a
    freed block) are keyed by the tensor OBJECT through a weak reference: a new tensor at the same address misses, and a
    slot dies with its tensor."""
    import gc
    import torch
    from theatergen_amd.attention_processor import StaticSlots, tensor_version
    s = StaticSlots()
    a = torch.zeros(4, 8)
    s.put(a, {"v": 1})
    assert s.get(a)["v"] == 1
    alias = a.view(4, 8)                         # same storage, same address, same version: a different object -> miss
    assert alias.data_ptr() == a.data_ptr() and alias._version == a._version and s.get(alias) is None
    key_id = id(a)
    del a, alias
    gc.collect()
    assert len(s) == 0                           # the slot (and the buffers it holds) is released with the tensor
    b = torch.zeros(4, 8)                        # whatever id / address b gets, nothing stale can be served
    assert s.get(b) is None and key_id is not None
    with torch.inference_mode():
        t = torch.ones(2)
    assert tensor_version(t) is None and tensor_version(b) == 0


def test_gemm_planner_kernel_choice(monkeypatch):
    """tg_gemm_plan is host logic: which kernel / tile / split a descriptor gets (no launch, works without a GPU).
    kernel_kind: 0 GEMM, 1 implicit-GEMM conv, 2 LDS-halo conv, 3 big-tile GEMM, 4 slab conv (GroupNorm prologue), 5 loader / compute GEMM."""
    import ctypes as C
    from theatergen_amd import _lib
    h = _lib.lib()
    monkeypatch.delenv("TG_GEMM_FLAGS", raising=False)

    def conv(batch, hh, ww, cin, cout, c1=0, stride=1, force_tile=0, a_coef=None):
        d = _lib.GemmDesc()
        d.dtype, d.mode = 0, 1
        d.a0 = d.w = d.out = 16
        d.a1 = 16 if c1 else None
        d.c0, d.c1 = cin, c1
        oh, ow = (hh + 2 - 3) // stride + 1, (ww + 2 - 3) // stride + 1
        d.batch, d.in_h, d.in_w, d.out_h, d.out_w, d.stride, d.upsample = batch, hh, ww, oh, ow, stride, 0
        d.M, d.N, d.K = batch * oh * ow, cout, 9 * (cin + c1)
        d.ldc, d.out_scale, d.force_tile = cout, 1.0, force_tile
        d.a_coef = a_coef
        return d

    def plan(d):
        tm, tn, sp, kk = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
        assert h.tg_gemm_plan(C.byref(d), C.byref(tm), C.byref(tn), C.byref(sp), C.byref(kk)) == 0, h.tg_last_error()
        return tm.value, tn.value, sp.value, kk.value

    # SD-1.5 ResnetBlock convs at CFG batch 16 (the bench): slab kernel on the 64 / 32-wide maps, split in two on the 16-wide maps
    assert plan(conv(16, 64, 64, 320, 320)) == (128, 320, 1, 4)
    assert plan(conv(16, 64, 64, 640, 320, c1=320)) == (128, 320, 1, 4)
    assert plan(conv(16, 32, 32, 640, 640)) == (128, 320, 1, 4)
    assert plan(conv(16, 16, 16, 1280, 1280)) == (128, 320, 2, 4)
    d = conv(16, 16, 16, 1280, 1280)
    assert h.tg_gemm_workspace_bytes(C.byref(d)) == 32 * 4 * 2 * 128 * 320 * 4
    # CFG batch 2 (configs 1 / 4 / 5): 64 tiles do not fill 256 CUs -> the halo kernel; stride 2 -> implicit GEMM; N = 128 -> never
    assert plan(conv(2, 64, 64, 320, 320))[3] == 2
    assert plan(conv(16, 64, 64, 320, 320, stride=2))[3] == 1
    assert plan(conv(16, 64, 64, 128, 128))[3] != 4
    assert plan(conv(2, 64, 64, 320, 320, force_tile=11)) == (128, 320, 1, 4)
    # round 3, second pass: the small-M levels of the batch-2 plans that used to fall to the implicit-GEMM conv take the slab kernel on
    # generic patch tiles with a K split over the channel chunks (one round of work items on the 256 CUs); halo-eligible layers stay
    assert plan(conv(2, 48, 48, 640, 640)) == (128, 320, 3, 4)               # SD-2.1 level 1: 72 tiles x 3
    assert plan(conv(2, 48, 48, 640, 640, c1=640)) == (128, 320, 3, 4)
    assert plan(conv(2, 24, 24, 1280, 1280))[2:] == (5, 4)                   # SD-2.1 level 2: two 8 x 8 patches per tile, 36 tiles x 5
    assert plan(conv(2, 32, 32, 1280, 1280)) == (128, 320, 4, 4)             # SDXL level 2 at batch 2: 64 tiles x 4
    d = conv(2, 32, 32, 1280, 1280)
    assert h.tg_gemm_workspace_bytes(C.byref(d)) == 16 * 4 * 4 * 128 * 320 * 4
    assert plan(conv(2, 12, 12, 1280, 1280))[3] == 1                         # 12 x 12: no patch geometry, weight-streaming bound anyway
    assert plan(conv(16, 8, 8, 1280, 1280))[3] == 2                          # the 8 x 8 level keeps the halo kernel (measured)
    assert plan(conv(2, 8, 8, 1280, 1280))[3] == 1                           # 4 tiles: never enough work items
    monkeypatch.setenv("TG_GEMM_FLAGS", "2048")                               # dev switch: the round-3 first-pass rules only
    assert plan(conv(2, 48, 48, 640, 640))[3] == 1
    monkeypatch.delenv("TG_GEMM_FLAGS")
    # the dev switch turns the slab kernel off
    monkeypatch.setenv("TG_GEMM_FLAGS", "128")
    assert plan(conv(16, 64, 64, 320, 320))[3] == 2
    monkeypatch.delenv("TG_GEMM_FLAGS")
    # GroupNorm prologue outside the slab kernel is refused before any launch
    bad = conv(2, 8, 8, 128, 128, a_coef=16)
    assert h.tg_gemm(C.byref(bad), None) != 0 and b"slab conv kernel" in h.tg_last_error()

    def gemm(M, N, K, force_tile=0, geglu=0):
        d = _lib.GemmDesc()
        d.dtype, d.mode = 0, 0
        d.a0 = d.w = d.out = 16
        d.c0 = K
        d.M, d.N, d.K = M, N, K
        d.ldc, d.out_scale, d.force_tile, d.geglu = (N // 2 if geglu else N), 1.0, force_tile, geglu
        return d
    assert plan(gemm(65536, 320, 320))[:2] == (128, 128) and plan(gemm(65536, 320, 320))[3] == 0
    assert plan(gemm(65536, 2560, 320, geglu=1)) == (256, 256, 1, 3)          # fused GEGLU FF1 on the big tile
    # round 5: tile-count-aware 128 x 160 tiles where they fill whole rounds (512 / 256 tiles) and 128 x 128 does not (640 / 320); round 6 moves the
    # M = 16384 / 65536 ones of them to the ping-pong 256 x 160 tiles (kind 7; TG_PP bit 3)
    assert plan(gemm(4096, 1280, 1280))[:2] == (128, 160)
    assert plan(gemm(16384, 640, 2560)) == (256, 160, 1, 7) and plan(gemm(16384, 640, 640)) == (256, 160, 1, 7) and plan(gemm(16384, 1920, 640)) == (256, 160, 1, 7)
    assert plan(gemm(512, 320, 640, force_tile=25)) == (256, 160, 1, 7)
    monkeypatch.setenv("TG_PP", "7")
    assert plan(gemm(16384, 640, 2560))[:2] == (128, 160) and plan(gemm(16384, 640, 640))[:2] == (128, 160)
    assert plan(gemm(65536, 320, 320))[:2] == (128, 128) and plan(gemm(1024, 1280, 1280))[:2] == (64, 64)
    # round 6: the ping-pong 256 x 256 tiles (kernel_kind 7) where M, N are multiples of 256, K >= 640 and the tiles fill the persistent grid's rounds:
    # the FeedForward GEGLU GEMMs of the 32 x 32 / 16 x 16 levels, SDXL's batch-2 shapes; short K, ragged N and small grids keep their kernels
    assert plan(gemm(4096, 10240, 1280)) == (256, 256, 1, 7) and plan(gemm(4096, 10240, 1280, geglu=1)) == (256, 256, 1, 7)
    assert plan(gemm(16384, 5120, 640, geglu=1)) == (256, 256, 1, 7) and plan(gemm(2048, 10240, 1280, geglu=1)) == (256, 256, 1, 7)
    assert plan(gemm(4096, 1280, 5120))[3] != 7 and plan(gemm(65536, 2560, 320, geglu=1))[3] == 3
    assert plan(gemm(512, 512, 640, force_tile=24)) == (256, 256, 1, 7)
    bad = gemm(300, 512, 640, force_tile=24)
    assert h.tg_gemm(C.byref(bad), None) != 0 and b"force_tile 24" in h.tg_last_error()
    monkeypatch.setenv("TG_PP", "0")
    assert plan(gemm(4096, 10240, 1280))[:2] == (128, 128) and plan(gemm(16384, 5120, 640, geglu=1)) == (256, 256, 1, 3)
    monkeypatch.setenv("TG_PP", "1")
    assert plan(gemm(4096, 10240, 1280))[3] != 7 and plan(gemm(4096, 10240, 1280, geglu=1))[3] == 7
    monkeypatch.delenv("TG_PP")
    assert plan(gemm(65536, 320, 1280)) == (256, 160, 1, 7)
    monkeypatch.setenv("TG_PP", "7")
    assert plan(gemm(65536, 320, 1280))[:2] == (128, 160)                     # N = 320 = 2.5 tiles of 128: a sixth of the MFMA work would be padding
    monkeypatch.setenv("TG_T160", "3")
    assert plan(gemm(65536, 320, 1280))[:2] == (128, 128) and plan(gemm(16384, 640, 640))[:2] == (128, 160)
    monkeypatch.setenv("TG_T160", "0")
    assert plan(gemm(16384, 640, 2560))[:2] == (128, 128)
    monkeypatch.delenv("TG_T160")
    monkeypatch.delenv("TG_PP")


def test_shift_tensor_ignore_last_dim_matches_reference_formula():
    """utils/utils.py:143-178 with ignore_last_dim=True (attention maps [heads, H, W, tokens]): restated index arithmetic"""
    import torch
    from theatergen_amd.utils import shift_tensor
    g = torch.Generator().manual_seed(3)
    t = torch.randn(2, 16, 24, 5, generator=g)
    for (dx, dy) in ((3, -2), (-5, 4), (0, 0), (24, 1)):
        got = shift_tensor(t, dx, dy, ignore_last_dim=True)
        ref = torch.zeros_like(t)
        ow, oh = 24 - abs(dx), 16 - abs(dy)
        ys, yd = (0, dy) if dy >= 0 else (-dy, 0)
        xs, xd = (0, dx) if dx >= 0 else (-dx, 0)
        ref[..., yd:yd + oh, xd:xd + ow, :] = t[..., ys:ys + oh, xs:xs + ow, :]
        assert got.shape == t.shape and torch.equal(got, ref)
    # normalised offsets quantised to the 8 x 8 base grid (:150-153)
    got = shift_tensor(t, 0.25, -0.125, base_w=8, base_h=8, offset_normalized=True, ignore_last_dim=True)
    assert torch.equal(got, shift_tensor(t, round(0.25 * 8) * 3, round(-0.125 * 8) * 2, ignore_last_dim=True))


def test_processor_branches_outside_the_hot_path_fail_loudly():
    """Processor features no diffusers SD / SDXL module builds (ip_adapter/attention_processor.py:316-317 spatial_norm, :350-352
    attn_process_fn, added_kv) are refused with an explicit error — never silently ignored, never a CPU fallback.  group_norm,
    norm_cross and attention_mask ARE implemented since round 4 (tests/test_round4_gpu.py); a malformed mask is a RuntimeError."""
    import pytest
    import torch
    from theatergen_amd.attention_processor import Attention, AttnProcessor, IPAttnProcessor
    attn = Attention(query_dim=64, heads=2, dim_head=32)
    x = torch.zeros(1, 8, 64)
    with pytest.raises(NotImplementedError):
        AttnProcessor()(attn, x, attn_process_fn=lambda p: p)
    with pytest.raises(RuntimeError):
        AttnProcessor()(attn, x)                                     # CPU tensor: no fallback
    with pytest.raises(RuntimeError):
        AttnProcessor()(attn, x, attention_mask=torch.zeros(1, 8, 8))     # still a CPU tensor
    cross = Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32)
    with pytest.raises(RuntimeError):                                 # the reference's baddbmm rejects the shapes too (:461-477)
        IPAttnProcessor(64, 32)(cross, x, encoder_hidden_states=torch.zeros(1, 9, 32), attention_mask=torch.zeros(1, 8, 9))
    for kw in (dict(spatial_norm_dim=8), dict(added_kv_proj_dim=16)):
        with pytest.raises(ValueError):
            Attention(query_dim=64, heads=2, dim_head=32, **kw)
    with pytest.raises(ValueError):
        Attention(query_dim=64, heads=2, dim_head=32, cross_attention_norm="batch_norm")
    # the norms the reference constructor builds (:80-110) are built here as well
    a2 = Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32, norm_num_groups=8, cross_attention_norm="layer_norm", bias=True)
    assert isinstance(a2.group_norm, torch.nn.GroupNorm) and isinstance(a2.norm_cross, torch.nn.LayerNorm) and a2.to_q.bias is not None
    a3 = Attention(query_dim=64, cross_attention_dim=64, heads=2, dim_head=32, cross_attention_norm="group_norm")
    assert isinstance(a3.norm_cross, torch.nn.GroupNorm) and a3.norm_cross.num_groups == 32
    spat = Attention(query_dim=64, heads=2, dim_head=32)
    spat.spatial_norm = object()
    with pytest.raises(NotImplementedError):
        AttnProcessor()(spat, x)
    # round 5 (VERDICT r4 "missing" 2): one refusal per branch and per processor — the same three on IPAttnProcessor and on a FOREIGN module that carries
    # the attribute (a diffusers Attention built with added_kv_proj_dim keeps it as an attribute; reference :316-317, :350-354)
    enc = torch.zeros(1, 9, 32)
    with pytest.raises(NotImplementedError):
        IPAttnProcessor(64, 32)(cross, x, encoder_hidden_states=enc, attn_process_fn=lambda p: p)
    cross_sp = Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32)
    cross_sp.spatial_norm = object()
    with pytest.raises(NotImplementedError):
        IPAttnProcessor(64, 32)(cross_sp, x, encoder_hidden_states=enc)
    for proc, kw in ((AttnProcessor(), {}), (IPAttnProcessor(64, 32), dict(encoder_hidden_states=enc))):
        akv = Attention(query_dim=64, cross_attention_dim=32 if kw else None, heads=2, dim_head=32)
        akv.added_kv_proj_dim = 16
        with pytest.raises(NotImplementedError):
            proc(akv, x, **kw)


def _fake_ops():
    """shape-only stand-ins for theatergen_amd.ops (no compute): lets the processors' HOST logic run on CPU tensors"""
    import types
    import torch

    def gemm(a0, w, M, N, K, **kw):
        return kw["out"] if kw.get("out") is not None else torch.empty((M, N), dtype=a0.dtype)

    def attn_probs(q, q_ld, q_bs, k, k_ld, k_bs, batch, b0, heads, head_dim, n_q, length, scale, tokens=None):
        return torch.zeros((batch - b0, heads, n_q, length if tokens is None else tokens.numel()))

    return types.SimpleNamespace(
        gemm=gemm, attn_probs=attn_probs,
        linear=lambda x, w, bias=None, **kw: torch.empty((x.shape[0], w.shape[0]), dtype=x.dtype),
        attention=lambda *a, **kw: a[15],
        groupnorm=lambda x0, *a, **kw: torch.empty_like(x0),
        layernorm=lambda x, *a, **kw: torch.empty_like(x),
        transpose=lambda src, batch, rows, cols, out=None: torch.empty((batch, cols, rows), dtype=src.dtype))


class _AttrRecorder:
    """forwards attribute reads to a REFERENCE Attention instance and records every name that instance does not have"""

    def __init__(self, target):
        object.__setattr__(self, "_target", target)
        object.__setattr__(self, "seen", set())
        object.__setattr__(self, "missing", set())

    def __getattr__(self, name):
        self.seen.add(name)
        if not hasattr(self._target, name):
            self.missing.add(name)
            raise AttributeError(name)
        return getattr(self._target, name)


def _reference_attention_module():
    """ip_adapter/attention_processor.py of the reference, imported by path with a stub `diffusers.utils` (build container only)"""
    import importlib.util
    import logging
    import sys
    import types
    ref = os.environ.get("TG_REFERENCE", "/root/reference")
    path = os.path.join(ref, "ip_adapter", "attention_processor.py")
    if not os.path.exists(path):
        import pytest
        pytest.skip("the reference checkout is only present in the build container")
    saved = {k: sys.modules.get(k) for k in ("diffusers", "diffusers.utils")}
    du = types.ModuleType("diffusers.utils")
    du.deprecate = lambda *a, **k: None
    du.logging = types.SimpleNamespace(get_logger=logging.getLogger)
    d = types.ModuleType("diffusers")
    d.utils = du
    d.__path__ = []
    sys.modules.update({"diffusers": d, "diffusers.utils": du})
    try:
        spec = importlib.util.spec_from_file_location("ref_ip_attnproc_boundary", path)
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return m


def test_processors_touch_only_attributes_of_the_reference_attention(monkeypatch):
    """SURVEY §8(b) / INTEGRATION.md level 2: our processors installed on the REFERENCE `Attention` (ip_adapter/attention_processor.py
    :12-279).  Every attribute they read must exist on that class — `inner_dim` / `dim_head` / packed-weight helpers do not."""
    import torch
    from theatergen_amd import attention_processor as AP
    A = _reference_attention_module()
    monkeypatch.setattr(AP, "ops", _fake_ops())
    monkeypatch.setattr(AP, "_need_gpu", lambda t: None)
    x = torch.zeros(2, 16, 64)
    enc = torch.zeros(2, 77 + 4, 32)
    cases = [
        (A.Attention(query_dim=64, heads=2, dim_head=32), AP.AttnProcessor(), dict()),
        (A.Attention(query_dim=64, heads=2, dim_head=32, norm_num_groups=8, residual_connection=True, bias=True), AP.AttnProcessor(),
         dict(attention_mask=torch.zeros(2, 1, 16))),
        (A.Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32, cross_attention_norm="layer_norm"), AP.AttnProcessor(),
         dict(encoder_hidden_states=enc, save_attn_to_dict={}, attn_key=["mid", 0, 0, 0], return_cond_ca_only=True, return_token_ca_only=3)),
        (A.Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32), AP.IPAttnProcessor(64, 32, scale=0.4, num_tokens=4),
         dict(encoder_hidden_states=enc, save_attn_to_dict={}, attn_key=["mid", 0, 0, 0])),
        (A.Attention(query_dim=64, cross_attention_dim=32, heads=2, dim_head=32), AP.CNAttnProcessor(num_tokens=4),
         dict(encoder_hidden_states=enc)),
    ]
    for ref_attn, proc, kw in cases:
        rec = _AttrRecorder(ref_attn)
        out = proc(rec, x, **kw)
        out = out[0] if isinstance(out, tuple) else out
        assert out.shape == x.shape
        assert not rec.missing, f"{type(proc).__name__} read attributes the reference Attention lacks: {sorted(rec.missing)}"
        assert {"heads", "to_q", "to_out"} <= rec.seen
    # 4-D input (the VAE mid-block usage) on the reference module, dispatched through ITS set_processor / forward
    ref_attn = A.Attention(query_dim=64, heads=2, dim_head=32, norm_num_groups=8, residual_connection=True, rescale_output_factor=2.0)
    ref_attn.set_processor(AP.AttnProcessor())
    assert ref_attn(torch.zeros(2, 64, 4, 4)).shape == (2, 64, 4, 4)


def test_ip_scale_is_a_plain_mutable_attribute():
    """IPAdapter.set_scale assigns ``processor.scale`` (ip_adapter/ip_adapter.py:155-158): it must read back as assigned"""
    from theatergen_amd.attention_processor import IPAttnProcessor
    p = IPAttnProcessor(64, 32, scale=0.4, num_tokens=4)
    assert p.scale == 0.4
    p.scale = 0.0
    assert p.scale == 0.0
    p.scale = 1
    assert p.scale == 1 and "scale" not in dict(p.named_parameters())


def test_rc_pack_layout_emulated_mfma():
    """``weights_pack.rc_pack`` / ``rc_pack_tiles`` against a CPU emulation of what ``tg_rc_linear`` does with the stream: MFMA A fragment of
    block (chunk c, tile u, k-step s) x B fragment (the wave's 32 token rows, input channel 64 (s >> 2) + 32 hi + 8 (s & 3) + j), accumulator
    register rho of lane half hi = output channel 64 c + 32 hi + 16 u + rho (32x32 C/D layout: row = (rho & 3) + 8 (rho >> 2) + 4 hi)."""
    import torch
    from theatergen_amd.weights_pack import rc_pack, rc_pack_tiles
    torch.manual_seed(0)
    N, K, M = 128, 320, 32
    W = torch.randn(N, K).bfloat16()
    x = torch.randn(M, K).bfloat16()
    v = torch.randn(N)
    KS = K // 16

    def emulate(get_block, get_v):
        out = torch.zeros(M, N)
        for c in range(N // 64):
            for u in range(2):
                D = torch.zeros(32, M)
                for s in range(KS):
                    frag = get_block(c, u, s)                     # [hi, r, j]
                    A = torch.zeros(32, 16)
                    Bm = torch.zeros(16, M)
                    for hi in range(2):
                        A[:, 8 * hi:8 * hi + 8] = frag[hi]
                        ch = 64 * (s >> 2) + 32 * hi + 8 * (s & 3)
                        Bm[8 * hi:8 * hi + 8] = x[:, ch:ch + 8].float().T
                    D += A @ Bm
                for hi in range(2):
                    for rho in range(16):
                        r = (rho & 3) + 8 * (rho >> 2) + 4 * hi
                        out[:, 64 * c + 32 * hi + 16 * u + rho] = D[r] + get_v(c, u, hi, rho)
        return out
    ref = x.float() @ W.float().T + v
    pk = rc_pack(W, v)
    CB = 128 * K + 1024
    assert pk.numel() == (N // 64) * CB
    frag = lambda c, u, s: pk[c * CB:c * CB + 128 * K].view(torch.bfloat16).float().reshape(2, KS, 2, 32, 8)[u, s]
    vec = lambda c, u, hi, rho: pk[c * CB + 128 * K:(c + 1) * CB].view(torch.float32)[32 * hi + 16 * u + rho]
    assert (emulate(frag, vec) - ref).abs().max() < 1e-4
    pt = rc_pack_tiles(W, v)
    TB = 64 * K + 1024
    assert pt.numel() == (N // 32) * TB + 3072
    fragt = lambda c, u, s: pt[(2 * c + u) * TB:(2 * c + u) * TB + 64 * K].view(torch.bfloat16).float().reshape(KS, 2, 32, 8)[s]
    vect = lambda c, u, hi, rho: pt[(2 * c + u) * TB + 64 * K:(2 * c + u + 1) * TB].view(torch.float32)[16 * hi + rho]
    assert (emulate(fragt, vect) - ref).abs().max() < 1e-4


def test_rc_xattn_slot_maps_are_permutations():
    from theatergen_amd import rowchain
    q, o = rowchain._q_slot_channels(), rowchain._o_slot_channels()
    assert sorted(q.tolist()) == list(range(320)) and sorted(o.tolist()) == list(range(320))
    # a head pair owns exactly five q k-steps (80 slots) and five O k-steps
    for m in range(4):
        for perm in (q, o):
            slots = [n for n in range(320) if perm[n] // 80 == m]
            ksteps = sorted({4 * (n // 64) + ((n // 8) & 3) for n in slots})
            assert ksteps == list(range(5 * m, 5 * m + 5))


def test_pack_ff_streams_emulated():
    """``rowchain.pack_ff`` against a CPU emulation of ``tg_rc_ff``'s index maps (csrc/tg_rowchain.hip): per 32-channel hidden slice the value /
    gate tiles over the normalised rows (K order of ``rc_pack``), accumulator register rho of lane half hi = hidden channel 32 j + 16 hi + rho,
    GEGLU, then net.2's block (tile t, k-step kk) contracts hidden channels 32 j + 16 hi + 8 kk + [0, 8) into output channel
    64 c + 32 hi' + 16 u + rho (t = 2 c + u)."""
    import torch
    import torch.nn.functional as F
    from theatergen_amd import rowchain
    torch.manual_seed(1)
    C, inner, M = 320, 64, 32
    w1 = (torch.randn(2 * inner, C) / C ** 0.5).bfloat16()
    b1 = 0.2 * torch.randn(2 * inner)
    w2 = (torch.randn(C, inner) / inner ** 0.5).bfloat16()
    b2 = 0.1 * torch.randn(C)
    gamma, beta = 1 + 0.2 * torch.randn(C), 0.1 * torch.randn(C)
    xn = torch.randn(M, C)                                           # already normalised rows (what the kernel forms in registers)
    s1, s2, bb2 = rowchain.pack_ff(w1, b1, gamma, beta, w2, b2)
    KS, ns = C // 16, inner // 32
    assert s1.numel() == ns * 41 * 1024 + 3072 and s2.numel() == ns * 20 * 1024
    out = bb2[None, :].repeat(M, 1).clone()

    def mfma_rows(frag, Bm):                                         # frag [s][hi][r][j], Bm(s) -> [16, M]; returns D [32, M]
        D = torch.zeros(32, M)
        for s in range(frag.shape[0]):
            A = torch.zeros(32, 16)
            for hi in range(2):
                A[:, 8 * hi:8 * hi + 8] = frag[s, hi]
            D += A @ Bm(s)
        return D

    def b_rows(s):                                                   # the rows' B operand of k-step s
        Bm = torch.zeros(16, M)
        for hi in range(2):
            ch = 64 * (s >> 2) + 32 * hi + 8 * (s & 3)
            Bm[8 * hi:8 * hi + 8] = xn[:, ch:ch + 8].T
        return Bm
    for j in range(ns):
        blk = s1[j * 41 * 1024:(j + 1) * 41 * 1024]
        frag = blk[:40 * 1024].view(torch.bfloat16).float().reshape(2, KS, 2, 32, 8)      # [value / gate][s][hi][r][j]
        page = blk[40 * 1024:].view(torch.float32)
        Da, Dg = mfma_rows(frag[0], b_rows), mfma_rows(frag[1], b_rows)
        hid = torch.zeros(M, 2, 16)                                   # [token][hi][rho]
        for hi in range(2):
            for rho in range(16):
                r = (rho & 3) + 8 * (rho >> 2) + 4 * hi
                a = Da[r] + page[16 * hi + rho]
                g = Dg[r] + page[32 + 16 * hi + rho]
                hid[:, hi, rho] = a * F.gelu(g)
        f2 = s2[j * 20 * 1024:(j + 1) * 20 * 1024].view(torch.bfloat16).float().reshape(10, 2, 2, 32, 8)     # [t][kk][hi][r][j]
        for t in range(10):
            D = torch.zeros(32, M)
            for kk in range(2):
                A = torch.zeros(32, 16)
                Bm = torch.zeros(16, M)
                for hi in range(2):
                    A[:, 8 * hi:8 * hi + 8] = f2[t, kk, hi]
                    Bm[8 * hi:8 * hi + 8] = hid[:, hi, 8 * kk:8 * kk + 8].T
                D += A @ Bm
            c, u = t >> 1, t & 1
            for hi in range(2):
                for rho in range(16):
                    out[:, 64 * c + 32 * hi + 16 * u + rho] += D[(rho & 3) + 8 * (rho >> 2) + 4 * hi]
    w1g = (w1.float() * gamma[None, :]).bfloat16().float()          # rounded once, as the stream holds it
    pr = xn @ w1g.T + (w1.float() @ beta + b1)
    ref = (pr[:, :inner] * F.gelu(pr[:, inner:])) @ w2.float().T + b2
    assert (out - ref).abs().max() < 2e-4


def test_skinny_pack_layout():
    """``weights_pack.skinny_pack``: block (nt, ks), lane (hi, l31), element j = W[32 nt + l31, 16 ks + 8 hi + j] — the A fragment of a 32x32x16 MFMA as one
    contiguous KiB (csrc/tg_skinny.hip reads ``wpk + ((nt * K / 16 + ks) * 64 + lane) * 8``)."""
    import torch
    from theatergen_amd.weights_pack import skinny_pack
    torch.manual_seed(1)
    N, K = 96, 128
    W = torch.randn(N, K).to(torch.bfloat16)
    pk = skinny_pack(W).reshape(N // 32, K // 16, 64, 8)
    for nt, ks, lane in [(0, 0, 0), (2, 7, 63), (1, 3, 37), (2, 0, 31), (0, 5, 32)]:
        hi, l31 = lane >> 5, lane & 31
        assert torch.equal(pk[nt, ks, lane], W[32 * nt + l31, 16 * ks + 8 * hi:16 * ks + 8 * hi + 8])
    # the whole product through the emulated fragment order: sum over ks of A_frag^T B_frag
    x = torch.randn(32, K).to(torch.bfloat16)
    ref = x.float() @ W.float().t()
    got = torch.zeros(32, N)
    for nt in range(N // 32):
        for ks in range(K // 16):
            a = pk[nt, ks].reshape(2, 32, 8).float()                     # [hi, out row, j]
            bfr = x[:, 16 * ks:16 * ks + 16].reshape(32, 2, 8).float()   # [token, hi, j]
            got[:, 32 * nt:32 * nt + 32] += torch.einsum("hrj,thj->tr", a, bfr)
    assert torch.allclose(got, ref, atol=1e-4, rtol=1e-4)


def test_ctypes_structures_match_the_c_header(tmp_path):
    """Every descriptor the ctypes binding mirrors has the C header's size and field offsets (a probe compiled with gcc from include/theatergen_hip.h):
    a field added on one side only would shift every later argument silently."""
    import ctypes, os, shutil, subprocess
    from theatergen_amd import _lib
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = [("tg_gemm_desc", _lib.GemmDesc), ("tg_attn_desc", _lib.AttnDesc), ("tg_attn_bwd_desc", _lib.AttnBwdDesc),
             ("tg_attn_bwd_cross_desc", _lib.AttnBwdCrossDesc), ("tg_rc_linear_desc", _lib.RcLinearDesc), ("tg_rc_xattn_desc", _lib.RcXattnDesc),
             ("tg_xq_attn_desc", _lib.XqAttnDesc), ("tg_rc_ff_desc", _lib.RcFfDesc), ("tg_skinny_seg", _lib.SkinnySeg), ("tg_skinny_desc", _lib.SkinnyDesc),
             ("tg_rc_front_desc", _lib.RcFrontDesc), ("tg_guidance_item", _lib.GuidanceItem), ("tg_guidance_pitem", _lib.GuidancePItem)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{root}/include/theatergen_hip.h"', 'int main(void) {']
    for cname, cls in pairs:
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ['  return 0;', '}']
    src = tmp_path / "probe.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines())
    for cname, cls in pairs:
        assert int(out[cname]) == ctypes.sizeof(cls), (cname, out[cname], ctypes.sizeof(cls))
        for fname, _ in cls._fields_:
            assert int(out[f"{cname}.{fname}"]) == getattr(cls, fname).offset, (cname, fname)


def test_inline_asm_mfma_loops_leave_accumulators_alone():
    """round 6: kernels whose K loop issues MFMAs as inline asm (tied accumulators) are opaque to the compiler's hazard recogniser; the build gate disassembles the
    object and refuses any other instruction inside the loop that names an accumulator register (an allocator copy / spill there reads values the matrix pipe has
    not written yet).  Runs on the objects the build left behind; skipped on a tree that was never built."""
    import os
    from theatergen_amd import build
    objs = [os.path.join(build.OBJ, o) for o in build.ASM_MFMA_KERNELS]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("no build objects in this tree")
    assert build.check_mfma_loops(verbose=False) == []
