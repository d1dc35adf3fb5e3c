#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by running the IMPORTED REFERENCE on CPU (fp32).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The reference never travels to the GPU box; only the .npz outputs of this script are committed.
Import recipe: SURVEY.md Appendix A.1 (stub ``diffusers.utils``; load hot-path files by path; CPU shim
for the reference's hard-coded ``.cuda()`` / ``device="cuda"``).
"""
import importlib
import importlib.util
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from tests.golden import gen_common as gc  # noqa: E402

REF = os.environ.get("TG_REFERENCE", "/root/reference")


def load_reference():
    du = types.ModuleType("diffusers.utils")
    du.deprecate = lambda *a, **k: None
    du.logging = types.SimpleNamespace(get_logger=logging.getLogger)
    dtu = types.ModuleType("diffusers.utils.torch_utils")
    dtu.maybe_allow_in_graph = lambda c: c
    du.torch_utils = dtu
    du.maybe_allow_in_graph = lambda c: c
    dm = types.ModuleType("diffusers.models")
    dme = types.ModuleType("diffusers.models.embeddings")
    dme.CombinedTimestepLabelEmbeddings = object
    dm.embeddings = dme
    d = types.ModuleType("diffusers")
    d.utils = du
    d.models = dm
    d.__path__ = []
    sys.modules.update({"diffusers": d, "diffusers.utils": du, "diffusers.utils.torch_utils": dtu,
                        "diffusers.models": dm, "diffusers.models.embeddings": dme})

    def by_path(name, rel):
        spec = importlib.util.spec_from_file_location(name, f"{REF}/{rel}")
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    ref = types.SimpleNamespace()
    ref.resampler = by_path("ref_resampler", "ip_adapter/resampler.py")
    ref.attnproc = by_path("ref_ip_attnproc", "ip_adapter/attention_processor.py")
    # ImageProjModel / MLPProjModel live in ip_adapter/ip_adapter.py which imports diffusers pipelines;
    # their forward is 3 lines (ip_adapter.py:41-47, :62-64) -> built from torch modules below.
    # utils.*: CPU shim for hard-coded CUDA
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.Tensor.is_cuda = property(lambda self: True)
    _z = torch.zeros
    torch.zeros = lambda *a, **k: _z(*a, **({**k, "device": "cpu"} if k.get("device") == "cuda" else k))
    _t = torch.tensor
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    import utils as rutils  # noqa
    rutils.utils.torch_device = "cpu"
    from utils import guidance as rguid, schedule as rsched  # noqa
    import utils.latents as rlat  # noqa
    rlat.torch_device = "cpu"
    ref.utils, ref.guidance, ref.schedule, ref.latents = rutils, rguid, rsched, rlat
    pkg = types.ModuleType("refmodels")
    pkg.__path__ = [f"{REF}/models"]
    sys.modules["refmodels"] = pkg
    ref.attention = importlib.import_module("refmodels.attention")
    return ref


def gen_attention(ref, out):
    A = ref.attnproc
    for ci, (name, C, heads, ctx, N, T) in enumerate(gc.ATTN_CASES):
        w = gc.attn_weights(C, ctx, seed=100 + ci)
        x, enc = gc.attn_inputs(C, ctx, N, T, seed=200 + ci)
        attn = A.Attention(query_dim=C, cross_attention_dim=ctx, heads=heads, dim_head=C // heads)
        attn.load_state_dict({k: v for k, v in w.items() if "_ip" not in k})
        # self-attention uses a query_dim x query_dim K/V: separate module
        ws = gc.attn_weights(C, C, seed=300 + ci, with_ip=False)
        sattn = A.Attention(query_dim=C, cross_attention_dim=None, heads=heads, dim_head=C // heads)
        sattn.load_state_dict(ws)
        with torch.no_grad():
            out[f"{name}.self"] = A.AttnProcessor()(sattn, x).numpy()
            for s in gc.case_scales(ci):
                proc = A.IPAttnProcessor(hidden_size=C, cross_attention_dim=ctx, scale=s, num_tokens=T)
                proc.load_state_dict({"to_k_ip.weight": w["to_k_ip.weight"], "to_v_ip.weight": w["to_v_ip.weight"]})
                out[f"{name}.ip.scale{s}"] = proc(attn, x, encoder_hidden_states=enc).numpy()
            # ControlNet text-only slice
            cn = A.CNAttnProcessor(num_tokens=T)
            out[f"{name}.cn"] = cn(attn, x, encoder_hidden_states=enc).numpy()
            # attention-map capture: save_attn_to_dict + cond-only + token selection (int / index tensor)
            proc = A.IPAttnProcessor(hidden_size=C, cross_attention_dim=ctx, scale=0.4, num_tokens=T)
            proc.load_state_dict({"to_k_ip.weight": w["to_k_ip.weight"], "to_v_ip.weight": w["to_v_ip.weight"]})
            d1, d2, d3 = {}, {}, {}
            key = ("mid", 0, 0, 0)
            proc(attn, x, encoder_hidden_states=enc, attn_key=list(key), save_attn_to_dict=d1, save_keys=[key],
                 return_cond_ca_only=True, return_token_ca_only=5)
            proc(attn, x, encoder_hidden_states=enc, attn_key=list(key), save_attn_to_dict=d2,
                 return_cond_ca_only=True, return_token_ca_only=torch.tensor([1, 3, 7]))
            proc(attn, x, encoder_hidden_states=enc, attn_key=list(key), save_attn_to_dict=d3, save_keys=[("up", 1, 0, 0)])
            assert len(d3) == 0
            if N * heads <= 2048:
                out[f"{name}.cap.int5"] = d1[key].numpy()
                out[f"{name}.cap.idx137"] = d2[key].numpy()
            # 4-D input path + residual connection + rescale
            if ci == 1:
                h = int(N ** 0.5)
                x4 = x.transpose(1, 2).reshape(2, C, h, h).contiguous()
                attn.residual_connection = True
                attn.rescale_output_factor = 2.0
                out[f"{name}.ip.4d"] = proc(attn, x4, encoder_hidden_states=enc).numpy()
                attn.residual_connection = False
                attn.rescale_output_factor = 1.0
        print("attention", name, "done")


def gen_attention_branches(ref, out):
    """AttnProcessor's pre-projection branches on the reference's own Attention / AttnProcessor (:316-347): group_norm (+ 4-D
    input, residual, rescale), q/k/v bias, norm_cross (LayerNorm / GroupNorm), attention_mask as an additive bias."""
    A = ref.attnproc
    for ci, (name, kw, cross, mkind, four_d) in enumerate(gc.BRANCH_CASES):
        sd, x, enc = gc.branch_params(name, kw, cross, seed=700 + ci)
        attn = A.Attention(query_dim=gc.BRANCH_C, cross_attention_dim=gc.BRANCH_CTX if cross else None, heads=gc.BRANCH_HEADS,
                           dim_head=gc.BRANCH_C // gc.BRANCH_HEADS, **kw)
        attn.load_state_dict(sd)
        mask = gc.branch_mask(mkind, cross, seed=800 + ci)
        xin = x
        if four_d:
            h = int(gc.BRANCH_N ** 0.5)
            xin = x.transpose(1, 2).reshape(2, gc.BRANCH_C, h, h).contiguous()
        with torch.no_grad():
            out[f"{name}.out"] = A.AttnProcessor()(attn, xin, encoder_hidden_states=enc, attention_mask=mask).numpy()
        print("attention branch", name, "done")


def gen_resampler(ref, out):
    from theatergen_amd import weights as W
    for ci, (name, case) in enumerate(gc.RESAMPLER_CASES.items()):
        kw = {k: v for k, v in case.items() if k != "seq"}
        sd = W.random_resampler_state_dict(seed=400 + ci, **kw)
        m = ref.resampler.Resampler(**kw)
        m.load_state_dict(sd)
        x = gc.resampler_input(case, seed=500 + ci)
        with torch.no_grad():
            out[f"{name}.out"] = m(x).numpy()
            # zero-image path used for the uncond tokens (ip_adapter.py:311-316 feeds CLIP(zeros); here the
            # projection of an all-zero feature map exercises the LN-of-constant corner)
            out[f"{name}.zero"] = m(torch.zeros_like(x)).numpy()
        print("resampler", name, "done")
    # ImageProjModel (ip_adapter.py:30-47) built from torch modules with the identical 3-line forward
    sd, e = gc.imageproj_params()
    proj = torch.nn.Linear(1024, 4 * 768)
    norm = torch.nn.LayerNorm(768)
    proj.load_state_dict({"weight": sd["proj.weight"], "bias": sd["proj.bias"]})
    norm.load_state_dict({"weight": sd["norm.weight"], "bias": sd["norm.bias"]})
    with torch.no_grad():
        out["imageproj.out"] = norm(proj(e).reshape(-1, 4, 768)).numpy()
        out["imageproj.zero"] = norm(proj(torch.zeros_like(e)).reshape(-1, 4, 768)).numpy()


def load_proj_classes():
    """The REAL ``ImageProjModel`` / ``MLPProjModel`` classes of reference ip_adapter/ip_adapter.py:30-64.  The module itself
    cannot be imported (it pulls diffusers pipelines at import time), so the two class definitions are cut out of its AST
    and executed as they stand (build container only; nothing of the source is stored)."""
    import ast
    path = f"{REF}/ip_adapter/ip_adapter.py"
    tree = ast.parse(open(path).read(), filename=path)
    wanted = [n for n in tree.body if isinstance(n, ast.ClassDef) and n.name in ("ImageProjModel", "MLPProjModel")]
    assert len(wanted) == 2
    ns = {"torch": torch}
    exec(compile(ast.Module(body=wanted, type_ignores=[]), path, "exec"), ns)
    return ns["ImageProjModel"], ns["MLPProjModel"]


def gen_imageproj(ref, out):
    ImageProjModel, MLPProjModel = load_proj_classes()
    sd, e = gc.imageproj_params()
    m = ImageProjModel(cross_attention_dim=768, clip_embeddings_dim=1024, clip_extra_context_tokens=4)
    m.load_state_dict(sd)
    with torch.no_grad():
        out["imageproj.out"] = m(e).numpy()
        out["imageproj.zero"] = m(torch.zeros_like(e)).numpy()
    sd2, e2 = gc.mlpproj_params()
    m2 = MLPProjModel(cross_attention_dim=768, clip_embeddings_dim=1280)
    m2.load_state_dict(sd2)
    with torch.no_grad():
        out["mlpproj.out"] = m2(e2).numpy()
        out["mlpproj.zero"] = m2(torch.zeros_like(e2)).numpy()


def gen_latents_half(ref, out):
    """The latent recipe with the reference's REAL adapter dtype (generate.py:77-81: fp16; bf16 for this build's bench):
    utils/latents.py draws `torch.randn(..., dtype=unet.dtype)` on the CPU generator and blends half-precision tensors."""
    L = ref.latents
    boxes = [[40 / 512, 150 / 512, 230 / 512, 450 / 512], [280 / 512, 150 / 512, 470 / 512, 450 / 512]]
    for name, dt in (("fp16", torch.float16), ("bf16", torch.bfloat16)):
        class _Cfg: in_channels = 4
        class _Unet: config = _Cfg(); dtype = dt
        class _Sched: init_noise_sigma = 1.0
        class _Pipe: unet = _Unet(); scheduler = _Sched()
        class _Adapter: pipe = _Pipe()
        ad = _Adapter()
        lst, bg, seeds = L.get_input_latents_list(None, bg_seed=0, fg_seed_start=123456789, fg_blending_ratio=0.01,
                                                  height=512, width=512, adapter=ad, so_boxes=boxes)
        assert lst[0].dtype == dt and bg.dtype == dt
        # stored as fp32 (exact: half-precision values), compared bit for bit
        out[f"{name}.input0"] = lst[0].float().numpy(); out[f"{name}.input1"] = lst[1].float().numpy()
        out[f"{name}.bg"] = bg.float().numpy()
        one = L.get_input_latents_lne(1, ad, None, bg_seed=7, fg_seed_start=7 + 123456789, fg_blending_ratio=0.01,
                                      height=512, width=512, so_boxes=boxes)
        out[f"{name}.lne_seed7_idx1"] = one.float().numpy()


def gen_ff(ref, out):
    att = ref.attention
    g = torch.Generator().manual_seed(700)
    ff = att.FeedForward(64, dropout=0.0, activation_fn="geglu")
    with torch.no_grad():
        for p in ff.parameters():
            p.copy_((torch.rand(p.shape, generator=g) * 2 - 1) * 0.2)
        x = torch.randn(2, 16, 64, generator=g)
        out["ff.x"] = x.numpy()
        sd = ff.state_dict()
        for k, v in sd.items():
            out["ff.sd." + k] = v.numpy().copy()
        out["ff.out"] = ff(x).numpy()


def gen_guidance(ref, out):
    G = ref.guidance
    boxes_sets, positions, keys = gc.GUIDANCE_BOXES, gc.GUIDANCE_POSITIONS, gc.GUIDANCE_KEYS
    import warnings
    for nbox in (1, 2, 4):
        maps, g = gc.guidance_attn_maps(nbox)
        saved = {k: v.clone().requires_grad_(True) for k, v in maps.items()}
        for mode, kw in (("max", dict(use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2, fg_weight=1.0, bg_weight=4.0)),
                         ("ratio", dict(use_ratio_based_loss=True))):
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                loss = G.compute_ca_lossv3(saved_attn=saved, bboxes=boxes_sets[nbox], object_positions=positions[nbox],
                                           guidance_attn_keys=keys, **kw)
            grads = torch.autograd.grad(loss, [saved[k] for k in keys])
            out[f"guid.{nbox}.{mode}.loss"] = loss.detach().numpy()
            if nbox == 2:
                for k, gr in zip(keys, grads):
                    out[f"guid.{nbox}.{mode}.grad.{'_'.join(map(str, k))}"] = gr.numpy().astype(np.float32)
            else:
                out[f"guid.{nbox}.{mode}.gradsum"] = np.array([float(gr.double().abs().sum()) for gr in grads])
        # reference-attention transfer loss (guidance.py:150-242) for the 2-box set
        if nbox == 2:
            ref_attns = gc.guidance_ref_maps(g)
            loss = G.compute_ca_lossv3(saved_attn=saved, bboxes=boxes_sets[2], object_positions=positions[2],
                                       guidance_attn_keys=keys, ref_ca_saved_attns=ref_attns, index=3,
                                       ref_ca_loss_weight=2.0, word_token_indices=[3, 7],
                                       use_ratio_based_loss=False, fg_top_p=0.2, bg_top_p=0.2)
            out["guid.2.withref.loss"] = loss.detach().numpy()
    # phrase indices with the fake tokenizer
    tok = gc.FakeTokenizer()
    prompt = "a photo of a red cat and a small brown dog , park"
    pos, wti = G.get_phrase_indices(tok, prompt, ["a red cat", "a small brown dog"], words=["cat", "dog"],
                                    return_word_token_indices=True)
    out["phrase.pos0"] = np.array(pos[0]); out["phrase.pos1"] = np.array(pos[1]); out["phrase.wti"] = np.array(wti)
    pos2, newp = G.get_phrase_indices(tok, "a street", ["a blue car"], add_suffix_if_not_found=True)
    out["phrase.suffix.pos0"] = np.array(pos2[0])
    out["phrase.suffix.prompt"] = np.array(newp)


def gen_geometry_latents(ref, out):
    U, L, S = ref.utils.utils if hasattr(ref.utils, "utils") else ref.utils, ref.latents, ref.schedule
    U = ref.utils.utils
    boxes = [[40 / 512, 150 / 512, 230 / 512, 450 / 512], [280 / 512, 150 / 512, 470 / 512, 450 / 512],
             [0.013, 0.49, 0.377, 0.999], [-0.1, 0.2, 0.3, 1.2], [0.5, 0.5, 0.5, 0.5], [0.03125, 0.046875, 0.546875, 0.578125]]
    sp = []
    for b in boxes:
        for (H, W) in ((64, 64), (16, 16), (8, 8), (96, 96)):
            sp.append(list(U.scale_proportion(b, H, W)) + list(U.scale_proportion(b, H, W, use_legacy=True)))
    out["geo.boxes"] = np.array(boxes)
    out["geo.scale_proportion"] = np.array(sp)
    out["geo.mask64"] = np.stack([U.proportion_to_mask(b, 64, 64).numpy() for b in boxes])
    out["geo.centered"] = np.array([U.get_centered_box(b) for b in boxes[:3]]
                                   + [U.get_centered_box(b, horizontal_center_only=False) for b in boxes[:3]]
                                   + [U.get_centered_box(b, horizontal_center_only=False, vertical_placement="floor_padding", floor_padding=0.05) for b in boxes[:3]])
    g = torch.Generator().manual_seed(900)
    masks = []
    for i in range(3):
        m = torch.zeros(64, 64, dtype=torch.bool)
        y0, x0 = 5 + 11 * i, 8 + 9 * i
        m[y0:y0 + 20 + 3 * i, x0:x0 + 14 + 5 * i] = True
        m &= torch.rand(64, 64, generator=g) > 0.15
        masks.append(m)
    out["geo.masks"] = torch.stack(masks).numpy()
    out["geo.mask_box"] = np.array([[int(v) for v in U.binary_mask_to_box(m)] for m in masks])
    out["geo.mask_box_mask"] = torch.stack([U.binary_mask_to_box_mask(m, to_device=False) for m in masks]).numpy()
    out["geo.mask_center"] = np.array([U.binary_mask_to_center(m, normalize=True) for m in masks])
    t = torch.randn(3, 1, 4, 64, 64, generator=torch.Generator().manual_seed(901))
    shifts = [(0.13, -0.21), (-0.5, 0.0), (0.07, 0.06), (0.99, 0.3)]
    out["geo.shifts"] = np.array(shifts)
    out["geo.shift_out"] = np.stack([U.shift_tensor(t, xo, yo, offset_normalized=True).numpy() for xo, yo in shifts])
    ts = torch.arange(981, 0, -20)
    out["sched.fast_10_2"] = S.get_fast_schedule(ts, 10, 2).numpy()
    out["sched.fast_49_2"] = S.get_fast_schedule(ts, 49, 2).numpy()

    # latents: reference signature needs an adapter stand-in (utils/latents.py:261-264)
    class _Cfg: in_channels = 4
    class _Unet: config = _Cfg(); dtype = torch.float32
    class _Sched: init_noise_sigma = 1.0
    class _Pipe: unet = _Unet(); scheduler = _Sched()
    class _Adapter: pipe = _Pipe()
    ad = _Adapter()
    so_boxes = boxes[:2]
    lst, bg, seeds = L.get_input_latents_list(None, bg_seed=0, fg_seed_start=123456789, fg_blending_ratio=0.01,
                                              height=512, width=512, adapter=ad, so_boxes=so_boxes)
    out["lat.input0"] = lst[0].numpy(); out["lat.input1"] = lst[1].numpy(); out["lat.bg"] = bg.numpy()
    out["lat.seeds"] = np.array(seeds)
    one = L.get_input_latents_lne(1, ad, None, bg_seed=7, fg_seed_start=7 + 123456789, fg_blending_ratio=0.01,
                                  height=512, width=512, so_boxes=so_boxes)
    out["lat.lne_seed7_idx1"] = one.numpy()
    # align + compose on synthetic per-step latents [51,1,4,64,64]
    lat_all = [torch.randn(51, 1, 4, 64, 64, generator=g) for _ in range(3)]
    out["lat.all_seed"] = np.array(900)
    new_l, new_m, offs = L.align_with_bboxes(lat_all, masks, bboxes=boxes[:3])
    out["lat.align_offsets"] = np.array(offs)
    out["lat.align_masks"] = torch.stack(new_m).numpy()
    out["lat.align_l_checksum"] = np.array([float(x.double().sum()) for x in new_l] + [float(x.double().abs().sum()) for x in new_l])
    comp, fgidx = L.compose_latents(ad, None, new_l, new_m, 50, 1, 512, 512, latents_bg=bg)
    out["lat.compose_fgidx"] = fgidx.numpy()
    out["lat.compose_step0"] = comp[0].numpy()
    out["lat.compose_step37"] = comp[37].numpy()
    out["lat.compose_checksum"] = np.array([float(comp.double().sum()), float(comp.double().abs().sum())])


def gen_mid_image(ref, out):
    """``prepare_mid_image`` and ``compose_latents_with_alignment`` of the imported reference (utils/latents.py:48-135, 242-255) on the synthetic
    characters of ``gen_common.mid_image_case``.  The reference saves two PNGs under ./visualization: run in a scratch directory."""
    import tempfile
    from PIL import Image
    L = ref.latents
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as tmp:
        os.makedirs(os.path.join(tmp, "visualization"))
        os.chdir(tmp)
        try:
            for case in (0, 1, 2):
                masks, images, boxes = gc.mid_image_case(case)
                mask_img, canvas = L.prepare_mid_image("1.5", 0, masks, [Image.fromarray(i) for i in images], boxes)
                out[f"mid{case}.mask"] = np.array(mask_img)
                out[f"mid{case}.image"] = np.array(canvas)
            # the whole hand-off: align the 51-step latents / 64 x 64 masks to the boxes, paste, compose
            masks, images, boxes = gc.mid_image_case(0)
            g = torch.Generator().manual_seed(910)
            masks64 = [m.view(64, 8, 64, 8).any(3).any(1) for m in masks]
            lat_all = [torch.randn(51, 1, 4, 64, 64, generator=g) for _ in masks]
            bg = torch.randn(1, 4, 64, 64, generator=g)

            class _Cfg: in_channels = 4
            class _Unet: config = _Cfg(); dtype = torch.float32
            class _Sched: init_noise_sigma = 1.0
            class _Pipe: unet = _Unet(); scheduler = _Sched()
            class _Adapter: pipe = _Pipe()
            comp, fgidx, inp_mask, inp_img = L.compose_latents_with_alignment(
                "1.5", _Adapter(), 0, masks, [Image.fromarray(i) for i in images], None, lat_all, masks64, 50, 1, 512, 512,
                align_with_overall_bboxes=True, overall_bboxes=[[boxes[0]], [boxes[1]]], horizontal_shift_only=False, latents_bg=bg)
            out["cwa.fgidx"] = fgidx.numpy()
            out["cwa.step0"] = comp[0].numpy()
            out["cwa.step23"] = comp[23].numpy()
            out["cwa.checksum"] = np.array([float(comp.double().sum()), float(comp.double().abs().sum())])
            out["cwa.mask"] = np.array(inp_mask)
            out["cwa.image"] = np.array(inp_img)
        finally:
            os.chdir(cwd)


def gen_block(ref, out):
    """SD-1.5 first-level geometry composed from the reference's parts.  ``BasicTransformerBlock.forward`` / ``Transformer2DModel.forward``
    cannot execute as shipped (models/attention_processor.py:161 hands ``object_positions`` to processors that do not take it: TypeError), so
    the residual structure of models/attention.py:186-236 and models/transformer_2d.py:285-327 is restated HERE, in the generator, around the
    reference's own ``Attention`` / ``AttnProcessor`` / ``IPAttnProcessor`` (ip_adapter/attention_processor.py) and ``FeedForward``
    (models/attention.py:243-292) modules and torch's LayerNorm / GroupNorm / Conv2d (what the reference instantiates)."""
    A, att = ref.attnproc, ref.attention
    C, H, ctx = gc.BLOCK_C, gc.BLOCK_HEADS, gc.BLOCK_CTX
    for T in gc.BLOCK_T:
        sd, x, enc = gc.block_params(T)
        b = "transformer_blocks.0."
        sub = lambda pre: {k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)}
        def ln(name):
            m = torch.nn.LayerNorm(C)
            m.load_state_dict(sub(b + name + "."))
            return m
        n1, n2, n3 = ln("norm1"), ln("norm2"), ln("norm3")
        a1 = A.Attention(query_dim=C, cross_attention_dim=None, heads=H, dim_head=C // H)
        a1.load_state_dict(sub(b + "attn1."))
        a2 = A.Attention(query_dim=C, cross_attention_dim=ctx, heads=H, dim_head=C // H)
        a2.load_state_dict({k: v for k, v in sub(b + "attn2.").items() if not k.startswith("processor.")})
        ff = att.FeedForward(C, dropout=0.0, activation_fn="geglu")
        ff.load_state_dict(sub(b + "ff."))
        gn = torch.nn.GroupNorm(32, C, eps=1e-6)
        gn.load_state_dict(sub("norm."))
        pin, pout = torch.nn.Conv2d(C, C, 1), torch.nn.Conv2d(C, C, 1)
        pin.load_state_dict(sub("proj_in.")); pout.load_state_dict(sub("proj_out."))

        def ipproc(s):
            p = A.IPAttnProcessor(hidden_size=C, cross_attention_dim=ctx, scale=s, num_tokens=T)
            p.load_state_dict(sub(b + "attn2.processor."))
            return p

        with torch.no_grad():
            B, _, hh, ww = x.shape
            tok = x.permute(0, 2, 3, 1).reshape(B, hh * ww, C)                    # transformer_2d.py:289 on the raw input: the sub-block fixtures' stream
            for s in gc.IP_SCALES:
                # models/attention.py:206-224: norm2 -> attn2 -> + hidden_states
                out[f"T{T}.xattn.scale{s}"] = (ipproc(s)(a2, n2(tok), encoder_hidden_states=enc) + tok).numpy()
            # plain AttnProcessor on attn2 (text tokens only: what a UNet without IP-Adapter runs)
            out[f"T{T}.xattn.plain"] = (A.AttnProcessor()(a2, n2(tok), encoder_hidden_states=enc[:, :77]) + tok).numpy()

            def block(h, s):
                h = A.AttnProcessor()(a1, n1(h)) + h                              # :186-204
                h = ipproc(s)(a2, n2(h), encoder_hidden_states=enc) + h           # :206-224
                return ff(n3(h)) + h                                              # :226-236
            out[f"T{T}.block.scale0.4"] = block(tok, 0.4).numpy()
            # transformer_2d.py:285-327 (conv projections: use_linear_projection False, SD-1.5)
            res = x
            h = pin(gn(x))
            h = h.permute(0, 2, 3, 1).reshape(B, hh * ww, C)
            h = block(h, 0.4)
            h = h.reshape(B, hh, ww, C).permute(0, 3, 1, 2).contiguous()
            out[f"T{T}.transformer.scale0.4"] = (pout(h) + res).numpy()
        print("block T", T, "done")


def gen_xq(ref, out):
    """inner-level cross-attention sub-block: the reference's Attention + IPAttnProcessor / AttnProcessor on torch's LayerNorm output
    (models/attention.py:206-224 without the residual add: what the processor returns)"""
    A = ref.attnproc
    for ci, (name, C, heads, ctx, N, T, scale, ip) in enumerate(gc.XQ_CASES):
        w, norm, x, enc = gc.xq_params(ci)
        attn = A.Attention(query_dim=C, cross_attention_dim=ctx, heads=heads, dim_head=C // heads)
        attn.load_state_dict({k: v for k, v in w.items() if "_ip" not in k})
        ln = torch.nn.LayerNorm(C)
        ln.load_state_dict(norm)
        with torch.no_grad():
            if ip:
                proc = A.IPAttnProcessor(hidden_size=C, cross_attention_dim=ctx, scale=scale, num_tokens=T)
                proc.load_state_dict({"to_k_ip.weight": w["to_k_ip.weight"], "to_v_ip.weight": w["to_v_ip.weight"]})
            else:
                proc = A.AttnProcessor()
            out[f"{name}.out"] = proc(attn, ln(x), encoder_hidden_states=enc).numpy()
        print("xq", name, "done")


def main():
    ref = load_reference()
    torch.set_num_threads(8)
    jobs = {"attn": gen_attention, "attn_branches": gen_attention_branches, "resampler": gen_resampler, "ff_geglu": gen_ff, "guidance": gen_guidance,
            "geometry_latents": gen_geometry_latents, "imageproj": gen_imageproj, "latents_half": gen_latents_half, "block": gen_block, "xq": gen_xq, "mid_image": gen_mid_image}
    only = sys.argv[1:]
    for name, fn in jobs.items():
        if only and name not in only:
            continue
        out = {}
        fn(ref, out)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(name, "->", path, f"{os.path.getsize(path) / 1e6:.2f} MB", len(out), "arrays")


if __name__ == "__main__":
    main()
