"""``CLIPVisionModelWithProjection`` on the native kernels: IP-Adapter's image encoder, run once per
character reference image (reference ``ip_adapter/ip_adapter.py:78-80`` load, ``:147-148`` ``image_embeds``, ``:310-315``
``hidden_states[-2]`` for the Plus adapters).  Same call surface and state-dict names as ``transformers``:

    out = encoder(pixel_values, output_hidden_states=True);  out.image_embeds, out.hidden_states[-2], out.last_hidden_state

Token-major throughout: the patch embedding is one GEMM over the unfolded patches (K = 3*p*p zero-padded to a multiple of
8; the unfold itself is data movement), LayerNorm kernels, one fused Q|K|V^T projection GEMM per layer (bias in the
epilogue), the flash attention kernel (ViT-H: 16 heads x d = 80, 257 tokens), output projection and fc2 with the residual
fused, fc1 with GELU / quick-GELU in the epilogue.  ``EmbeddingCache`` keys the results by object id: the reference's PNG
"database" of character images (``models/pipelines.py:185, 477``) becomes an in-memory embedding cache.
"""
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .unet import _Packed


class CLIPVisionConfig(SimpleNamespace):
    pass


def vit_h14_config():
    """laion/CLIP-ViT-H-14-laion2B-s32B-b79K vision tower (h94/IP-Adapter ``models/image_encoder``)."""
    return CLIPVisionConfig(hidden_size=1280, intermediate_size=5120, num_hidden_layers=32, num_attention_heads=16, image_size=224,
                            patch_size=14, projection_dim=1024, hidden_act="gelu", layer_norm_eps=1e-5)


_ACT = {"gelu": ops.ACT_GELU, "quick_gelu": ops.ACT_QUICK_GELU}


class _Layer(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        D, I = cfg.hidden_size, cfg.intermediate_size
        self.layer_norm1 = nn.LayerNorm(D, eps=cfg.layer_norm_eps)
        att = nn.Module()
        att.q_proj, att.k_proj, att.v_proj, att.out_proj = nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D), nn.Linear(D, D)
        self.self_attn = att
        self.layer_norm2 = nn.LayerNorm(D, eps=cfg.layer_norm_eps)
        mlp = nn.Module()
        mlp.fc1, mlp.fc2 = nn.Linear(D, I), nn.Linear(I, D)
        self.mlp = mlp
        self._p = _Packed()

    def run(self, x, B, L, cfg, causal=False):
        D, H = cfg.hidden_size, cfg.num_attention_heads
        d = D // H
        a = self.self_attn
        ws = [a.q_proj.weight, a.k_proj.weight, a.v_proj.weight, a.q_proj.bias, a.k_proj.bias, a.v_proj.bias]
        w, b = self._p.get("qkv", ws, lambda: (torch.cat([t.detach() for t in ws[:3]], 0).contiguous(),
                                               torch.cat([t.detach() for t in ws[3:]], 0).contiguous()))
        y = ops.layernorm(x, self.layer_norm1.weight, self.layer_norm1.bias, cfg.layer_norm_eps)
        ldt = (L + 7) // 8 * 8
        qk = torch.empty((B * L, 2 * D), dtype=x.dtype, device=x.device)
        vt = torch.zeros((B, D, ldt), dtype=x.dtype, device=x.device)
        ops.gemm(y, w, B * L, 3 * D, D, bias=b, rows_per_batch=L, out=qk, n_split=2 * D, out_t=vt, ldt=ldt)
        o = torch.empty((B * L, D), dtype=x.dtype, device=x.device)
        ops.attention(qk, 2 * D, L * 2 * D, qk[:, D:], 2 * D, L * 2 * D, vt, ldt, D * ldt, L, B, H, d, L, float(d) ** -0.5, o, D, L * D,
                      causal=causal)
        x = ops.linear(o, a.out_proj.weight, a.out_proj.bias, res=x)
        y = ops.layernorm(x, self.layer_norm2.weight, self.layer_norm2.bias, cfg.layer_norm_eps)
        y = ops.linear(y, self.mlp.fc1.weight, self.mlp.fc1.bias, act=_ACT[cfg.hidden_act])
        return ops.linear(y, self.mlp.fc2.weight, self.mlp.fc2.bias, res=x)


class CLIPVisionOutput(SimpleNamespace):
    pass


class CLIPVisionModelWithProjection(nn.Module):
    def __init__(self, config=None):
        super().__init__()
        cfg = config if config is not None else vit_h14_config()
        self.config = cfg
        D = cfg.hidden_size
        n_pos = (cfg.image_size // cfg.patch_size) ** 2 + 1
        vm = nn.Module()
        emb = nn.Module()
        emb.class_embedding = nn.Parameter(torch.zeros(D))
        emb.patch_embedding = nn.Conv2d(3, D, cfg.patch_size, stride=cfg.patch_size, bias=False)
        emb.position_embedding = nn.Embedding(n_pos, D)
        vm.embeddings = emb
        vm.pre_layrnorm = nn.LayerNorm(D, eps=cfg.layer_norm_eps)        # (sic) transformers' attribute name
        enc = nn.Module()
        enc.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])
        vm.encoder = enc
        vm.post_layernorm = nn.LayerNorm(D, eps=cfg.layer_norm_eps)
        self.vision_model = vm
        self.visual_projection = nn.Linear(D, cfg.projection_dim, bias=False)
        self._p = _Packed()
        for p_ in self.parameters():
            p_.requires_grad_(False)

    @property
    def dtype(self):
        return self.visual_projection.weight.dtype

    @property
    def device(self):
        return self.visual_projection.weight.device

    def forward(self, pixel_values, output_hidden_states=False, return_dict=True):
        if not pixel_values.is_cuda:
            raise RuntimeError("theatergen_amd CLIP encoder runs on the GPU only (no CPU fallback)")
        cfg, vm, dt = self.config, self.vision_model, self.dtype
        B, _, H, W = pixel_values.shape
        ps, D = cfg.patch_size, cfg.hidden_size
        gh, gw = H // ps, W // ps
        n = gh * gw
        L = n + 1
        K = 3 * ps * ps
        Kp = (K + 7) // 8 * 8
        wp = self._p.get("patch", [vm.embeddings.patch_embedding.weight],
                         lambda: torch.nn.functional.pad(vm.embeddings.patch_embedding.weight.detach().reshape(D, K), (0, Kp - K)).contiguous())
        # unfold = data movement only: [B, 3, gh, ps, gw, ps] -> [B, gh, gw, 3, ps, ps] -> [B*n, K] (+ zero pad to Kp)
        patches = torch.zeros((B * n, Kp), dtype=dt, device=pixel_values.device)
        patches[:, :K].copy_(pixel_values.to(dt).reshape(B, 3, gh, ps, gw, ps).permute(0, 2, 4, 1, 3, 5).reshape(B * n, K))
        # residual operand of the patch GEMM = position embeddings of the patch tokens (rows 1..n), repeated per image
        pos = vm.embeddings.position_embedding.weight
        x = torch.empty((B, L, D), dtype=dt, device=pixel_values.device)
        x[:, 0].copy_((vm.embeddings.class_embedding + pos[0]).to(dt))          # one row per image: plumbing
        pe = ops.gemm(patches, wp, B * n, D, Kp, res=pos[1:].to(dt).repeat(B, 1).contiguous())
        x[:, 1:].copy_(pe.reshape(B, n, D))
        x = ops.layernorm(x.reshape(B * L, D), vm.pre_layrnorm.weight, vm.pre_layrnorm.bias, cfg.layer_norm_eps)
        hs = [x.reshape(B, L, D)] if output_hidden_states else None
        for layer in vm.encoder.layers:
            x = layer.run(x, B, L, cfg)
            if output_hidden_states:
                hs.append(x.reshape(B, L, D))
        last = x.reshape(B, L, D)
        pooled = ops.layernorm(last[:, 0].contiguous(), vm.post_layernorm.weight, vm.post_layernorm.bias, cfg.layer_norm_eps)
        emb = ops.linear(pooled, self.visual_projection.weight)
        out = CLIPVisionOutput(image_embeds=emb, last_hidden_state=last, hidden_states=tuple(hs) if hs is not None else None)
        return out if return_dict else (emb, last, out.hidden_states)

    __call__ = forward

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", dtype=torch.bfloat16):
        m = cls(config)
        sd = {k: v for k, v in state_dict.items() if "position_ids" not in k}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if unexpected or missing:
            raise RuntimeError(f"state dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        return m.to(device=device, dtype=dtype)


class CLIPTextConfig(SimpleNamespace):
    pass


def clip_l14_text_config():
    """openai/clip-vit-large-patch14 text tower = the SD-1.5 text encoder (``text_encoder/config.json``)."""
    return CLIPTextConfig(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                          max_position_embeddings=77, hidden_act="quick_gelu", layer_norm_eps=1e-5, eos_token_id=49407)


class CLIPTextOutput(SimpleNamespace):
    def __getitem__(self, i):                       # ``text_encoder(ids)[0]`` as the reference's encode_prompt does
        return (self.last_hidden_state, self.pooler_output)[i]


class CLIPTextModel(nn.Module):
    """``transformers.CLIPTextModel`` on the native kernels: token + position embedding (a gather: data movement), the same
    pre-LN layers as the vision tower with the attention kernel's CAUSAL mask, final_layer_norm; ``[0]`` = last_hidden_state
    (what ``models/models.py:53-79`` / ``encode_prompt`` feed to the UNet), pooler_output = state at the EOS token."""

    def __init__(self, config=None):
        super().__init__()
        cfg = config if config is not None else clip_l14_text_config()
        self.config = cfg
        D = cfg.hidden_size
        tm = nn.Module()
        emb = nn.Module()
        emb.token_embedding = nn.Embedding(cfg.vocab_size, D)
        emb.position_embedding = nn.Embedding(cfg.max_position_embeddings, D)
        tm.embeddings = emb
        enc = nn.Module()
        enc.layers = nn.ModuleList([_Layer(cfg) for _ in range(cfg.num_hidden_layers)])
        tm.encoder = enc
        tm.final_layer_norm = nn.LayerNorm(D, eps=cfg.layer_norm_eps)
        self.text_model = tm
        for p_ in self.parameters():
            p_.requires_grad_(False)

    @property
    def dtype(self):
        return self.text_model.final_layer_norm.weight.dtype

    @property
    def device(self):
        return self.text_model.final_layer_norm.weight.device

    def forward(self, input_ids, return_dict=True):
        if not self.device.type == "cuda":
            raise RuntimeError("theatergen_amd CLIP text encoder runs on the GPU only (no CPU fallback)")
        cfg, tm = self.config, self.text_model
        ids = input_ids.to(self.device)
        B, L = ids.shape
        D = cfg.hidden_size
        tok = tm.embeddings.token_embedding.weight[ids.reshape(-1)]                          # gather = data movement
        pos = tm.embeddings.position_embedding.weight[:L].repeat(B, 1)
        x = ops.add(tok.contiguous(), pos.contiguous())
        for layer in tm.encoder.layers:
            x = layer.run(x, B, L, cfg, causal=True)
        x = ops.layernorm(x, tm.final_layer_norm.weight, tm.final_layer_norm.bias, cfg.layer_norm_eps).reshape(B, L, D)
        eos = (ids == cfg.eos_token_id).int().argmax(dim=-1)
        out = CLIPTextOutput(last_hidden_state=x, pooler_output=x[torch.arange(B, device=x.device), eos])
        return out if return_dict else (out.last_hidden_state, out.pooler_output)

    __call__ = forward

    @classmethod
    def from_state_dict(cls, config, state_dict, device="cuda", dtype=torch.bfloat16):
        m = cls(config)
        sd = {(k if k.startswith("text_model.") else "text_model." + k): v for k, v in state_dict.items() if "position_ids" not in k}
        missing, unexpected = m.load_state_dict(sd, strict=False)
        if unexpected or missing:
            raise RuntimeError(f"state dict mismatch: missing {missing[:5]}... unexpected {unexpected[:5]}...")
        return m.to(device=device, dtype=dtype)


class EmbeddingCache:
    """In-memory replacement of the reference's PNG "database" of character reference images (``models/pipelines.py:185,
    477``): the CLIP features of a character are computed the first time its id shows up and reused on later turns."""

    def __init__(self, encoder, penultimate=False):
        self.encoder, self.penultimate, self._c = encoder, penultimate, {}

    def get(self, obj_id, pixel_values_fn):
        if obj_id not in self._c:
            out = self.encoder(pixel_values_fn(), output_hidden_states=self.penultimate)
            self._c[obj_id] = out.hidden_states[-2] if self.penultimate else out.image_embeds
        return self._c[obj_id]

    def __len__(self):
        return len(self._c)
