"""Oracle (test infrastructure): CPU fp32 restatement of ``transformers.CLIPVisionModelWithProjection.forward`` — the
image encoder IP-Adapter loads (reference ``ip_adapter/ip_adapter.py:78-80``) and queries for ``image_embeds`` (:147-148)
or ``hidden_states[-2]`` (Plus adapters, :310-315).

PINNED against the library itself: ``tests/golden/clip_vision.npz`` holds tiny seeded models (weights, inputs, outputs)
produced by ``tests/golden/make_clip_golden.py`` with the installed ``transformers``; ``tests/test_oracle_golden.py``
checks this file against them.

  embeddings: patch conv (kernel = stride = patch, no bias) -> [B, n, D]; prepend class_embedding; + position_embedding
  pre_layrnorm (sic); per layer: x += out_proj(softmax(q k^T d^-0.5) v) on layer_norm1(x)  (q/k/v/out with bias);
                      x += fc2(act(fc1(layer_norm2(x))))   act = gelu (erf) | quick_gelu = x * sigmoid(1.702 x)
  hidden_states = (embeddings after pre_layrnorm, output of every layer)
  pooled = post_layernorm(last[:, 0]);  image_embeds = visual_projection(pooled)  (no bias)
"""
import torch
import torch.nn.functional as F


def _act(x, kind):
    if kind == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if kind == "gelu":
        return F.gelu(x)
    raise ValueError(kind)


def clip_vision_forward(cfg, sd, pixel_values, p="vision_model"):
    """cfg: dict(hidden_size, num_attention_heads, patch_size, hidden_act, layer_norm_eps).  -> dict(image_embeds,
    last_hidden_state, hidden_states)"""
    D, H, eps, act = cfg["hidden_size"], cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-5), cfg.get("hidden_act", "gelu")
    B = pixel_values.shape[0]
    x = F.conv2d(pixel_values, sd[f"{p}.embeddings.patch_embedding.weight"], stride=cfg["patch_size"])
    x = x.flatten(2).transpose(1, 2)
    cls = sd[f"{p}.embeddings.class_embedding"].reshape(1, 1, D).expand(B, 1, D)
    x = torch.cat([cls, x], dim=1) + sd[f"{p}.embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (D,), sd[f"{p}.pre_layrnorm.weight"], sd[f"{p}.pre_layrnorm.bias"], eps)
    hs = [x]
    i = 0
    while f"{p}.encoder.layers.{i}.layer_norm1.weight" in sd:
        q = f"{p}.encoder.layers.{i}"
        y = F.layer_norm(x, (D,), sd[q + ".layer_norm1.weight"], sd[q + ".layer_norm1.bias"], eps)
        L, d = y.shape[1], D // H
        qh = F.linear(y, sd[q + ".self_attn.q_proj.weight"], sd[q + ".self_attn.q_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        kh = F.linear(y, sd[q + ".self_attn.k_proj.weight"], sd[q + ".self_attn.k_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        vh = F.linear(y, sd[q + ".self_attn.v_proj.weight"], sd[q + ".self_attn.v_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        a = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1) @ vh
        a = a.transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(a, sd[q + ".self_attn.out_proj.weight"], sd[q + ".self_attn.out_proj.bias"])
        y = F.layer_norm(x, (D,), sd[q + ".layer_norm2.weight"], sd[q + ".layer_norm2.bias"], eps)
        y = _act(F.linear(y, sd[q + ".mlp.fc1.weight"], sd[q + ".mlp.fc1.bias"]), act)
        x = x + F.linear(y, sd[q + ".mlp.fc2.weight"], sd[q + ".mlp.fc2.bias"])
        hs.append(x)
        i += 1
    pooled = F.layer_norm(x[:, 0], (D,), sd[f"{p}.post_layernorm.weight"], sd[f"{p}.post_layernorm.bias"], eps)
    emb = F.linear(pooled, sd["visual_projection.weight"])
    return {"image_embeds": emb, "last_hidden_state": x, "hidden_states": hs}


def clip_text_forward(cfg, sd, input_ids, p="text_model"):
    """``transformers.CLIPTextModel`` (the SD text encoder, reference ``models/models.py:53-79`` / ``encode_prompt``): token +
    position embeddings, the same pre-LN layers with a CAUSAL mask, final_layer_norm; pooled = state at the EOS token
    (first occurrence of ``eos_token_id``).  cfg: dict(hidden_size, num_attention_heads, hidden_act, eos_token_id).
    PINNED by ``tests/golden/clip_text.npz`` (captured from the installed library)."""
    D, H, eps, act = cfg["hidden_size"], cfg["num_attention_heads"], cfg.get("layer_norm_eps", 1e-5), cfg.get("hidden_act", "quick_gelu")
    B, L = input_ids.shape
    x = sd[f"{p}.embeddings.token_embedding.weight"][input_ids] + sd[f"{p}.embeddings.position_embedding.weight"][:L][None]
    mask = torch.full((L, L), float("-inf")).triu(1)
    i = 0
    while f"{p}.encoder.layers.{i}.layer_norm1.weight" in sd:
        q = f"{p}.encoder.layers.{i}"
        y = F.layer_norm(x, (D,), sd[q + ".layer_norm1.weight"], sd[q + ".layer_norm1.bias"], eps)
        d = D // H
        qh = F.linear(y, sd[q + ".self_attn.q_proj.weight"], sd[q + ".self_attn.q_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        kh = F.linear(y, sd[q + ".self_attn.k_proj.weight"], sd[q + ".self_attn.k_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        vh = F.linear(y, sd[q + ".self_attn.v_proj.weight"], sd[q + ".self_attn.v_proj.bias"]).reshape(B, L, H, d).transpose(1, 2)
        a = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5 + mask, dim=-1) @ vh
        a = a.transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(a, sd[q + ".self_attn.out_proj.weight"], sd[q + ".self_attn.out_proj.bias"])
        y = F.layer_norm(x, (D,), sd[q + ".layer_norm2.weight"], sd[q + ".layer_norm2.bias"], eps)
        y = _act(F.linear(y, sd[q + ".mlp.fc1.weight"], sd[q + ".mlp.fc1.bias"]), act)
        x = x + F.linear(y, sd[q + ".mlp.fc2.weight"], sd[q + ".mlp.fc2.bias"])
        i += 1
    x = F.layer_norm(x, (D,), sd[f"{p}.final_layer_norm.weight"], sd[f"{p}.final_layer_norm.bias"], eps)
    eos = (input_ids == cfg["eos_token_id"]).int().argmax(dim=-1)
    return {"last_hidden_state": x, "pooler_output": x[torch.arange(B), eos]}
