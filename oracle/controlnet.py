"""Oracle (test infrastructure): CPU fp32 restatement of ``diffusers.ControlNetModel.forward`` as the reference's
stage-2 loop calls it (``models/pipelines.py:770-778``: ``controlnetpipe.controlnet(control_model_input, t,
encoder_hidden_states=controlnet_prompt_embeds, controlnet_cond=imagetight, conditioning_scale=cond_scale,
guess_mode=guess_mode, return_dict=False)``), whose outputs feed the UNet's residual injection points
(``models/unet_2d_condition.py:938-946, 975-976``; ``models/pipelines.py:804-818``).

PARITY UNPINNED: ``ControlNetModel`` / ``ControlNetConditioningEmbedding`` live in the third-party
``diffusers==0.21.4`` (``requirements.txt:13``; source not under /root/reference, package not installed).  Restated
from the documented 0.21.4 semantics:

  ControlNetConditioningEmbedding(conditioning_embedding_channels = block_out_channels[0], conditioning_channels = 3,
      block_out_channels = (16, 32, 96, 256)):
      e = silu(conv_in(cond));  for c_in, c_out in pairs: e = silu(conv3x3(c_in, c_in)(e)); e = silu(conv3x3(c_in, c_out,
      stride 2)(e));  e = conv_out(e)          (conv_out is zero-initialised in a fresh ControlNet)
  ControlNetModel.forward:
      emb = time_embedding(timesteps(t));  x = conv_in(sample) + e
      res = (x,) + outputs of every ResBlock(+Transformer) and downsampler of the down blocks;  x = mid_block(x)
      down_i = controlnet_down_blocks[i](res_i)  (1x1 convs, zero-initialised);  mid = controlnet_mid_block(x)
      guess_mode (and not global_pool_conditions): scales = logspace(-1, 0, len(down) + 1) * conditioning_scale
      else: every output * conditioning_scale;  global_pool_conditions: mean over (h, w), keepdim.

The encoder (conv_in, time embedding, down blocks, mid block) is the UNet's own code path (``oracle/unet.py``); with
IP-Adapter installed its cross-attention runs ``CNAttnProcessor`` (text tokens only, pinned in ``oracle/attention.py``).
State-dict names are diffusers' (``controlnet_cond_embedding.blocks.0.weight``, ``controlnet_down_blocks.3.bias`` ...).
"""
import torch
import torch.nn.functional as F

from .unet import _lin, _tuple, cfg_get, resnet_block, timestep_sinusoid, transformer_2d


def cond_embedding(sd, cond, p="controlnet_cond_embedding"):
    e = F.silu(F.conv2d(cond, sd[f"{p}.conv_in.weight"], sd[f"{p}.conv_in.bias"], padding=1))
    i = 0
    while f"{p}.blocks.{i}.weight" in sd:
        stride = 2 if i % 2 == 1 else 1
        e = F.silu(F.conv2d(e, sd[f"{p}.blocks.{i}.weight"], sd[f"{p}.blocks.{i}.bias"], stride=stride, padding=1))
        i += 1
    return F.conv2d(e, sd[f"{p}.conv_out.weight"], sd[f"{p}.conv_out.bias"], padding=1)


def controlnet_forward(cfg, sd, sample, timestep, encoder_hidden_states, controlnet_cond, conditioning_scale=1.0,
                       guess_mode=False, cross_mode="cn", num_tokens=4, global_pool_conditions=False):
    """-> (list of down residuals [B, C_i, h_i, w_i], mid residual)"""
    boc = tuple(cfg_get(cfg, "block_out_channels"))
    nb = len(boc)
    down_types = tuple(cfg_get(cfg, "down_block_types"))
    lpb = _tuple(cfg_get(cfg, "layers_per_block", 2), nb)
    heads_t = _tuple(cfg_get(cfg, "attention_head_dim", 8), nb)
    tl_t = _tuple(cfg_get(cfg, "transformer_layers_per_block", 1), nb)
    use_linear = bool(cfg_get(cfg, "use_linear_projection", False))
    groups = cfg_get(cfg, "norm_num_groups", 32)
    eps = cfg_get(cfg, "norm_eps", 1e-5)
    ca_kwargs = {}

    B = sample.shape[0]
    t = torch.as_tensor(timestep).reshape(-1).expand(B)
    t_emb = timestep_sinusoid(t, boc[0], cfg_get(cfg, "flip_sin_to_cos", True), cfg_get(cfg, "freq_shift", 0))
    emb = _lin(sd, "time_embedding.linear_2", F.silu(_lin(sd, "time_embedding.linear_1", t_emb)))

    x = F.conv2d(sample, sd["conv_in.weight"], sd["conv_in.bias"], padding=1)
    x = x + cond_embedding(sd, controlnet_cond)
    res = [x]
    for i, bt in enumerate(down_types):
        for j in range(lpb[i]):
            x = resnet_block(sd, f"down_blocks.{i}.resnets.{j}", x, emb, groups, eps)
            if bt == "CrossAttnDownBlock2D":
                ca_kwargs["attn_key"] = ["down", i, j]
                x = transformer_2d(sd, f"down_blocks.{i}.attentions.{j}", x, encoder_hidden_states, heads_t[i],
                                   tl_t[i], use_linear, groups, ca_kwargs, 0.0, num_tokens, cross_mode)
            res.append(x)
        if i != nb - 1:
            x = F.conv2d(x, sd[f"down_blocks.{i}.downsamplers.0.conv.weight"],
                         sd[f"down_blocks.{i}.downsamplers.0.conv.bias"], stride=2, padding=1)
            res.append(x)
    x = resnet_block(sd, "mid_block.resnets.0", x, emb, groups, eps)
    ca_kwargs["attn_key"] = ["mid", 0, 0]
    x = transformer_2d(sd, "mid_block.attentions.0", x, encoder_hidden_states, heads_t[-1], tl_t[-1], use_linear, groups,
                       ca_kwargs, 0.0, num_tokens, cross_mode)
    x = resnet_block(sd, "mid_block.resnets.1", x, emb, groups, eps)

    down = [F.conv2d(r, sd[f"controlnet_down_blocks.{i}.weight"], sd[f"controlnet_down_blocks.{i}.bias"])
            for i, r in enumerate(res)]
    mid = F.conv2d(x, sd["controlnet_mid_block.weight"], sd["controlnet_mid_block.bias"])
    if guess_mode and not global_pool_conditions:
        scales = torch.logspace(-1, 0, len(down) + 1) * conditioning_scale
        down = [d * s for d, s in zip(down, scales)]
        mid = mid * scales[-1]
    else:
        down = [d * conditioning_scale for d in down]
        mid = mid * conditioning_scale
    if global_pool_conditions:
        down = [d.mean(dim=(2, 3), keepdim=True) for d in down]
        mid = mid.mean(dim=(2, 3), keepdim=True)
    return down, mid
