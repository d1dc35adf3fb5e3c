"""Golden vectors for the CLIP vision encoder: the reference loads
``transformers.CLIPVisionModelWithProjection`` (``ip_adapter/ip_adapter.py:78-80``) and uses ``.image_embeds`` (:147-148) or,
for the Plus adapters, ``hidden_states[-2]`` (:310-315).  ``transformers`` is a third-party dependency that IS installed
in the build container, so the oracle (``oracle/clip.py``) is pinned against the library itself: tiny seeded models,
weights + inputs + outputs stored here.

Run in the build container:   python tests/golden/make_clip_golden.py      (writes tests/golden/clip_vision.npz)
"""
import os

import numpy as np
import torch

CASES = {
    # name: (hidden, intermediate, layers, heads, image, patch, projection, act)
    "gelu": (64, 128, 3, 4, 56, 14, 32, "gelu"),                 # OpenCLIP ViT-H/14 style (IP-Adapter's image encoder)
    "quick_gelu": (64, 192, 2, 2, 42, 14, 48, "quick_gelu"),     # OpenAI CLIP style
}


TEXT_CASES = {
    # name: (vocab, hidden, intermediate, layers, heads, max_pos, act, eos)
    "text_quick_gelu": (96, 64, 128, 3, 4, 24, "quick_gelu", 95),     # OpenAI CLIP ViT-L/14 text tower style (SD-1.5)
    "text_gelu": (80, 64, 192, 2, 2, 77, "gelu", 79),                  # OpenCLIP style (SD-2.1), full 77-token context
}


def make_text(out):
    from transformers import CLIPTextConfig, CLIPTextModel
    for name, (vocab, hid, inter, layers, heads, max_pos, act, eos) in TEXT_CASES.items():
        torch.manual_seed(4321 + len(name))
        cfg = CLIPTextConfig(vocab_size=vocab, hidden_size=hid, intermediate_size=inter, num_hidden_layers=layers,
                             num_attention_heads=heads, max_position_embeddings=max_pos, hidden_act=act,
                             eos_token_id=eos, bos_token_id=0, pad_token_id=1)
        m = CLIPTextModel(cfg).eval()
        with torch.no_grad():
            for p in m.parameters():
                p.add_(0.05 * torch.randn_like(p))
            ids = torch.randint(2, eos, (3, max_pos))
            ids[:, 0] = 0
            for b, n in enumerate((max_pos - 1, max_pos // 2, 5)):        # EOS position, padding after it
                ids[b, n] = eos
                ids[b, n + 1:] = 1
            o = m(input_ids=ids)
        for k, v in m.state_dict().items():
            # classic (transformers 4.x, the reference's requirement) parameter names carry the "text_model." prefix
            kk = k if k.startswith("text_model.") else "text_model." + k
            out[f"{name}.w.{kk}"] = v.numpy().astype(np.float32)
        out[f"{name}.ids"] = ids.numpy()
        out[f"{name}.last_hidden_state"] = o.last_hidden_state.numpy()
        out[f"{name}.pooler_output"] = o.pooler_output.numpy()


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    out = {}
    tout = {}
    make_text(tout)
    tpath = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_text.npz")
    np.savez_compressed(tpath, **tout)
    print("wrote", tpath, os.path.getsize(tpath) // 1024, "KiB")
    for name, (hid, inter, layers, heads, img, patch, proj, act) in CASES.items():
        torch.manual_seed(1234 + len(name))
        cfg = CLIPVisionConfig(hidden_size=hid, intermediate_size=inter, num_hidden_layers=layers, num_attention_heads=heads,
                               image_size=img, patch_size=patch, projection_dim=proj, hidden_act=act)
        m = CLIPVisionModelWithProjection(cfg).eval()
        with torch.no_grad():
            for p in m.parameters():                       # HF's init is near-degenerate for norms / biases: jitter everything
                p.add_(0.05 * torch.randn_like(p))
            x = torch.randn(2, 3, img, img)
            o = m(x, output_hidden_states=True)
        for k, v in m.state_dict().items():
            out[f"{name}.w.{k}"] = v.numpy().astype(np.float32)
        out[f"{name}.x"] = x.numpy()
        out[f"{name}.image_embeds"] = o.image_embeds.numpy()
        out[f"{name}.last_hidden_state"] = o.last_hidden_state.numpy()
        out[f"{name}.penultimate"] = o.hidden_states[-2].numpy()
        out[f"{name}.n_hidden_states"] = np.array(len(o.hidden_states))
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "clip_vision.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
