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


def main():
    from transformers import CLIPVisionConfig, CLIPVisionModelWithProjection
    out = {}
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
