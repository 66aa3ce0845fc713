"""Deterministic, construction-order independent weights for model-level parity tests.

Default initialisation makes MSDeformAttn degenerate (zero offset/attention weights) and the class
head silent (bias -4.6), so parity tests perturb every tensor with noise drawn from a generator seeded
by (seed, crc32(name)): the reference model and ours get identical values whatever order their
modules were created in, and the same values are regenerated on the GPU box without shipping weights.
"""
import zlib

import torch


def perturb_state_dict(model, seed=0, scale=1.0):
    sd = model.state_dict()
    new = {}
    for name, t in sd.items():
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if not t.is_floating_point():
            new[name] = t.clone()
            continue
        noise = torch.randn(t.shape, generator=g, dtype=torch.float32).to(t.dtype)
        if name.endswith("running_var"):
            new[name] = (t + 0.2 * scale * noise.abs()).clone()
        elif "sampling_offsets.weight" in name:
            new[name] = t + 0.05 * scale * noise
        elif "attention_weights.weight" in name:
            new[name] = t + 0.1 * scale * noise
        elif "class_embed" in name and name.endswith("bias"):
            b = t * 0 + 0.3 * scale * noise - 0.5  # replace the focal prior so that queries fire
            if b.numel() == 20:
                b[0] += 1.7                        # ... often as class 0 ("person", what the tracker keeps)
            new[name] = b
        elif "class_embed" in name and name.endswith("weight"):
            new[name] = 2.5 * t + 0.3 * scale * noise * t.std().clamp(min=0.02)
        elif "bbox_embed" in name and "layers.2" in name:
            new[name] = t + 0.02 * scale * noise
        elif t.dim() <= 1:
            new[name] = t + 0.05 * scale * noise
        else:
            new[name] = t + 0.05 * scale * noise * t.std().clamp(min=1e-3)
    model.load_state_dict(new, strict=True)
    return model
