#!/usr/bin/env python3
"""Golden vectors for the Lovasz-softmax loss: imports /root/reference/src/utils/lovasz_losses.py (pure
torch) IN THIS CONTAINER and records inputs, loss values and input gradients for the call form the
reference's training step uses (`lovasz_softmax(F.softmax(logits, 1), labels)`, base_exp.py:519-575).
Run once here; the .npz travels, the reference does not."""
import importlib.util, os, sys
import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/src/utils/lovasz_losses.py"
spec = importlib.util.spec_from_file_location("ref_lovasz", REF)
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

out = {}
g = torch.Generator().manual_seed(7)
cases = [("small", 40, 5, None), ("present_subset", 300, 18, [0, 3, 4, 11, 17]), ("one_class", 64, 6, [2]),
         ("single_pixel", 1, 4, None), ("large", 3000, 18, None)]
for name, P, C, allowed in cases:
    logits = torch.randn(P, C, generator=g) * 2      # fp32: the reference mixes a float() Jaccard vector in
    if allowed is None:
        labels = torch.randint(0, C, (P,), generator=g)
    else:
        labels = torch.tensor(allowed)[torch.randint(0, len(allowed), (P,), generator=g)]
    x = logits.clone().requires_grad_(True)
    loss = ref.lovasz_softmax(F.softmax(x, dim=1), labels)
    loss.backward()
    out[name + "_logits"] = logits.numpy()
    out[name + "_labels"] = labels.numpy()
    out[name + "_loss"] = np.float64(loss.item())
    out[name + "_grad"] = x.grad.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lovasz_golden.npz"), **out)
print({k: v for k, v in out.items() if k.endswith("_loss")})
