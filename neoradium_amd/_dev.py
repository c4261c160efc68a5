"""NumPy <-> device plumbing for the NumPy-in / NumPy-out class surface (torch is only the memory manager)."""
import numpy as np


def device():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("neoradium_amd needs an AMD GPU (MI355X / gfx950): there is no CPU fallback. "
                           "torch.cuda.is_available() is False.")
    return torch.device('cuda', torch.cuda.current_device())


def D(x, dtype=None):
    """Host array -> contiguous device tensor."""
    import torch
    t = torch.from_numpy(np.ascontiguousarray(x))
    t = t.to(device())
    return t if dtype is None else t.to(dtype)


def N(t):
    """Device tensor -> host NumPy array."""
    return t.detach().cpu().numpy()
