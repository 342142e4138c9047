import torch


def _get_device(mod):
    if isinstance(mod, torch.nn.Module):
        for t in list(mod.parameters()) + list(mod.buffers()):
            return t.device
    return "cpu"
