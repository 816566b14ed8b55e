import torch.nn as nn


class ConformerBlock(nn.Module):  # only subclassed (reference decoder.py:163); never instantiated
    def __init__(self, *a, **k):
        raise NotImplementedError
