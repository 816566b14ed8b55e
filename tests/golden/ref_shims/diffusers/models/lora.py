import torch.nn as nn


class LoRACompatibleLinear(nn.Linear):
    pass
