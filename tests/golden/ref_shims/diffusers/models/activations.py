import torch.nn as nn


def get_activation(name):
    table = {"silu": nn.SiLU, "swish": nn.SiLU, "mish": nn.Mish, "gelu": nn.GELU, "relu": nn.ReLU}
    return table[name.lower()]()
