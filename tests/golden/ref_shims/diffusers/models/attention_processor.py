"""Self-attention with diffusers' Attention / AttnProcessor2_0 semantics as used at
reference transformer.py:180-188,253-258: q/k/v Linear without bias, out Linear with bias,
mask [B, T] -> [B, heads, 1, T] handed to SDPA unchanged (a float mask is ADDITIVE)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Attention(nn.Module):
    def __init__(self, query_dim, cross_attention_dim=None, heads=8, dim_head=64, dropout=0.0, bias=False,
                 upcast_attention=False, **_):
        super().__init__()
        assert cross_attention_dim is None
        inner = heads * dim_head
        self.heads = heads
        self.to_q = nn.Linear(query_dim, inner, bias=bias)
        self.to_k = nn.Linear(query_dim, inner, bias=bias)
        self.to_v = nn.Linear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([nn.Linear(inner, query_dim, bias=True), nn.Dropout(dropout)])

    def forward(self, hidden_states, encoder_hidden_states=None, attention_mask=None, **_):
        assert encoder_hidden_states is None
        b, t, _c = hidden_states.shape
        if attention_mask is not None:
            # prepare_attention_mask: repeat_interleave over heads, then view [B, heads, 1, T]
            attention_mask = attention_mask.repeat_interleave(self.heads, dim=0)
            attention_mask = attention_mask.view(b, self.heads, -1, attention_mask.shape[-1])
        q = self.to_q(hidden_states)
        k = self.to_k(hidden_states)
        v = self.to_v(hidden_states)
        hd = q.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, hd).transpose(1, 2)
        k = k.view(b, -1, self.heads, hd).transpose(1, 2)
        v = v.view(b, -1, self.heads, hd).transpose(1, 2)
        o = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=0.0, is_causal=False)
        o = o.transpose(1, 2).reshape(b, -1, self.heads * hd).to(q.dtype)
        o = self.to_out[0](o)
        return self.to_out[1](o)
