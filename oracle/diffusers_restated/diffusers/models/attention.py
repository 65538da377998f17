import torch.nn.functional as F
from torch import nn

from .attention_processor import Attention  # noqa: F401  (re-exported, as diffusers does)


class GEGLU(nn.Module):
    def __init__(self, dim_in, dim_out, bias=True):
        super().__init__()
        self.proj = nn.Linear(dim_in, dim_out * 2, bias=bias)

    def forward(self, hidden_states):
        hidden_states, gate = self.proj(hidden_states).chunk(2, dim=-1)
        return hidden_states * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim, dim_out=None, mult=4, dropout=0.0, activation_fn="geglu", final_dropout=False,
                 inner_dim=None, bias=True):
        super().__init__()
        assert activation_fn == "geglu"
        inner_dim = inner_dim or int(dim * mult)
        dim_out = dim_out or dim
        self.net = nn.ModuleList([GEGLU(dim, inner_dim, bias=bias), nn.Dropout(dropout),
                                  nn.Linear(inner_dim, dim_out, bias=bias)])
        if final_dropout:
            self.net.append(nn.Dropout(dropout))

    def forward(self, hidden_states):
        for m in self.net:
            hidden_states = m(hidden_states)
        return hidden_states


class AdaLayerNorm(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the AVSyncD path")


class AdaLayerNormZero(nn.Module):
    def __init__(self, *a, **k):
        raise NotImplementedError("not on the AVSyncD path")
