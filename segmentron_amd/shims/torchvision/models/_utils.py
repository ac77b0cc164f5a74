"""IntermediateLayerGetter as imported by segmentron/models/pointrend.py:5: runs the children of
a model in registration order and collects the outputs named in `return_layers`."""
from collections import OrderedDict

import torch.nn as nn


class IntermediateLayerGetter(nn.ModuleDict):
    def __init__(self, model, return_layers):
        names = [n for n, _ in model.named_children()]
        if not set(return_layers).issubset(names):
            raise ValueError("return_layers are not present in model")
        wanted = {str(k): str(v) for k, v in return_layers.items()}
        remaining, layers = dict(wanted), OrderedDict()
        for name, module in model.named_children():
            layers[name] = module
            remaining.pop(name, None)
            if not remaining:
                break
        super().__init__(layers)
        self.return_layers = wanted

    def forward(self, x):
        out = OrderedDict()
        for name, module in self.items():
            x = module(x)
            if name in self.return_layers:
                out[self.return_layers[name]] = x
        return out
