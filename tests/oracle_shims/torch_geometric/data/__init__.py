"""Attribute-bag stand-ins for PyG Data / Batch (enough for the golden generator)."""
import torch


class Data:
    def __init__(self, **kw):
        for k, v in kw.items():
            setattr(self, k, v)

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class Batch(Data):
    pass
