import logging as _logging
from collections import OrderedDict
from dataclasses import fields


class BaseOutput(OrderedDict):
    def __post_init__(self):
        for f in fields(self):
            v = getattr(self, f.name)
            if v is not None:
                self[f.name] = v

    def to_tuple(self):
        return tuple(self[k] for k in self.keys())


class logging:  # noqa: N801 - mirrors `diffusers.utils.logging`
    @staticmethod
    def get_logger(name):
        return _logging.getLogger(name)


WEIGHTS_NAME = "diffusion_pytorch_model.bin"
