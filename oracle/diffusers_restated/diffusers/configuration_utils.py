import functools
import inspect


class FrozenDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e


class ConfigMixin:
    config_name = "config.json"

    def register_to_config(self, **kwargs):
        self._internal_dict = FrozenDict(kwargs)

    @property
    def config(self):
        return self._internal_dict

    @classmethod
    def from_config(cls, config, **kwargs):
        sig = inspect.signature(cls.__init__).parameters
        args = {k: v for k, v in dict(config).items() if k in sig and not k.startswith("_")}
        args.update(kwargs)
        return cls(**args)


def register_to_config(init):
    @functools.wraps(init)
    def inner(self, *args, **kwargs):
        sig = inspect.signature(init)
        bound = sig.bind(self, *args, **kwargs)
        bound.apply_defaults()
        cfg = {k: v for k, v in bound.arguments.items() if k != "self" and not k.startswith("_")}
        init(self, *args, **kwargs)
        self.register_to_config(**cfg)

    return inner
