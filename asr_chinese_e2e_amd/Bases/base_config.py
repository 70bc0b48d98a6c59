"""Config container with the reference's merge semantics (Predictor/Bases/base_config.py:5-46):
class attributes are defaults, fn_build(kwargs) overrides AND silently adds unknown keys,
fn_combine(other) merges another config's public attributes over this one."""


class BaseConfig:
    def _public(self):
        for name in dir(self):
            if not name.startswith("_") and not name.startswith("fn_"):
                yield name, getattr(self, name)

    def fn_build(self, kwargs, verbose=False):
        for k, v in kwargs.items():
            if verbose:
                if hasattr(self, k):
                    if getattr(self, k) != v:
                        print(f"\tchanged {k}:{getattr(self, k)} to {v}")
                else:
                    print(f"\tadd {k}:{v}")
            setattr(self, k, v)
        return self

    def fn_combine(self, config, verbose=False):
        for k, v in config.fn_get_attrs():
            if verbose and hasattr(self, k) and getattr(self, k) != v:
                print(f"\tchanged {k}:{getattr(self, k)} to {v}")
            setattr(self, k, v)
        return self

    def fn_get_attrs(self):
        yield from self._public()

    def fn_show(self):
        print("\nconfigs: ")
        for k, v in self._public():
            print(f"\t{k}:\t\t{v}")
        print("\n")

    def fn_save(self, path):
        import torch
        torch.save(dict(self._public()), path)

    def fn_load(self, path):
        import torch
        self.fn_build(torch.load(path, weights_only=True))
        return self
