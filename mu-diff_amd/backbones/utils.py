"""Model registry with the reference's names (backbones/utils.py:8-29 of the reference)."""
_MODELS = {}


def register_model(cls=None, *, name=None):
    def _register(c):
        key = c.__name__ if name is None else name
        if key in _MODELS:
            raise ValueError(f'Already registered model with name: {key}')
        _MODELS[key] = c
        return c
    return _register if cls is None else _register(cls)


def get_model(name):
    return _MODELS[name]
