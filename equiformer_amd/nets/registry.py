"""Model registry -- the drop-in boundary of the reference (nets/registry.py:15-44):
`@register_model` records a factory under its function name, `model_entrypoint(name)` returns it."""

__all__ = ["model_entrypoint", "register_model", "list_models"]

_ENTRYPOINTS = {}
_MODULE_OF = {}


def register_model(fn):
    name = fn.__name__
    _ENTRYPOINTS[name] = fn
    _MODULE_OF[name] = fn.__module__.rsplit(".", 1)[-1]
    return fn


def model_entrypoint(model_name):
    try:
        return _ENTRYPOINTS[model_name]
    except KeyError:
        raise KeyError("unknown model %r; registered: %s" % (model_name, ", ".join(sorted(_ENTRYPOINTS)))) from None


def list_models():
    return sorted(_ENTRYPOINTS)
