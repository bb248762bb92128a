import importlib


def ensure_tuple_rep(v, dim):
    if isinstance(v, (list, tuple)):
        assert len(v) == dim
        return tuple(v)
    return (v,) * dim


def look_up_option(opt, supported, default="no_default"):
    return supported[opt] if isinstance(supported, dict) else opt


def optional_import(module, name=""):
    try:
        m = importlib.import_module(module)
        return (getattr(m, name) if name else m), True
    except Exception:
        return None, False
