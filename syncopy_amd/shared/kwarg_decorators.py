"""FieldTrip-style `cfg` calls of the metafunctions (contract of syncopy/shared/kwarg_decorators.py:32-300,
`unwrap_cfg`; `StructDict` as syncopy/shared/tools.py:27-68)."""
import contextlib
import functools

from .errors import SPYError, SPYTypeError, SPYValueError


class StructDict(dict):
    """dict whose items are attributes as well (`cfg.method = "coh"`)."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]


@contextlib.contextmanager
def attached_selection(data, select):
    """`select=` of a metafunction call (shared/kwarg_decorators.py:302-415, `unwrap_select`): the selection is attached
    for the duration of the call and removed afterwards - only if this call attached it.  A selection the user made
    in place beforehand (`data.selectdata(...)`) is honoured and left alone; giving both is an error."""
    mine = False
    if select is not None:
        if data.selection is not None:
            raise SPYError(f"Selection found both in kwarg 'select' ({select}) and in passed Syncopy Data object of "
                           f"type '{type(data).__name__}'")
        data.selectdata(select)
        mine = True
    try:
        yield data
    finally:
        if mine:
            data.selection = None


def get_defaults(func):
    """Keyword defaults of a metafunction as a StructDict (spy.get_defaults, shared/tools.py:152-221)."""
    import inspect
    func = getattr(func, "__wrapped__", func)
    skip = {"select", "compute_method", "routine_classes"}
    return StructDict({k: v.default for k, v in inspect.signature(func).parameters.items()
                       if v.default is not v.empty and k not in skip})


def unwrap_cfg(func):
    """Accept `func(cfg, data)`, `func(data, cfg)`, `func(data, cfg=cfg)`, `func(cfg)` / `func(cfg=cfg)` with the data
    object in `cfg.data` or `cfg.dataset`, next to plain keyword calls; a saved `out.cfg` ({func name: {...}}) replays
    a call; "yes"/"no" entries become booleans.  A parameter may come from `cfg` or from a keyword, never both."""
    from ..datatype import _Base

    @functools.wraps(func)
    def wrapper_cfg(*args, **kwargs):
        args = list(args)
        dicts = [i for i, a in enumerate(args) if isinstance(a, dict)]
        if len(dicts) > 1:
            raise SPYValueError(legal="single `cfg` input", varname="cfg",
                                actual=f"{len(dicts)} `cfg` objects in input arguments")
        cfg = args.pop(dicts[0]) if dicts else None
        if kwargs.get("cfg") is not None:
            if cfg:
                raise SPYValueError(legal="`cfg` either as positional or keyword argument, not both", varname="cfg")
            cfg = kwargs.pop("cfg")
        else:
            kwargs.pop("cfg", None)
        if cfg:
            if not isinstance(cfg, dict):
                raise SPYTypeError(cfg, varname="cfg", expected="dictionary-like")
            if func.__name__ in cfg:
                cfg = cfg[func.__name__]                    # replay of a saved front-end call
            cfg = StructDict(cfg)                           # a copy: the user's cfg is left alone
            for key in kwargs:
                if key not in ("data", "dataset") and key in cfg:
                    raise SPYValueError(legal=f"parameter set either via `cfg.{key}=...` or directly via keyword",
                                        varname=f"cfg.{key} & {key}", actual="set in both")
            for key, val in list(cfg.items()):
                if isinstance(val, str) and val in ("yes", "no"):
                    cfg[key] = val == "yes"
        else:
            cfg = StructDict()
        data = cfg.pop("data", None)
        if cfg.get("dataset") is not None:
            if data is not None:
                raise SPYValueError(legal="either 'data' or 'dataset' in `cfg`/keywords, not both", varname="cfg")
            data = cfg.pop("dataset")
        for key in ("data", "dataset"):
            if kwargs.get(key) is not None:
                if data is not None:
                    raise SPYValueError(legal="Syncopy data object provided either via `cfg` or as keyword argument, "
                                              "not both", varname="cfg.data")
                data = kwargs.pop(key)
        if data is not None and any(isinstance(a, _Base) for a in args):
            raise SPYValueError(legal="Syncopy data object provided either via `cfg`/keyword or positional "
                                      "arguments, not both", varname="cfg/data")
        if data is None:
            objs = [a for a in args if isinstance(a, _Base)]
            if len(objs) > 1:
                raise SPYValueError("only one Syncopy data object", varname="data")
            if not objs:
                cfg.update(kwargs)
                return func(*args, **cfg)                   # no data object: the metafunction raises its own error
            data = objs[0]
            args = [a for a in args if a is not data]
        if not isinstance(data, _Base):
            raise SPYTypeError(data, varname="data", expected="Syncopy data object")
        cfg.update(kwargs)
        res = func(data, *args, **cfg)
        # replay record (connectivity_analysis.py:765-770 / freqanalysis.py:1058-1062): the cfgs of earlier front-end
        # calls travel with the data, this call's parameters are filed under the function's name
        record = StructDict({k: v for k, v in (getattr(data, "cfg", None) or {}).items() if isinstance(v, dict)})
        record[func.__name__] = StructDict({k: v for k, v in cfg.items()
                                            if k not in ("compute_method", "routine_classes")})
        try:
            res.cfg = record
        except AttributeError:
            pass
        return res

    return wrapper_cfg
