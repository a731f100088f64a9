if "." not in __name__:
    # imported as a top-level package (this directory's parent is on sys.path, the reference's layout):
    # become an alias of the real package so that relative imports resolve -- see ../_alias.py
    import importlib.util as _u
    import os as _os
    _s = _u.spec_from_file_location("_p2p_alias_boot", _os.path.join(_os.path.dirname(_os.path.dirname(
        _os.path.abspath(__file__))), "_alias.py"))
    _m = _u.module_from_spec(_s)
    _s.loader.exec_module(_m)
    _m.bootstrap(__name__, __file__)
