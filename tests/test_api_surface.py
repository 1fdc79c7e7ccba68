"""Drop-in check: every public callable of the reference exists here with the same parameters (names, order,
kinds, defaults).  Extra trailing keyword parameters with defaults are allowed (documented extensions)."""
import inspect

import sys

import qsparse_amd as qs
from golden_io import Golden

# the package re-exports `quantize` / `convert` (functions) under the names of their modules, like the reference
Q, S, U, FU, IM = (sys.modules["qsparse_amd." + m] for m in ("quantize", "sparse", "util", "fuse", "imitation"))

MODS = {"": qs, "quantize.": Q, "sparse.": S, "util.": U, "imitation.": IM, "fuse.": FU}


def _sig(obj):
    target = obj.__init__ if inspect.isclass(obj) else obj
    out = []
    for name, prm in inspect.signature(target).parameters.items():
        if name == "self":
            continue
        d = prm.default
        if d is inspect.Parameter.empty:
            dflt = "<required>"
        elif isinstance(d, (int, float, str, bool, type(None))):
            dflt = repr(d)
        elif isinstance(d, (list, tuple, set)):
            dflt = repr(sorted(d) if isinstance(d, set) else list(d))
        else:
            dflt = "<" + type(d).__name__ + ">"
        out.append([name, str(prm.kind), dflt])
    return out


def _compatible(mine, ref, what):
    assert len(mine) >= len(ref), (what, mine, ref)
    for a, b in zip(mine, ref):
        assert a == b, (what, a, b)
    for extra in mine[len(ref):]:
        assert extra[2] != "<required>" or extra[1] in ("VAR_KEYWORD", "VAR_POSITIONAL"), (what, extra)


def test_public_api_matches_reference():
    g = Golden("f12_api_surface")
    api = g.cases[0]["api"]
    assert len(api) >= 27
    for name, ref in api.items():
        prefix = next(p for p in sorted(MODS, key=len, reverse=True) if name.startswith(p))
        obj = getattr(MODS[prefix], name[len(prefix):])
        _compatible(_sig(obj), ref, name)


def test_callback_protocol_methods_match_reference():
    g = Golden("f12_api_surface")
    for name, ref in g.cases[0]["methods"].items():
        cls, meth = name.split(".")
        obj = getattr(getattr(Q, cls, None) or getattr(S, cls), meth)
        _compatible(_sig(obj), ref, name)
