"""The C ABI: include/lcgan_hip.h, the ctypes table and the built library must agree (CPU: no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CT = {"int": "I", "float": "F", "double": "D", "long long": "LL"}


def _parse_header():
    src = open(os.path.join(ROOT, "include", "lcgan_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\bint\s+(lcgan_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(1), m.group(2).strip()
        kinds = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    kinds.append("P")
                else:
                    t = re.sub(r"\b\w+$", "", a).replace("const", "").strip()
                    kinds.append(_CT[t])
        protos[name] = kinds
    return protos


def test_header_matches_ctypes_table():
    from lcgan_amd import _lib
    protos = _parse_header()
    names = {"P": _lib.P, "I": _lib.I, "F": _lib.F, "D": _lib.D, "LL": _lib.LL}
    assert set(protos) == set(_lib.SIGNATURES), set(protos) ^ set(_lib.SIGNATURES)
    for fn, kinds in protos.items():
        assert [names[k] for k in kinds] == _lib.SIGNATURES[fn], fn


def test_library_loads_and_exports_every_declared_symbol():
    from lcgan_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    for fn in _parse_header():
        assert hasattr(lib, fn), fn
    assert lib.lcgan_prof_active() == 0


def test_every_entry_point_cites_the_reference():
    hdr = open(os.path.join(ROOT, "include", "lcgan_hip.h")).read()
    for token in ("custom_layers.py", "cnn.py", "loss.py", "ema.py", "worker.py"):
        assert token in hdr


def test_routing_switches_round_trip_and_reject_unknown():
    """lcgan_set_option is host-only: each of the 27 switches the header lists returns its previous value, an unknown one LCGAN_EINVAL"""
    from lcgan_amd import _lib, build
    build.build(verbose=False)
    lib = _lib.load()
    hdr = re.sub(r"\s*\n\s*\*\s*", " ", open(os.path.join(ROOT, "include", "lcgan_hip.h")).read())
    n_opt = 27
    assert f"; {n_opt - 1} " in hdr and f"; {n_opt} " not in hdr[hdr.index("tuning switches"):hdr.index("int lcgan_set_option")]
    for opt in range(n_opt):
        old = lib.lcgan_set_option(opt, 1)
        assert old >= 0, opt
        assert lib.lcgan_set_option(opt, old) == 1
        assert lib.lcgan_set_option(opt, old) == old
    assert lib.lcgan_set_option(n_opt, 0) == -1
    assert lib.lcgan_set_option(-5, 0) == -1
