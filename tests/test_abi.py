"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol
include/rlarm_hip.h declares, the ctypes table covers the header, and the product never
imports the oracle."""
import ctypes
import os
import re

import pytest

from conftest import REPO

HEADER = os.path.join(REPO, "include", "rlarm_hip.h")
DEBUG_HEADER = os.path.join(REPO, "include", "rlarm_hip_debug.h")
PKG = os.path.join(REPO, "rl_arm_under_sparse_reward_amd")


def _symbols(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(hp_[a-z0-9_]+)\s*\(", txt)))


def header_symbols():
    """stable surface + diagnostics: everything the library must export"""
    return sorted(set(_symbols(HEADER)) | set(_symbols(DEBUG_HEADER)))


def test_library_exports_every_header_symbol():
    so = os.path.join(PKG, "librlarm_hip.so")
    if not os.path.exists(so):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(so)
    syms = header_symbols()
    assert len(syms) > 40
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_ctypes_table_matches_header():
    from rl_arm_under_sparse_reward_amd import _lib
    assert sorted(_lib.PROTOTYPES) == header_symbols()


def test_stable_and_diagnostic_surfaces_are_separate():
    """HP_ABI_VERSION names the stable header only: diagnostics and test hooks live in rlarm_hip_debug.h, and the version the
    Python binding checks is the header's."""
    from rl_arm_under_sparse_reward_amd import _lib
    stable, debug = set(_symbols(HEADER)), set(_symbols(DEBUG_HEADER))
    assert not (stable & debug), stable & debug
    assert debug == _lib.DEBUG_SYMBOLS
    assert not [s for s in stable if "debug" in s or s in ("hp_agent_set_adam", "hp_ctx_launch_floor")]
    version = int(re.search(r"#define\s+HP_ABI_VERSION\s+(\d+)", open(HEADER).read()).group(1))
    assert version == _lib.ABI_VERSION == 4
    if os.path.exists(os.path.join(PKG, "librlarm_hip.so")):
        assert ctypes.CDLL(os.path.join(PKG, "librlarm_hip.so")).hp_abi_version() == version


def test_no_device_fails_loudly_not_silently():
    from rl_arm_under_sparse_reward_amd import _lib
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(_lib.HpError, match="no CPU fallback"):
        _lib.Context(0)


def test_product_never_imports_oracle():
    offenders = []
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M) or "oracle/" in src.replace(
                        "oracle/running_norm.py for the probe", ""):
                    offenders.append(f)
    assert not offenders, offenders


def test_squared_threshold_matches_oracle():
    from oracle.her_replay import squared_distance_threshold
    from rl_arm_under_sparse_reward_amd.her import squared_threshold
    for thr in (0.05, 0.01, 0.1, 1.0, 0.049999999, 3.3e-3):
        assert squared_threshold(thr) == squared_distance_threshold(thr)


def test_pending_updates_are_issued_in_front_of_any_other_library_call():
    """_lib.py "deferred updates": the library proxy issues whatever is pending before it forwards a call -- except the two
    entry points error handling itself needs.  No device involved: the forwarded call may fail, the order is what counts."""
    import ctypes as C
    from rl_arm_under_sparse_reward_amd import _lib
    lib = _lib.load()
    calls = []

    class Pending:
        def _flush_updates(self):
            calls.append("flush")

    p = Pending()
    _lib.register_pending(p)
    _lib.register_pending(p)                   # registering twice is one entry
    lib.hp_abi_version()                       # no flush in front of these two
    lib.hp_last_error()
    assert calls == []
    h = C.c_void_p()
    lib.hp_ctx_create(10 ** 6, C.byref(h))     # any other entry point (this one fails: no such device) flushes first
    assert calls == ["flush"]
    lib.hp_ctx_create(10 ** 6, C.byref(h))     # ... once: nothing is pending any more
    assert calls == ["flush"]
    _lib.register_pending(p)
    _lib.unregister_pending(p)
    lib.hp_ctx_create(10 ** 6, C.byref(h))
    assert calls == ["flush"]


def test_switch_list_is_the_documented_one():
    """Every RLARM_* switch the product reads is in DESIGN.md section 4's list, the list names nothing else, and it stays at 15
    (the forms measured and lost go with their switches) + the one failure-injection hook of round 6
    (RLARM_PEER_INJECT)."""
    read = set()
    for root, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc")):
                src = open(os.path.join(root, f), encoding="utf-8").read()
                if f.endswith(".py"):
                    read |= set(re.findall(r"""environ(?:\.get|\.pop|\.setdefault)?[\(\[]\s*["'](RLARM_[A-Z0-9_]+)""", src))
                else:
                    read |= set(re.findall(r"""(?:getenv|tri)\(\s*"(RLARM_[A-Z0-9_]+)""", src))
    design = open(os.path.join(os.path.dirname(PKG), "DESIGN.md"), encoding="utf-8").read()
    sec = design[design.index("Switches (A/B and debugging"):design.index("## 5. Parity")]
    documented = set(re.findall(r"`(RLARM_[A-Z0-9_]+)", sec))
    assert read == documented, (sorted(read - documented), sorted(documented - read))
    assert len(documented) <= 16, sorted(documented)
