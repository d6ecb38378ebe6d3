"""CPU: the C-ABI library builds/loads and exports every symbol include/haet_hip.h declares
(no compute calls -- there is no GPU here), and the host binding agrees with the header."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "haet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(haet_[a-z_0-9]+)\s*\(", text)))


def test_header_declares_the_path():
    syms = declared_symbols()
    for s in ("haet_slice_fwd", "haet_slice_merge", "haet_token_pos", "haet_balltree_build",
              "haet_erwin_fwd", "haet_deslice_fwd", "haet_last_error_string"):
        assert s in syms


def test_library_exports_every_declared_symbol():
    from haet_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "build with: make -C haet_amd/csrc"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for s in declared_symbols():
        assert hasattr(lib, s), f"{s} declared in include/haet_hip.h but not exported"
    assert set(_lib.SIGNATURES) == set(declared_symbols())


def test_binding_loads_and_reports_errors_without_gpu():
    from haet_amd import _lib

    lib = _lib.load()
    assert lib.haet_abi_version() == 9
    # argument validation happens before any HIP call, so it is testable on CPU
    assert lib.haet_slice_num_parts(1, 1000, 8, 48, 64) < 0  # dim_head 48 unsupported
    assert b"dim_head" in lib.haet_last_error_string()
    assert lib.haet_slice_num_parts(1, 1_000_000, 8, 32, 64) > 0
    assert lib.haet_slice_part_floats(1, 8, 32, 64) == 8 * 64 * 34


def test_erwin_desc_struct_matches_header_size():
    from haet_amd import _lib

    # 5 ints + 8 layers of (5 + 4*16) ints + 8 resamples of 7 ints and 2 floats
    assert ctypes.sizeof(_lib.ErwinDesc) == 4 * (5 + 8 * (5 + 4 * 16) + 8 * 9)
    # haet_erwin_train: 3 pointer tables of 2 * MAX_LEVELS + 2 pointers
    assert ctypes.sizeof(_lib.ErwinTrain) == 8 * (3 * 2 * _lib.MAX_LEVELS + 4)  # + tape pointer, tape_op_floats (padded)
    assert _lib.load().haet_erwin_train_workspace_bytes(ctypes.byref(_lib.ErwinDesc()), 8) >= 64
    d = _lib.ErwinDesc()
    assert _lib.load().haet_erwin_lds_bytes(ctypes.byref(d), 64) == 0  # invalid desc -> 0
