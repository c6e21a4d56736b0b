"""CPU-side checks of the C ABI: the library builds for gfx950, loads, and exports every symbol
include/melissa_hip.h declares; argument validation works without touching a GPU."""
import ctypes as C
import os
import re

import pytest

from melissa_amd import _lib, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    return _lib.load()


def test_header_symbols_are_exported(lib):
    header = open(os.path.join(ROOT, "include", "melissa_hip.h")).read()
    declared = set(re.findall(r"^(?:mel_status|size_t|const char\*|void\*|void|int32_t)\s+(mel_[a-z_0-9]+)\(", header, flags=re.M))
    assert declared == set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_header(lib):
    # layout sanity of the ctypes mirrors (8-byte pointers, natural alignment)
    assert C.sizeof(_lib.MelLinear) == 24
    assert C.sizeof(_lib.MelMlp) == 24 * _lib.MAX_HEAD_LAYERS + 8
    # every mirror against sizeof() as the library was compiled
    mirrors = [_lib.MelLinear, _lib.MelGatv2, _lib.MelMlp, _lib.MelWeights, _lib.MelSelect, _lib.MelEnvBatch,
               _lib.MelEpisodePool, _lib.MelEnvObs, _lib.MelRoundReplay, _lib.MelGraphPool, _lib.MelEpisodeStream,
               _lib.MelReplayBatch, _lib.MelAdamTensors]
    for which, cls in enumerate(mirrors):
        assert C.sizeof(cls) == lib.mel_abi_sizeof(which), cls.__name__
    assert lib.mel_abi_sizeof(99) == 0


def test_validation_errors_without_gpu(lib):
    w = _lib.MelWeights()
    assert lib.mel_workspace_bytes(None, 4, 20) == 0
    st = lib.mel_ldgn_forward(None, None, 4, 20, 161, None, None, 0, None)
    assert st == _lib.ERR_INVALID_ARG and b"null" in lib.mel_last_error()
    w.model = _lib.MODEL_HLDGN
    st = lib.mel_ldgn_forward(C.byref(w), None, 4, 20, 161, None, None, 0, None)
    assert st == _lib.ERR_INVALID_ARG
    assert lib.mel_env_state_bytes(0, 20) == 0 and lib.mel_env_state_bytes(4, 129) == 0       # MEL_MAX_NODES = 128
    assert lib.mel_env_state_bytes(4, 50) > 0
    # two-word node sets beyond 64 nodes: the state grows by the second word of every set
    assert lib.mel_env_state_bytes(4, 100) > 2 * lib.mel_env_state_bytes(4, 50) - 4096
    assert lib.mel_version().startswith(b"melissa_hip")


def test_shape_error_maps_to_value_error(lib):
    import torch
    from melissa_amd.networks import LDGNNetwork
    net = LDGNNetwork(5, 128, 2, 4, 20, dueling_param=({"hidden_sizes": [128, 128]}, {"hidden_sizes": [128, 128]}))
    w = net._weights()
    st = lib.mel_ldgn_forward(C.byref(w), None, 4, 20, 160, None, None, 0, None)
    assert st == _lib.ERR_SHAPE
    with pytest.raises(ValueError, match="Expected 160 feature cols for nodes, got 159"):
        _lib.check(st)
    assert lib.mel_workspace_bytes(C.byref(w), 1024, 50) > 0


def test_build_is_fresh():
    assert os.path.exists(build.LIB_PATH)
