"""CPU test: the C-ABI library builds, loads and exports every symbol include/svo_abi.h declares.
No compute call is made without a GPU."""
import ctypes
import os
import re

import conftest


def _declared_symbols():
    hdr = open(os.path.join(conftest.ROOT, "include", "svo_abi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(svo_[a-z_0-9]+)\s*\(", hdr)))


def test_header_declares_expected_surface():
    syms = _declared_symbols()
    for s in ["svo_create", "svo_destroy", "svo_fast_detect", "svo_build_pyramid", "svo_lk_track",
              "svo_circular_match", "svo_triangulate", "svo_pnp_ransac", "svo_add_frame",
              "svo_track_batch", "svo_last_error"]:
        assert s in syms


def test_library_exports_every_declared_symbol(pkg):
    pkg.build_library()
    lib = ctypes.CDLL(pkg.library_path())
    missing = [s for s in _declared_symbols() if not hasattr(lib, s)]
    assert not missing, missing
    assert lib.svo_abi_version() == 9


def test_default_config_matches_reference_yaml(pkg):
    from importlib import import_module
    b = import_module(conftest.entry.PKG_NAME + ".binding")
    cfg = b.default_config(1241, 376)
    # config/default.yaml:33-47,66,69,77,80-82 and src/tracking.cpp:99,311
    assert cfg.fast_threshold == 20 and cfg.num_features_tracking == 5 and cfg.iterations == 500
    assert abs(cfg.reproj_err - 0.5) < 1e-7 and abs(cfg.confidence - 0.99) < 1e-6
    assert cfg.feature_match_error == 3.0 and cfg.inlier_rate == 0.01
    assert cfg.P1[0] == 718.856 and cfg.P1[2] == 607.193 and cfg.P1[6] == 185.216
    assert abs(cfg.P2[3] - 718.856 * -0.537) < 1e-12
    assert cfg.min_move2 == 0.0005 * 0.0005 and cfg.max_move2 == 100.0
    # ORBextractor arguments, config/default.yaml:89-93
    assert (cfg.orb_nfeatures, cfg.orb_nlevels, cfg.orb_ini_th, cfg.orb_min_th) == (2000, 8, 20, 7)
    assert abs(cfg.orb_scale_factor - 1.2) < 1e-7 and cfg.track_mode == 0
    # ABI v6 additions default to the reference's behaviour: every FAST corner, the canonical (order-free) LK sums
    assert cfg.lk_accum == b.LK_ACCUM_EXACT and cfg.fast_keep_strongest == 0
    assert b.load_library().svo_config_bytes() == ctypes.sizeof(b.Config)


def test_create_fails_loudly_without_gpu(pkg):
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(pkg.SvoError):
        pkg.Context(416, 128)
