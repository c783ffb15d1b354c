"""CPU: the C-ABI library is built, loads, and exports every symbol include/gan_amd.h declares (no compute calls)."""
import ctypes
import re
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parent.parent


def _declared():
    text = (REPO / "include" / "gan_amd.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ga_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    names = _declared()
    for must in ("ga_conv", "ga_wgrad", "ga_pack_weights", "ga_wn_bwd", "ga_gn_film_act_fwd", "ga_gn_film_act_bwd",
                 "ga_avgpool1d_fwd", "ga_stft_loss_fwd", "ga_reduce", "ga_grad_sqnorm", "ga_adamw_fused"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from gan_amd import _lib
    if not _lib.LIB_PATH.exists():
        import __graft_entry__
        __graft_entry__.build()
    lib = ctypes.CDLL(str(_lib.LIB_PATH))
    missing = [n for n in _declared() if not hasattr(lib, n)]
    assert not missing, missing
    assert sorted(_lib.EXPORTED) == _declared()
    lib.ga_build_target.restype = ctypes.c_char_p
    assert lib.ga_build_target() == b"gfx950"


def test_product_path_has_no_cpu_fallback():
    """CPU tensors are refused loudly; nothing under gan_amd/ imports the oracle."""
    import torch
    from gan_amd.lab3_codec_models import MultiScaleWaveDiscriminator, hinge_g_loss
    with pytest.raises(RuntimeError, match="GPU only"):
        MultiScaleWaveDiscriminator(1)(torch.zeros(1, 1, 256))
    with pytest.raises(RuntimeError, match="GPU only"):
        hinge_g_loss(torch.zeros(1, 1, 8))
    for f in (REPO / "gan_amd").rglob("*.py"):
        assert "oracle" not in f.read_text().replace("use oracle/ for CPU checks", ""), f
