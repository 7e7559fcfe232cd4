"""CPU checks of the boundary: the C-ABI library loads and exports every symbol
``include/pocomc_amd.h`` declares; struct layouts match; host logic (MAF spec / packing)."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "pocomc_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pmc_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from pocomc_amd import _lib
    lib = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(lib, s), s
        assert s in _lib.SIGNATURES, f"{s} declared in the header but not bound"
    assert lib.pmc_abi_version() == 2


def test_struct_sizes():
    from pocomc_amd import _lib
    assert ctypes.sizeof(_lib.pmc_maf_t) == 64
    assert ctypes.sizeof(_lib.pmc_scaler_t) == 7 * 8 + 4 * 4 + 8
    assert ctypes.sizeof(_lib.pmc_rng_t) == 48
    assert ctypes.sizeof(_lib.pmc_state_t) == 56
    assert ctypes.sizeof(_lib.pmc_proposal_t) == 72


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pocomc_amd import _lib, Flow
    with pytest.raises(_lib.PocomcAmdError):
        Flow(4, "maf3")


@pytest.mark.parametrize("D,T", [(2, 3), (3, 2), (4, 3), (10, 3), (32, 3), (50, 6)])
def test_pack_index_reproduces_masked_weights(D, T):
    """The packed image holds exactly the masked canonical weights (host logic)."""
    from pocomc_amd.maf_spec import MAFSpec
    spec = MAFSpec(D, T)
    flat = spec.init_params(0)
    idx = spec.pack_index()
    packed = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float32)
    assert packed.size == spec.pk_size
    for t in range(T):
        M0, M1, M2, M3 = spec.masks(t)
        base = t * spec.pk_per_transform
        rank = spec.orders[t]
        feat_of_rank = np.argsort(rank)
        su = spec.slot_unit
        # W0 natural [rank][slot]
        w0n = packed[base + spec.pk_offsets["w0n"]: base + spec.pk_offsets["w0n"] + spec.sz_w0n].reshape(spec.Dp, spec.Hp)
        W0 = spec.view(flat, t, "W0") * M0
        for s in range(spec.Hp):
            for r in range(D):
                exp = W0[su[s], feat_of_rank[r]] if su[s] >= 0 else 0.0
                assert w0n[r, s] == exp
        # fragment arrays reproduce W1 (A[i][k] of tile T, K-tile K, chunk c)
        f1 = packed[base + spec.pk_offsets["f1"]: base + spec.pk_offsets["f1"] + spec.sz_f12].reshape(spec.nT, spec.nT, 64, 4)
        W1 = spec.view(flat, t, "W1") * M1
        lane = np.arange(64)
        for Tt in range(spec.nT):
            for K in range(spec.nT):
                for c in range(4):
                    o = su[16 * Tt + (lane & 15)]
                    i = su[16 * K + 4 * c + (lane >> 4)]
                    exp = np.where((o >= 0) & (i >= 0), W1[np.maximum(o, 0), np.maximum(i, 0)], 0.0)
                    np.testing.assert_array_equal(f1[Tt, K, :, c], exp.astype(np.float32))
        # every unmasked weight appears; units sorted by degree => lower block-triangular tiles
        if spec.tri_ok:
            for Tt in range(spec.nT):
                for K in range(Tt + 1, spec.nT):
                    assert not f1[Tt, K].any()
    # degree groups never straddle a tile when tri_ok
    if spec.tri_ok:
        qd = spec.quad_deg.reshape(-1, 4)
        for g in range(1, D):
            tiles = {i for i in range(qd.shape[0]) if (qd[i] == g).any()}
            assert len(tiles) == 1


def test_flop_accounting_matches_survey():
    from pocomc_amd.maf_spec import MAFSpec
    s = MAFSpec(32, 3)
    assert s.hidden == 128
    assert s.flops_forward_dense() == 270336
    assert s.flops_inverse_naive() == 33 * 270336
    assert MAFSpec(50, 6).flops_forward_dense() == 2033664
    assert MAFSpec(128, 8, 512).flops_forward_dense() == 11534336
