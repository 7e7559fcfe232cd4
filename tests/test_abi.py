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
    assert lib.pmc_abi_version() == 9


def test_product_library_has_no_measurement_hooks():
    """The default build exports no ``pmc_debug_*`` entry point and never reads the environment on a launch path: the
    in-kernel profiles and A/B switches exist only in ``make DEBUG_HOOKS=1`` (``libpocomc_amd_debug.so``)."""
    import subprocess
    from pocomc_amd import _lib
    _lib.load()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    assert len([s for s in exported if s.startswith("pmc_")]) >= 80
    assert not [s for s in exported if s.startswith("pmc_debug")]
    csrc = os.path.join(ROOT, "pocomc_amd", "csrc")
    uses = [f for f in sorted(os.listdir(csrc)) if f.endswith(".hip") and "getenv" in open(os.path.join(csrc, f)).read()]
    assert uses == [], uses


def test_a_stale_library_is_refused(monkeypatch):
    """``pmc_build_id()`` is the hash of the sources the library was built from; ``_lib.load()`` recomputes it from the
    tree and raises for a library built from other sources (here: the tree's id is made to differ)."""
    from pocomc_amd import _lib
    lib = _lib.load()
    assert lib.pmc_build_id().decode().split("+")[0] == _lib.source_build_id()
    assert re.fullmatch(r"[0-9a-f]{16}", _lib.source_build_id())
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "source_build_id", lambda: "0123456789abcdef")
    with pytest.raises(_lib.PocomcAmdError, match="built from other sources"):
        _lib.load()


def test_struct_sizes():
    from pocomc_amd import _lib
    assert ctypes.sizeof(_lib.pmc_maf_t) == 80
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


@pytest.mark.parametrize("D,T,H,uni", [(2, 3, None, "affine"), (10, 3, None, "affine"), (32, 3, None, "rqs"),
                                       (50, 6, 256, "affine"), (128, 8, 512, "affine"), (17, 2, None, "rqs")])
def test_train_jobs_cover_every_gradient_entry_exactly_once(D, T, H, uni):
    """MAFSpec.train_jobs (the work list of maf_dw_kernel): every unmasked parameter -- weight or bias -- is written by
    exactly one job, masked entries by none, and a job's operand tiles lie inside a row set's scratch block."""
    from pocomc_amd.maf_spec import MAFSpec
    s = MAFSpec(D, T, H, univariate=uni)
    jobs = s.train_jobs()
    _, gm = s.train_index()
    assert jobs.dtype == np.int32 and jobs.shape[1] == 8 and len(jobs) > 0
    hits = np.zeros(s.n_params, dtype=np.int64)
    sizes = {0: (T + 1) * s.Dp * 16, 1: T * 3 * s.Hp * 16, 2: T * 3 * s.Hp * 16, 3: T * s.par_per_transform()}
    assert jobs[0][0] >= 0                                   # (block 0 also adds up the loss)
    assert (jobs[:, 0] < 0).sum() <= 8 + len(jobs) // 4     # padding of the XCD placement stays small
    for ka, oa, kb, ob, gw, gb, _, _ in jobs:
        if ka < 0:
            assert gw < 0 and gb < 0
            continue
        assert 0 <= oa and oa + 256 <= sizes[ka] and oa % 256 == 0
        if gw >= 0:
            assert 0 <= ob and ob + 256 <= sizes[kb] and ob % 256 == 0
            g = gm[gw: gw + 256]
            assert (g >= 0).any()
            np.add.at(hits, g[g >= 0], 1)
        if gb >= 0:
            g = gm[gb: gb + 16]
            np.add.at(hits, g[g >= 0], 1)
    mask = s.mask_flat() != 0
    assert np.array_equal(hits[mask], np.ones(mask.sum(), dtype=np.int64)) and not hits[~mask].any()


@pytest.mark.parametrize("D,T", [(2, 3), (4, 3), (10, 3), (32, 3)])
def test_spline_flow_packing(D, T):
    """nsf*: 23 hyper-network outputs per feature (pocomc/flow.py:69-86, bins=8); the tight output
    fragments and the per-rank padded copy used by the inverse sweep both hold exactly the masked W3 / b3."""
    from pocomc_amd.maf_spec import MAFSpec, spec_by_name
    assert spec_by_name(6, "nsf6").n_out == 23 and spec_by_name(6, "nsf6").n_transforms == 6
    assert spec_by_name(6, "maf12").n_out == 2
    spec = MAFSpec(D, T, univariate="rqs")
    assert spec.n_out == 23 and spec.Op == 23 * spec.Dp and spec.nOT * 16 == spec.Op
    flat = spec.init_params(1)
    idx = spec.pack_index()
    packed = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float32)
    su = spec.slot_unit
    lane = np.arange(64)
    for t in range(T):
        M3 = spec.masks(t)[3]
        W3 = spec.view(flat, t, "W3") * M3
        b3 = spec.view(flat, t, "b3")
        feat_of_rank = np.argsort(spec.orders[t])
        base = t * spec.pk_per_transform
        f3i = packed[base + spec.pk_offsets["f3i"]: base + spec.pk_offsets["f3i"] + spec.sz_f3i].reshape(D, 2, spec.nT, 64, 4)
        b3i = packed[base + spec.pk_offsets["b3i"]: base + spec.pk_offsets["b3i"] + spec.sz_b3i].reshape(D, 32)
        f3 = packed[base + spec.pk_offsets["f3"]: base + spec.pk_offsets["f3"] + spec.sz_f3].reshape(spec.nOT, spec.nT, 64, 4)
        for r in range(D):
            feat = feat_of_rank[r]
            np.testing.assert_array_equal(b3i[r, :23], b3[23 * feat: 23 * feat + 23])
            assert not b3i[r, 23:].any()
            for half in range(2):
                j = 16 * half + (lane & 15)
                for K in range(spec.nT):
                    for c in range(4):
                        i = su[16 * K + 4 * c + (lane >> 4)]
                        ok = (j < 23) & (i >= 0)
                        exp = np.where(ok, W3[23 * feat + np.minimum(j, 22), np.maximum(i, 0)], 0.0)
                        np.testing.assert_array_equal(f3i[r, half, K, :, c], exp.astype(np.float32))
        # tight layout: packed output row o = 23 * rank + j
        for O in range(spec.nOT):
            o = 16 * O + (lane & 15)
            r_out, j = o // 23, o % 23
            for K in range(spec.nT):
                for c in range(4):
                    i = su[16 * K + 4 * c + (lane >> 4)]
                    ok = (r_out < D) & (i >= 0)
                    crow = 23 * feat_of_rank[np.minimum(r_out, D - 1)] + j
                    exp = np.where(ok, W3[crow, np.maximum(i, 0)], 0.0)
                    np.testing.assert_array_equal(f3[O, K, :, c], exp.astype(np.float32))


def test_plateau_scheduler_matches_torch():
    """pocomc_amd.train.ReduceLROnPlateau is torch's scheduler as configured at flow.py:275-283."""
    import torch
    from pocomc_amd.train import ReduceLROnPlateau

    class Opt:
        lr = 1e-2

    rng = np.random.default_rng(0)
    metrics = np.concatenate([np.linspace(3.0, 1.0, 15), 1.0 + 0.05 * rng.random(40), np.linspace(1.0, 0.9, 10),
                              0.9 + 0.01 * rng.random(30)])
    p = torch.nn.Parameter(torch.zeros(1))
    topt = torch.optim.SGD([p], lr=1e-2)
    ref = torch.optim.lr_scheduler.ReduceLROnPlateau(topt, mode="min", factor=0.2, patience=5, threshold=1e-4,
                                                     threshold_mode="abs", min_lr=1e-6)
    o = Opt()
    mine = ReduceLROnPlateau(o, patience=5)
    for mval in metrics:
        ref.step(float(mval))
        mine.step(float(mval))
        assert abs(o.lr - topt.param_groups[0]["lr"]) < 1e-15


# ---------------------------------------------------------------------------------------------------------------
# The chain image of the lane-per-walker inverse sweep (MAFSpec.pack_index: cw1 / cw2 / cw0 / cw3 / f0c), checked on the
# CPU by walking csrc/maf_inverse_tri6.hip's decomposition in numpy: helpers multiply the tiles left of the diagonal
# (f1 / f2 / f3 fragments, f0c for the ranks before the previous tile's), the chain the diagonal tile, the layer-0
# columns of the previous and own tile's ranks (window k slots 0-3 / 4-7) and the output rows of its own ranks.
def _tri6_emulate(spec, flat, z):
    import numpy as np
    from pocomc_amd.maf_spec import LOG_SLOPE
    D, nT, nXT, nOT, Hp, Dp = spec.n_dim, spec.nT, spec.nXT, spec.nOT, spec.Hp, spec.Dp
    idx = spec.pack_index()
    packed = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float64)
    lane = np.arange(64)
    li, lk = lane & 15, lane >> 4
    ci, ck = lane & 3, lane >> 2
    n = len(z)
    y = np.asarray(z, np.float64)
    ladj = np.zeros(n)
    tg = spec.tile_groups()
    for t in reversed(range(spec.n_transforms)):
        P = packed[t * spec.pk_per_transform:(t + 1) * spec.pk_per_transform]
        po = spec.pk_offsets
        sec = lambda name, shape: P[po[name]:po[name] + int(np.prod(shape))].reshape(shape)

        def frag_block(f):           # [lane][4] -> dense 16 x 16 block, rows = out slot, cols = in slot
            B = np.zeros((16, 16))
            for c in range(4):
                B[li, 4 * c + lk] = f[:, c]
            return B
        f1, f2 = sec("f1", (nT, nT, 64, 4)), sec("f2", (nT, nT, 64, 4))
        f3, f0c = sec("f3", (nOT, nT, 64, 4)), sec("f0c", (nT, nXT, 64, 4))
        cw1, cw2, cw0 = sec("cw1", (nT, 64, 4)), sec("cw2", (nT, 64, 4)), sec("cw0", (nT, 64, 4))
        cw3 = sec("cw3", (nT, 64, 2))
        b0, b1, b2, b3 = sec("b0", (Hp,)), sec("b1", (Hp,)), sec("b2", (Hp,)), sec("b3", (spec.Op,))

        def chain_block(cw):         # [lane][4 out quads] -> dense 16 (out slot) x 16 (k slot)
            B = np.zeros((16, 16))
            for a in range(4):
                B[4 * a + ci, ck] = cw[:, a]
            return B
        rank = spec.orders[t]
        yr = np.zeros((n, Dp)); yr[:, rank] = y                       # by rank
        x = np.zeros((n, Dp))
        H0, H1, H2 = (np.zeros((n, Hp)) for _ in range(3))
        soft = lambda raw: raw / (1.0 + np.abs(raw / LOG_SLOPE))
        ls0 = soft(b3[1]); x[:, 0] = (yr[:, 0] - b3[0]) * np.exp(-ls0); ladj -= ls0
        a0n = np.zeros((n, 16))
        a0n += np.outer(x[:, 0], chain_block(cw0[0])[:, 0])           # rank 0 = k slot 0 of tile 0's window
        for T in range(nT):
            own = tg[T]
            if not own:
                break
            # helper: layer 0 (ranks before the previous tile's), hidden layers and output rows left of the diagonal tile
            a0 = b0[16 * T:16 * T + 16] + sum(x[:, 16 * X:16 * X + 16] @ frag_block(f0c[T, X]).T for X in range(nXT)) + a0n
            a0n = np.zeros((n, 16))
            p1 = b1[16 * T:16 * T + 16] + sum(H0[:, 16 * K:16 * K + 16] @ frag_block(f1[T, K]).T for K in range(T))
            p2 = b2[16 * T:16 * T + 16] + sum(H1[:, 16 * K:16 * K + 16] @ frag_block(f2[T, K]).T for K in range(T))
            W1d, W2d, W0w = chain_block(cw1[T]), chain_block(cw2[T]), chain_block(cw0[T])
            W0n = chain_block(cw0[T + 1]) if T + 1 < nT else np.zeros((16, 16))
            W3d = np.zeros((8, 16))                                   # rows (group, out) x k slot
            for sl in range(2):
                W3d[4 * sl + ci, ck] = cw3[T][:, sl]
            qd = spec.quad_deg[4 * T:4 * T + 4]
            for I, d in enumerate(own):
                quads = [j for j in range(4) if qd[j] == d]
                sl_ = np.concatenate([np.arange(4 * j, 4 * j + 4) for j in quads])
                h0 = np.maximum(a0[:, sl_], 0.0); H0[:, 16 * T + sl_] = h0
                p1 = p1 + h0 @ W1d[:, sl_].T                          # own block + blocks into the later quads
                h1 = np.maximum(p1[:, sl_] + h0, 0.0); H1[:, 16 * T + sl_] = h1
                p2 = p2 + h1 @ W2d[:, sl_].T
                h2 = np.maximum(p2[:, sl_] + h1, 0.0); H2[:, 16 * T + sl_] = h2
                O = d >> 3
                rows = [2 * (d & 7), 2 * (d & 7) + 1]
                p3 = b3[16 * O + np.array(rows)] + sum(
                    H2[:, 16 * K:16 * K + 16] @ frag_block(f3[O, K])[rows].T for K in range(T))
                # own tile: every group of the tile that is done so far (incl. this one) feeds the rows through cw3
                done = np.concatenate([np.arange(4 * j, 4 * j + 4) for j in range(4) if qd[j] <= d and qd[j] < D])
                p3 = p3 + H2[:, 16 * T + done] @ W3d[2 * I:2 * I + 2][:, done].T
                ls = soft(p3[:, 1])
                xg = (yr[:, d] - p3[:, 0]) * np.exp(-ls)
                ladj -= ls
                x[:, d] = xg
                a0 = a0 + np.outer(xg, W0w[:, 4 + I])                 # later quads of this tile
                a0n = a0n + np.outer(xg, W0n[:, I])                   # the next tile
        y = x[:, rank]                                                # back to feature order
    return y, ladj


@pytest.mark.parametrize("D,T", [(3, 2), (4, 3), (5, 3), (10, 3), (17, 2), (32, 3), (50, 2)])
def test_chain_image_of_the_lane_sweep_reproduces_the_inverse(D, T):
    import numpy as np
    from oracle.maf import OracleMAF
    from pocomc_amd.maf_spec import MAFSpec
    spec = MAFSpec(D, T)
    if not spec.tri_ok:
        pytest.skip("degree groups wider than a tile")
    flat = (spec.init_params(5) * np.float32(1.2)).astype(np.float32)
    z = (np.random.default_rng(D).normal(size=(12, D)) * 1.2).astype(np.float32)
    xo, lo = OracleMAF(spec, flat).inverse(z)
    x, l = _tri6_emulate(spec, flat, z)
    np.testing.assert_allclose(x, xo, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(l, lo, rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------------------------------------------
# The spline image of the two-wave spline sweep (MAFSpec.pack_index with univariate="rqs": f3i / b3i and -- new in round 3 --
# cw0 / f0c / b0t / b1t / b2t), checked on the CPU by walking csrc/maf_inverse_nsf2.hip's decomposition in numpy:
#   burst wave, a tile ahead:  transposed biases + f0c against the ranks before the previous tile's + f1 / f2 against the
#                              hidden tiles <= T-2; per rank  b3i + f3i against the h2 tiles <= T-2
#   chain wave:                the diagonal tile and the block (T+1, T) of f1 / f2 (right-looking share of the next tile),
#                              the layer-0 window columns cw0, the rank's f3i fragments against the previous tile and the
#                              own tile's quads so far; padding groups absorbed by the last live group; y read from the
#                              previous transform's x through the rank permutation (no re-ranking)
def _nsf2_emulate(spec, flat, z):
    import numpy as np
    from oracle.maf import rqs_inverse
    D, nT, nXT, Hp, Dp, T_ = spec.n_dim, spec.nT, spec.nXT, spec.Hp, spec.Dp, spec.n_transforms
    idx = spec.pack_index()
    packed = np.where(idx >= 0, flat[np.maximum(idx, 0)], 0.0).astype(np.float64)
    lane = np.arange(64)
    li, lk = lane & 15, lane >> 4
    ci, ck = lane & 3, lane >> 2
    n = len(z)
    tg = spec.tile_groups()
    live_tiles = int(spec.device_meta()[7])
    f_o_r = [np.argsort(o) for o in spec.orders]           # feature of rank, per transform
    tr = (np.arange(Hp) & ~15) + 4 * (np.arange(Hp) & 3) + ((np.arange(Hp) >> 2) & 3)
    ladj = np.zeros(n)
    src = None                                              # the previous transform's x, by ITS ranks
    for t in reversed(range(T_)):
        P = packed[t * spec.pk_per_transform:(t + 1) * spec.pk_per_transform]
        po = spec.pk_offsets
        sec = lambda name, shape: P[po[name]:po[name] + int(np.prod(shape))].reshape(shape)

        def frag_block(f):           # [lane][4] -> dense 16 x 16 block, rows = out slot, cols = in slot
            B = np.zeros((16, 16))
            for c in range(4):
                B[li, 4 * c + lk] = f[:, c]
            return B

        def window(cw):              # [lane][4 out quads] -> dense 16 (out slot) x 16 (k slot)
            B = np.zeros((16, 16))
            for a in range(4):
                B[4 * a + ci, ck] = cw[:, a]
            return B
        f1, f2 = sec("f1", (nT, nT, 64, 4)), sec("f2", (nT, nT, 64, 4))
        f0c, cw0 = sec("f0c", (nT, nXT, 64, 4)), sec("cw0", (nT, 64, 4))
        f3i, b3i = sec("f3i", (D, 2, nT, 64, 4)), sec("b3i", (D, 32))
        bt = [sec(k, (Hp,)) for k in ("b0t", "b1t", "b2t")]
        for k, name in enumerate(("b0", "b1", "b2")):                       # transposed biases: slot 16T + 4q + r <- unit 16T + 4r + q
            assert np.array_equal(bt[k], sec(name, (Hp,))[tr])
        b0, b1, b2 = (sec(k, (Hp,)) for k in ("b0", "b1", "b2"))

        def y_of(g):                 # the input of rank g: Y for the first transform, else the previous x through the permutation
            if t == T_ - 1:
                return np.asarray(z, np.float64)[:, f_o_r[t][g]]
            return src[:, spec.orders[t + 1][f_o_r[t][g]]]

        def params(g, upto_tile, own_quads, H2):
            """bias + the rank's two private output tiles against h2 tiles < upto_tile (burst), tile upto_tile - 1 ... :
            here simply everything final plus the own tile's quads so far (the split is checked by the zero pattern below)"""
            out = b3i[g].copy()[None, :].repeat(n, 0)
            for half in range(2):
                for K in range(upto_tile + 1):
                    B = frag_block(f3i[g, half, K])
                    cols = np.arange(16) if K < upto_tile else own_quads
                    out[:, 16 * half:16 * half + 16] += H2[:, 16 * K + cols] @ B[:, cols].T
                    if K == upto_tile:                                      # what the chain does not add must be exact zeros
                        later = np.setdiff1d(np.arange(16), own_quads)
                        assert not B[:, later].any()
            return out[:, :23]

        x = np.zeros((n, Dp))
        H0, H1, H2 = (np.zeros((n, Hp)) for _ in range(3))
        x0, l0 = rqs_inverse(y_of(0), b3i[0][None, :23].repeat(n, 0))
        x[:, 0] = x0; ladj -= l0
        a0n = np.outer(x[:, 0], window(cw0[0])[:, 0])                       # rank 0 = k slot 0 of tile 0's window
        n1 = np.zeros((n, 16)); n2 = np.zeros((n, 16))                      # the previous tile's right-looking share
        for T in range(live_tiles):
            own = tg[T]
            qd = spec.quad_deg[4 * T:4 * T + 4]
            # burst wave: staging of tile T from what is final two tiles back
            a0 = b0[16 * T:16 * T + 16] + sum(x[:, 16 * X:16 * X + 16] @ frag_block(f0c[T, X]).T for X in range(nXT)) + a0n
            p1 = b1[16 * T:16 * T + 16] + sum(H0[:, 16 * K:16 * K + 16] @ frag_block(f1[T, K]).T for K in range(T - 1)) + n1
            p2 = b2[16 * T:16 * T + 16] + sum(H1[:, 16 * K:16 * K + 16] @ frag_block(f2[T, K]).T for K in range(T - 1)) + n2
            a0n = np.zeros((n, 16)); n1 = np.zeros((n, 16)); n2 = np.zeros((n, 16))
            W1d, W2d = frag_block(f1[T, T]), frag_block(f2[T, T])
            W1n = frag_block(f1[T + 1, T]) if T + 1 < nT else np.zeros((16, 16))
            W2n = frag_block(f2[T + 1, T]) if T + 1 < nT else np.zeros((16, 16))
            W0w = window(cw0[T])
            W0n = window(cw0[T + 1]) if T + 1 < nT else np.zeros((16, 16))
            done = np.zeros(0, dtype=int)
            for I, d in enumerate(own):
                quads = [j for j in range(4) if qd[j] == d]
                if I == len(own) - 1:                                       # the last live group absorbs the padding quads
                    quads += [j for j in range(4) if qd[j] >= D]
                sl_ = np.concatenate([np.arange(4 * j, 4 * j + 4) for j in quads])
                h0 = np.maximum(a0[:, sl_], 0.0); H0[:, 16 * T + sl_] = h0
                p1 = p1 + h0 @ W1d[:, sl_].T; n1 = n1 + h0 @ W1n[:, sl_].T
                h1 = np.maximum(p1[:, sl_] + h0, 0.0); H1[:, 16 * T + sl_] = h1
                p2 = p2 + h1 @ W2d[:, sl_].T; n2 = n2 + h1 @ W2n[:, sl_].T
                h2 = np.maximum(p2[:, sl_] + h1, 0.0); H2[:, 16 * T + sl_] = h2
                done = np.concatenate([done, sl_])
                phi = params(d, T, done, H2)
                xg, lg = rqs_inverse(y_of(d), phi)
                x[:, d] = xg; ladj -= lg
                a0 = a0 + np.outer(xg, W0w[:, 4 + I])                       # the later quads of this tile
                a0n = a0n + np.outer(xg, W0n[:, I])                         # the next tile
        src = x
    out = np.zeros((n, D))
    out[:, f_o_r[0]] = src[:, :D]                                           # the last transform's x, by feature
    return out, ladj


@pytest.mark.parametrize("D,T", [(3, 2), (4, 3), (5, 3), (10, 3), (17, 2), (32, 3), (40, 2), (50, 2)])
def test_spline_image_of_the_two_wave_sweep_reproduces_the_inverse(D, T):
    import numpy as np
    from oracle.maf import OracleMAF
    from pocomc_amd.maf_spec import MAFSpec
    spec = MAFSpec(D, T, univariate="rqs")
    if not spec.tri_ok:
        pytest.skip("degree groups wider than a tile")
    flat = (spec.init_params(5) * np.float32(1.3)).astype(np.float32)
    z = (np.random.default_rng(D).normal(size=(10, D)) * 1.5).astype(np.float32)
    xo, lo = OracleMAF(spec, flat).inverse(z)
    x, l = _nsf2_emulate(spec, flat, z)
    np.testing.assert_allclose(x, xo, rtol=5e-5, atol=5e-5)
    np.testing.assert_allclose(l, lo, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("D,T,H", [(16, 2, 64), (50, 3, 256), (128, 2, 512)])
def test_bf16_training_image_covers_every_unmasked_parameter_once(D, T, H):
    """``MAFSpec.wide_index`` (gather map of the row-major bf16 image of ``csrc/maf_train_bf16.hip`` and scatter map of
    its weight gradients): every unmasked weight sits once in its ``W`` part and once in its ``W^T`` part at the
    transposed position, every bias once, masked entries nowhere; the emulated forward pass through the image equals
    the canonical masked matrices."""
    from pocomc_amd.maf_spec import MAFSpec
    spec = MAFSpec(D, T, hidden=H)
    L = spec.wide_layout()
    img, bias = spec.wide_index()
    assert img.size == T * L["per_transform"] and bias.size == T * L["bias_per_transform"]
    m = spec.mask_flat()
    is_bias = np.zeros(spec.n_params, dtype=bool)
    for t in range(T):
        for k in ("b0", "b1", "b2", "b3"):
            o, sz = spec.offsets[k]
            is_bias[t * spec.params_per_transform + o: t * spec.params_per_transform + o + sz] = True
    cnt = np.bincount(img[img >= 0], minlength=spec.n_params)
    cb = np.bincount(bias[bias >= 0], minlength=spec.n_params)
    assert np.all(cnt[(m > 0) & ~is_bias] == 2) and np.all(cnt[(m == 0) | is_bias] == 0)
    assert np.all(cb[is_bias] == 1) and np.all(cb[~is_bias] == 0)
    DK, HK, OK = L["DK"], L["HK"], L["OK"]
    flat = spec.init_params(1)
    vals = np.where(img >= 0, flat[np.maximum(img, 0)], 0.0)
    bv = np.where(bias >= 0, flat[np.maximum(bias, 0)], 0.0)
    su = np.full(HK, -1)
    su[:spec.Hp] = spec.slot_unit
    live = su >= 0
    rng = np.random.default_rng(0)
    x = rng.normal(size=D).astype(np.float32)
    for t in range(T):
        o = t * L["per_transform"]
        W0f = vals[o:o + HK * DK].reshape(HK, DK); o += HK * DK
        W0b = vals[o:o + DK * HK].reshape(DK, HK); o += DK * HK
        W1f = vals[o:o + HK * HK].reshape(HK, HK); o += HK * HK
        W1b = vals[o:o + HK * HK].reshape(HK, HK); o += 2 * HK * HK + HK * HK      # (skip W2f, W2b)
        W3f = vals[o:o + OK * HK].reshape(OK, HK); o += OK * HK
        W3b = vals[o:o + HK * OK].reshape(HK, OK)
        assert np.array_equal(W0b, W0f.T) and np.array_equal(W1b, W1f.T) and np.array_equal(W3b, W3f.T)
        M0, M1, M2, M3 = spec.masks(t)
        W0 = spec.view(flat, t, "W0") * M0
        b0 = spec.view(flat, t, "b0")
        h_img = W0f[:, :D] @ x + bv[t * L["bias_per_transform"]: t * L["bias_per_transform"] + HK]
        h_ref = W0 @ x + b0
        np.testing.assert_allclose(h_img[live], h_ref[su[live]], rtol=1e-6, atol=1e-6)
        assert np.all(h_img[~live] == 0.0)
        W3 = spec.view(flat, t, "W3") * M3                                 # rows 2 * feature + s
        hh = rng.normal(size=H).astype(np.float32)
        hs = np.zeros(HK, dtype=np.float32)
        hs[live] = hh[su[live]]
        np.testing.assert_allclose((W3f @ hs)[:2 * D], W3 @ hh, rtol=1e-5, atol=1e-5)


def test_host_prefetcher_runs_a_job_and_shuts_down():
    """``pmc_prefetcher_*`` (host code only): a helper thread waits for a completion word and reads a buffer; the call
    sequence create / submit / flag / destroy returns, nothing is written to the buffer."""
    import ctypes as C
    import time
    from pocomc_amd import _lib
    lib = _lib.load()                       # (loading and host-only entry points need no GPU)
    h = lib.pmc_prefetcher_create(2, None)
    assert h
    flag = np.zeros(1, dtype=np.int64)
    buf = np.arange(100003, dtype=np.float64)
    ref = buf.copy()
    for step in range(1, 4):
        assert lib.pmc_prefetcher_submit(h, flag.ctypes.data, step, buf.ctypes.data, buf.nbytes, 0.5) == 0
        flag[0] = step
    assert lib.pmc_prefetcher_submit(h, flag.ctypes.data, 99, buf.ctypes.data, buf.nbytes, 0.05) == 0    # never signalled: times out
    time.sleep(0.2)
    lib.pmc_prefetcher_destroy(h)
    assert np.array_equal(buf, ref)
    assert lib.pmc_prefetcher_submit(None, flag.ctypes.data, 1, buf.ctypes.data, 8, 0.1) != 0


def test_proposal_draws_of_the_16bit_guard_follow_the_proposal_law():
    """``mcmc._proposal_draws`` (the points the 16-bit sweep's guard is run on at the head of a kernel call): tpCN draws
    have the proposal's mean ``mu + sqrt(1 - sigma^2) (theta - mu)`` and Student-t tails (``mcmc.py:77-85``), RWM draws the
    covariance ``sigma^2 Sigma`` around the walker (``:251-253``); a strided sample over all walkers, no global RNG touched."""
    import torch
    from types import SimpleNamespace
    from pocomc_amd.mcmc import _proposal_draws
    rng = np.random.default_rng(0)
    D, n = 5, 20000
    A = rng.normal(size=(D, D))
    cov = A @ A.T + D * np.eye(D)
    mu = rng.normal(size=D)
    theta = torch.from_numpy((mu + rng.normal(size=(n, D)) @ np.linalg.cholesky(cov).T).astype(np.float32))
    geo = SimpleNamespace(t_mean=mu, t_cov=cov, t_nu=5.0, normal_cov=cov)
    state = np.random.get_state()[1].copy()
    t_state = torch.get_rng_state()
    sigma = 0.6
    thp = _proposal_draws("preconditioned_pcn", theta, geo, sigma, 8192).numpy().astype(np.float64)
    assert thp.shape == (8192, D) and np.isfinite(thp).all()
    idx = np.linspace(0, n - 1, 8192).astype(np.int64)
    resid = thp - (mu + np.sqrt(1 - sigma ** 2) * (theta.numpy()[idx] - mu))
    assert np.abs(resid.mean(axis=0)).max() < 0.25
    white = resid @ np.linalg.inv(np.linalg.cholesky(cov)).T / sigma
    kurt = (white ** 4).mean() / (white ** 2).mean() ** 2
    assert kurt > 3.5, kurt                                   # heavier than Gaussian: the Student-t scale mixture
    rw = _proposal_draws("preconditioned_rwm", theta, geo, sigma, 8192).numpy().astype(np.float64) - theta.numpy()[idx]
    np.testing.assert_allclose(np.cov(rw.T), sigma ** 2 * cov, rtol=0.15, atol=0.15)
    assert np.array_equal(np.random.get_state()[1], state) and torch.equal(torch.get_rng_state(), t_state)
