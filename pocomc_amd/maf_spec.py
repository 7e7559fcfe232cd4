"""Architecture of the masked autoregressive flow (MAF) used on the hot path.

This is host logic only (numpy): mask construction, the canonical parameter
layout and the index maps that pack canonical parameters into the layouts the
gfx950 kernels consume.  No arithmetic of the flow itself lives here.

What it restates
----------------
``pocomc/flow.py:46-68`` builds ``zuko.flows.MAF(n_dim, transforms=T,
hidden_features=[H]*3, residual=True)`` with ``H = max(next_pow2(3*n_dim), 32)``
(``pocomc/flow.py:49-52``).  ``zuko`` is a third-party package that is NOT under
``/root/reference`` (pinned only as ``zuko>=1.1.0`` in ``requirements.txt:3``),
so the architecture below is a restatement of zuko's published MAF, not of a
file in the reference (*parity unpinned*):

* ``T`` masked-autoregressive transforms; variable order alternates identity /
  reversed per transform; base distribution is a diagonal ``N(0, I)``.
* each transform: a masked MLP ``D -> H -> H -> H -> 2D`` with ReLU; the two
  ``H -> H`` layers carry a skip connection (``h' = relu(h + W h + b)``);
  outputs are ``(shift_j, raw_j)`` per feature ``j`` (row ``2j`` / ``2j+1``).
* univariate map ``y = x * exp(ls) + shift`` with the soft-clipped log-scale
  ``ls = raw / (1 + |raw / log(1e-3)|)``; ``ladj = sum_j ls_j``.
* masks (zuko ``MaskedMLP``): hidden unit ``k`` has degree
  ``deg(k) = 1 + k mod (D-1)``; it reads inputs of rank ``< deg(k)`` and hidden
  units of degree ``<= deg(k)``; the outputs of the feature of rank ``r`` read
  hidden units of degree ``<= r`` (rank 0 reads nothing: bias only).

Canonical parameter layout (one flat fp32 vector, transform after transform):
``W0[H,D] b0[H] W1[H,H] b1[H] W2[H,H] b2[H] W3[2D,H] b3[2D]`` with torch
``nn.Linear`` convention ``weight[out, in]``.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

LOG_SLOPE = math.log(1e-3)  # zuko MonotonicAffineTransform slope=1e-3


def next_power_of_2(n: int) -> int:
    """``pocomc/flow.py:49-50``."""
    return 1 if n == 0 else 2 ** (int(n) - 1).bit_length()


def default_hidden(n_dim: int) -> int:
    """``pocomc/flow.py:52``: ``np.maximum(next_power_of_2(3*n_dim), 32)``."""
    return max(next_power_of_2(3 * n_dim), 32)


def _ceil_to(x: int, m: int) -> int:
    return (x + m - 1) // m * m


@dataclass
class MAFSpec:
    """Static description of one MAF: shapes, masks, packing maps."""

    n_dim: int
    n_transforms: int = 3
    hidden: int | None = None
    univariate: str = "affine"      # "affine" (MAF) | "rqs" (NSF: monotonic rational-quadratic spline)
    bins: int = 8                   # spline bins (pocomc/flow.py:71), ignored for "affine"

    # filled by __post_init__
    orders: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        D = int(self.n_dim)
        if D < 2:
            # zuko raises "The adjacency matrix leads to a null Jacobian." for D=1
            raise ValueError("MAF needs n_dim >= 2 (the adjacency matrix of a 1-D "
                             "autoregressive transform leads to a null Jacobian)")
        self.n_dim = D
        self.n_transforms = int(self.n_transforms)
        if self.hidden is None:
            self.hidden = default_hidden(D)
        self.hidden = int(self.hidden)
        H = self.hidden
        if H < D - 1:
            raise NotImplementedError("hidden width must be >= n_dim - 1")
        if self.univariate not in ("affine", "rqs"):
            raise ValueError("univariate must be 'affine' or 'rqs'")
        self.bins = int(self.bins)
        if self.univariate == "rqs" and self.bins not in (4, 8, 16):
            # (the reference builds 8 bins, pocomc/flow.py:71,77,83, and accepts any zuko flow object, :87-88; the element
            #  code is instantiated for these three counts -- csrc/rqs.h -- and only the default 8 has the triangular
            #  inverse sweeps: the others invert with zuko's own D-pass algorithm)
            raise NotImplementedError("spline flows are built for bins in (4, 8, 16)")
        # hyper-network outputs per feature: (shift, raw log-scale) or (K widths, K heights, K-1 derivatives)
        self.n_out = 2 if self.univariate == "affine" else 3 * int(self.bins) - 1
        NO = self.n_out
        # rank of every input feature, per transform (zuko MAF: orders[i % 2])
        ident = np.arange(D)
        self.orders = [ident.copy() if t % 2 == 0 else ident[::-1].copy()
                       for t in range(self.n_transforms)]
        # degree of every hidden unit (same for the three hidden layers)
        self.degree = 1 + (np.arange(H) % (D - 1))
        # canonical offsets inside one transform
        sizes = [("W0", H * D), ("b0", H), ("W1", H * H), ("b1", H),
                 ("W2", H * H), ("b2", H), ("W3", NO * D * H), ("b3", NO * D)]
        off = 0
        self.offsets = {}
        for name, sz in sizes:
            self.offsets[name] = (off, sz)
            off += sz
        self.params_per_transform = off
        self.n_params = off * self.n_transforms
        self._build_device_layout()

    # ------------------------------------------------------------------ masks
    def masks(self, t: int):
        """Boolean masks ``(M0[H,D], M1[H,H], M2[H,H], M3[2D,H])`` of transform t."""
        D, H = self.n_dim, self.hidden
        rank = self.orders[t]                      # rank[j] of feature j
        deg = self.degree
        M0 = rank[None, :] < deg[:, None]
        M1 = deg[None, :] <= deg[:, None]
        M3 = deg[None, :] <= np.repeat(rank, self.n_out)[:, None]
        return M0, M1, M1.copy(), M3

    def shapes(self):
        D, H = self.n_dim, self.hidden
        return {"W0": (H, D), "b0": (H,), "W1": (H, H), "b1": (H,),
                "W2": (H, H), "b2": (H,), "W3": (self.n_out * D, H), "b3": (self.n_out * D,)}

    def view(self, flat: np.ndarray, t: int, name: str) -> np.ndarray:
        """View of one canonical tensor inside the flat parameter vector."""
        off, sz = self.offsets[name]
        base = t * self.params_per_transform + off
        return flat[base:base + sz].reshape(self.shapes()[name])

    def mask_flat(self) -> np.ndarray:
        """0/1 float mask over the flat parameter vector (1 for biases)."""
        m = np.ones(self.n_params, dtype=np.float32)
        for t in range(self.n_transforms):
            M0, M1, M2, M3 = self.masks(t)
            for name, M in (("W0", M0), ("W1", M1), ("W2", M2), ("W3", M3)):
                self.view(m, t, name)[...] = M.astype(np.float32)
        return m

    def init_params(self, seed: int | None = None) -> np.ndarray:
        """``nn.Linear`` default init: ``U(-1/sqrt(fan_in), 1/sqrt(fan_in))`` for
        weights and biases (fan_in = unmasked ``in_features``)."""
        rng = np.random.default_rng(seed)
        flat = np.empty(self.n_params, dtype=np.float32)
        fan = {"W0": self.n_dim, "b0": self.n_dim, "W1": self.hidden, "b1": self.hidden,
               "W2": self.hidden, "b2": self.hidden, "W3": self.hidden, "b3": self.hidden}
        for t in range(self.n_transforms):
            for name in self.offsets:
                v = self.view(flat, t, name)
                b = 1.0 / math.sqrt(fan[name])
                v[...] = rng.uniform(-b, b, size=v.shape).astype(np.float32)
        return flat

    # ---------------------------------------------------------- device layout
    def _build_device_layout(self):
        """Slot layout of the gfx950 kernels.

        Hidden units are sorted by degree; every degree group is padded to a
        multiple of 4 slots (one *quad* = the K of one ``mfma_f32_16x16x4f32``)
        and the total to a multiple of 16 (one *tile* = 4 quads = the M of the
        same MFMA).  Features are addressed by rank, padded to ``Dp`` (multiple
        of 16).  Output rows are ``2*rank`` (shift) and ``2*rank+1`` (raw),
        padded to ``Op = 2*Dp``.
        """
        D, H = self.n_dim, self.hidden
        deg = self.degree
        slots = []          # canonical unit index per slot, -1 = padding
        sdeg = []           # degree per slot (D = padding, never reached)
        tri_ok = True
        for g in range(1, D):
            units = np.nonzero(deg == g)[0]
            nq = _ceil_to(len(units), 4) // 4
            if nq <= 4:
                # a degree group never straddles a tile boundary: its units all
                # read each other's previous-layer values, so the sweep of the
                # triangular inverse finishes a group inside one tile
                in_tile = (len(slots) // 4) % 4
                if in_tile + nq > 4:
                    fill = (4 - in_tile) * 4
                    slots += [-1] * fill
                    sdeg += [D] * fill
            else:
                tri_ok = False
            pad = nq * 4 - len(units)
            slots += list(units) + [-1] * pad
            sdeg += [g] * (len(units) + pad)
        self.tri_ok = tri_ok
        Hp = _ceil_to(len(slots), 16)
        slots += [-1] * (Hp - len(slots))
        sdeg += [D] * (Hp - len(sdeg))          # trailing pad: never reached
        self.Hp = Hp
        self.nT = Hp // 16                       # hidden tiles
        self.nQ = Hp // 4                        # hidden quads
        self.Dp = _ceil_to(D, 16)
        self.nXT = self.Dp // 16                 # input (rank) tiles
        self.Op = _ceil_to(self.n_out * self.Dp, 16)   # output rows n_out*rank + j
        self.nOT = self.Op // 16                 # output tiles (8 ranks each for the affine map)
        self.slot_unit = np.asarray(slots, dtype=np.int64)
        self.slot_deg = np.asarray(sdeg, dtype=np.int32)
        qdeg = self.slot_deg.reshape(-1, 4)
        assert (qdeg == qdeg[:, :1]).all()
        qdeg = qdeg[:, 0].copy()
        # a quad is "last" of its degree if the next quad has a different degree
        qlast = np.ones(self.nQ, dtype=np.int32)
        qlast[:-1] = (qdeg[1:] != qdeg[:-1]).astype(np.int32)
        qlast[qdeg >= D] = 0                     # padding quads trigger nothing
        self.quad_deg = qdeg
        self.quad_last = qlast
        # quad meta word: degree | last << 16
        self.quad_meta = (qdeg.astype(np.int32) | (qlast << 16)).astype(np.int32)

        # ---- sizes of the packed arrays (floats), per transform
        nT, nXT, nOT = self.nT, self.nXT, self.nOT
        self.sz_f0 = nT * nXT * 256              # W0 fragments  [tile][xtile][lane][4]
        self.sz_f12 = nT * nT * 256              # W1/W2 fragments [tile][ktile][lane][4]
        self.sz_f3 = nOT * nT * 256              # W3 fragments  [otile][ktile][lane][4]
        self.sz_w0n = self.Dp * Hp               # W0 natural    [rank][slot]
        self.sz_b = Hp                           # b0,b1,b2      [slot]
        self.sz_b3 = self.Op                     # b3            [2*rank + {0,1}]
        parts = [("f0", self.sz_f0), ("f1", self.sz_f12), ("f2", self.sz_f12),
                 ("f3", self.sz_f3), ("w0n", self.sz_w0n), ("b0", self.sz_b),
                 ("b1", self.sz_b), ("b2", self.sz_b), ("b3", self.sz_b3)]
        if self.univariate == "rqs" and self.bins == 8:
            # inverse sweep: the 23 output rows of every rank padded to two 16-row tiles of their own
            self.sz_f3i = D * 2 * nT * 256       # [rank][half][ktile][lane][4]
            self.sz_b3i = D * 32                 # [rank][32]
            parts += [("f3i", self.sz_f3i), ("b3i", self.sz_b3i)]
            # the two-wave spline sweep (csrc/maf_inverse_nsf2.hip) shares the affine two-wave sweep's hidden part: the
            # layer-0 window columns (cw0), the masked layer-0 fragments (f0c), the transposed biases -- see below
            parts += [("cw0", nT * 256), ("f0c", self.sz_f0), ("b0t", Hp), ("b1t", Hp), ("b2t", Hp)]
        elif self.univariate == "affine":
            # chain image of the lane-per-walker inverse sweep (csrc/maf_inverse_tri6.hip): A operands of
            # v_mfma_f32_4x4x1_16b_f32 -- lane l holds W[out quad row l & 3][k slot l >> 2], one VGPR = a 4 x 16 block --
            # for the diagonal tile of the hidden layers (cw1, cw2: [tile][lane][out quad]), the layer-0 columns of the
            # ranks the previous and the own tile produce (cw0: k slots 0-3 / 4-7) and the output rows of the own
            # tile's ranks (cw3: [tile][lane][pair of groups]); f0c = f0 with the columns of those ranks zeroed (what
            # the helper wavefront multiplies while the chain adds the newest ranks from registers)
            parts += [("cw1", nT * 256), ("cw2", nT * 256), ("cw0", nT * 256), ("cw3", nT * 128),
                      ("f0c", self.sz_f0)]
            # the hidden layers' biases with the rows of every tile transposed (slot 16 T + 4 q + r holds the bias of unit
            # 16 T + 4 r + q): what a lane of the two-wave sweep's transposed accumulators (csrc/maf_chain_rot.h) starts
            # from, one 16-byte load per tile
            parts += [("b0t", Hp), ("b1t", Hp), ("b2t", Hp)]
        off = 0
        self.pk_offsets = {}
        for name, sz in parts:
            self.pk_offsets[name] = off
            off += sz
        self.pk_per_transform = off
        self.pk_size = off * self.n_transforms

    def pack_index(self) -> np.ndarray:
        """int32 gather map: ``packed[i] = flat[idx[i]] if idx[i] >= 0 else 0``.

        Masked-out weights map to -1 (exact zeros in the packed image), so the
        kernels never need the masks.
        """
        D, H, Hp, Dp = self.n_dim, self.hidden, self.Hp, self.Dp
        nT, nXT, nOT = self.nT, self.nXT, self.nOT
        idx = np.full(self.pk_size, -1, dtype=np.int64)
        lane = np.arange(64)
        lk, li = lane >> 4, lane & 15            # k within quad, row within tile
        for t in range(self.n_transforms):
            base_c = t * self.params_per_transform
            base_p = t * self.pk_per_transform
            rank = self.orders[t]
            feat_of_rank = np.argsort(rank)
            M0, M1, M2, M3 = self.masks(t)
            su = self.slot_unit

            def cidx(name, row, col, ncol, mask):
                """canonical flat index of weight[row, col] or -1."""
                off, _ = self.offsets[name]
                ok = (row >= 0) & (col >= 0)
                r = np.where(ok, row, 0)
                c = np.where(ok, col, 0)
                ok = ok & mask[r, c]
                return np.where(ok, base_c + off + r * ncol + c, -1)

            # --- W0 fragments: A[i][k] = W0[out slot 16*T+i][in rank 16*X + 4c + k]
            f0 = np.full((nT, nXT, 64, 4), -1, dtype=np.int64)
            for T in range(nT):
                out_unit = su[16 * T + li]                       # (64,)
                for X in range(nXT):
                    for c in range(4):
                        r_in = 16 * X + 4 * c + lk
                        feat = np.where(r_in < D, feat_of_rank[np.minimum(r_in, D - 1)], -1)
                        f0[T, X, :, c] = cidx("W0", out_unit, feat, D, M0)
            # --- W1/W2 fragments: A[i][k] = W[out slot 16*T+i][in slot 16*K + 4c + k]
            f12 = {}
            for name, M in (("W1", M1), ("W2", M2)):
                f = np.full((nT, nT, 64, 4), -1, dtype=np.int64)
                for T in range(nT):
                    out_unit = su[16 * T + li]
                    for K in range(nT):
                        for c in range(4):
                            in_unit = su[16 * K + 4 * c + lk]
                            f[T, K, :, c] = cidx(name, out_unit, in_unit, H, M)
                f12[name] = f
            # --- W3 fragments: A[i][k] = W3[out row of (rank, s)][in slot]
            f3 = np.full((nOT, nT, 64, 4), -1, dtype=np.int64)
            for O in range(nOT):
                orow = 16 * O + li                               # packed out row
                r_out = orow // self.n_out
                s = orow % self.n_out
                crow = np.where(r_out < D, self.n_out * feat_of_rank[np.minimum(r_out, D - 1)] + s, -1)
                for K in range(nT):
                    for c in range(4):
                        in_unit = su[16 * K + 4 * c + lk]
                        f3[O, K, :, c] = cidx("W3", crow, in_unit, H, M3)
            # --- W0 natural [rank][slot]
            w0n = np.full((Dp, Hp), -1, dtype=np.int64)
            for r in range(D):
                feat = np.full(Hp, feat_of_rank[r])
                w0n[r] = cidx("W0", su, feat, D, M0)

            def bidx(name):
                off, _ = self.offsets[name]
                return np.where(su >= 0, base_c + off + np.maximum(su, 0), -1)

            b3 = np.full(self.Op, -1, dtype=np.int64)
            off3, _ = self.offsets["b3"]
            for r in range(D):
                for j in range(self.n_out):
                    b3[self.n_out * r + j] = base_c + off3 + self.n_out * feat_of_rank[r] + j

            po = self.pk_offsets
            def put(name, arr):
                a = arr.reshape(-1)
                idx[base_p + po[name]: base_p + po[name] + a.size] = a
            put("f0", f0); put("f1", f12["W1"]); put("f2", f12["W2"]); put("f3", f3)
            put("w0n", w0n); put("b0", bidx("b0")); put("b1", bidx("b1")); put("b2", bidx("b2"))
            put("b3", b3)
            if self.univariate == "affine" or self.bins == 8:   # the sweeps' images (cw1 / cw2 / cw3: the affine one only)
                affine = self.univariate == "affine"
                ci, ck = lane & 3, lane >> 2                # 4x4x1 A operand: row within the out quad, k slot
                tg = self.tile_groups()                      # per tile: degrees of its groups (in order)
                cw1 = np.full((nT, 64, 4), -1, dtype=np.int64)
                cw2 = np.full((nT, 64, 4), -1, dtype=np.int64)
                cw0 = np.full((nT, 64, 4), -1, dtype=np.int64)
                cw3 = np.full((nT, 64, 2), -1, dtype=np.int64)
                f0c = np.full((nT, nXT, 64, 4), -1, dtype=np.int64)
                for T in range(nT):
                    in_unit = su[16 * T + ck]
                    prev = [0] if T == 0 else tg[T - 1]      # ranks produced before this tile's chain starts its own
                    own = tg[T]
                    win = np.full(16, -1, dtype=np.int64)    # rank of every k slot of the layer-0 window
                    win[:len(prev)] = prev
                    win[4:4 + len(own)] = own
                    wfeat = np.where(win >= 0, feat_of_rank[np.clip(win, 0, D - 1)], -1)
                    for a in range(4):
                        out_unit = su[16 * T + 4 * a + ci]
                        cw1[T, :, a] = cidx("W1", out_unit, in_unit, H, M1)
                        cw2[T, :, a] = cidx("W2", out_unit, in_unit, H, M2)
                        cw0[T, :, a] = cidx("W0", out_unit, wfeat[ck], D, M0)
                    for sl in range(2):
                        gi = 2 * sl + (ci >> 1)              # group of the tile this row belongs to
                        rk = np.array([own[g] if g < len(own) else -1 for g in gi])
                        crow = np.where(rk >= 0, self.n_out * feat_of_rank[np.clip(rk, 0, D - 1)] + (ci & 1), -1)
                        cw3[T, :, sl] = cidx("W3", crow, in_unit, H, M3)
                    cut = 0 if T == 0 else (prev[0] if len(prev) else D)   # first rank the chain adds itself
                    for X in range(nXT):
                        for c in range(4):
                            r_in = 16 * X + 4 * c + lk
                            f0c[T, X, :, c] = np.where(r_in < cut, f0[T, X, :, c], -1)
                if affine:
                    put("cw1", cw1); put("cw2", cw2); put("cw3", cw3)
                put("cw0", cw0); put("f0c", f0c)
                tr = (np.arange(Hp) & ~15) + 4 * (np.arange(Hp) & 3) + ((np.arange(Hp) >> 2) & 3)
                for name in ("b0", "b1", "b2"):
                    put(name + "t", bidx(name)[tr])
            if self.univariate == "rqs" and self.bins == 8:
                NO = self.n_out
                f3i = np.full((D, 2, nT, 64, 4), -1, dtype=np.int64)
                b3i = np.full((D, 32), -1, dtype=np.int64)
                for r in range(D):
                    feat = feat_of_rank[r]
                    for half in range(2):
                        j = 16 * half + li                           # output j of this rank held by row li
                        crow = np.where(j < NO, NO * feat + np.minimum(j, NO - 1), -1)
                        for K in range(nT):
                            for c in range(4):
                                in_unit = su[16 * K + 4 * c + lk]
                                f3i[r, half, K, :, c] = cidx("W3", crow, in_unit, H, M3)
                    b3i[r, :NO] = base_c + off3 + NO * feat + np.arange(NO)
                put("f3i", f3i); put("b3i", b3i)
        return idx.astype(np.int32)

    def tile_groups(self):
        """Per hidden tile: the degrees (= the ranks they produce) of its degree groups, in slot order."""
        out = []
        for T in range(self.nT):
            dq = self.quad_deg[4 * T:4 * T + 4]
            g = []
            for d in dq:
                if d < self.n_dim and (not g or g[-1] != d):
                    g.append(int(d))
            out.append(g)
        return out

    def device_meta(self) -> np.ndarray:
        """int32 metadata consumed by the kernels.

        ``[0:8]``  header: D, H, T, Hp, Dp, nT, pk_per_transform, live hidden tiles
        ``[8:8+T*D]``        feature index of every rank, per transform
        ``[8+T*D:8+2*T*D]``  rank of every feature, per transform
        ``[8+2*T*D: +nQ]``   quad meta words (degree | last<<16), shared by all transforms
        """
        D, T = self.n_dim, self.n_transforms
        live = int(np.sum((self.quad_deg.reshape(-1, 4) < D).any(axis=1)))      # hidden tiles with a degree group (the padding tiles trail)
        hdr = np.array([D, self.hidden, T, self.Hp, self.Dp, self.nT,
                        self.pk_per_transform, live], dtype=np.int32)
        f_o_r = np.concatenate([np.argsort(o) for o in self.orders]).astype(np.int32)
        r_o_f = np.concatenate(self.orders).astype(np.int32)
        return np.concatenate([hdr, f_o_r, r_o_f, self.quad_meta]).astype(np.int32)

    # ------------------------------------------------------- bf16 forward image
    def bf16_layout(self):
        """Sizes / offsets (in bf16 elements, per transform) of the weight fragments of the bf16 forward kernel
        (``csrc/maf_forward_bf16.hip``): A operands of ``v_mfma_f32_16x16x32_bf16`` -- lane l holds row ``l & 15`` of the
        16-row out tile and the 8 consecutive k ``8 (l >> 4) .. + 7`` of a 32-wide k tile --
        ``g0 [nT][nX2]``, ``g1 / g2 [nT][nK2]``, ``g3 [nOT][nK2]``, each ``[64 lanes][8]``; biases stay float32 (they are
        read from the float32 image)."""
        nX2, nK2 = -(-self.Dp // 32), -(-self.Hp // 32)
        sz = {"g0": self.nT * nX2 * 512, "g1": self.nT * nK2 * 512, "g2": self.nT * nK2 * 512, "g3": self.nOT * nK2 * 512}
        off, o = {}, 0
        for k, v in sz.items():
            off[k] = o
            o += v
        return dict(nX2=nX2, nK2=nK2, sz=sz, off=off, per_transform=o)

    def pack_index_bf16(self) -> np.ndarray:
        """int32 gather map of the bf16 fragment image: ``image[i] = bf16(flat[idx[i]]) if idx[i] >= 0 else 0``."""
        D, H = self.n_dim, self.hidden
        L = self.bf16_layout()
        nX2, nK2, nT, nOT = L["nX2"], L["nK2"], self.nT, self.nOT
        idx = np.full(L["per_transform"] * self.n_transforms, -1, dtype=np.int64)
        lane = np.arange(64)
        li, lg = lane & 15, lane >> 4
        su = self.slot_unit
        for t in range(self.n_transforms):
            base_c = t * self.params_per_transform
            rank = self.orders[t]
            feat_of_rank = np.argsort(rank)
            M0, M1, M2, M3 = self.masks(t)

            def cidx(name, row, col, ncol, mask):
                off, _ = self.offsets[name]
                ok = (row >= 0) & (col >= 0)
                r = np.where(ok, row, 0)
                c = np.where(ok, col, 0)
                ok = ok & mask[r, c]
                return np.where(ok, base_c + off + r * ncol + c, -1)

            def in_slot(k):                       # canonical hidden unit of packed slot k (or -1)
                return np.where(k < self.Hp, su[np.minimum(k, self.Hp - 1)], -1)

            g0 = np.full((nT, nX2, 64, 8), -1, dtype=np.int64)
            g1 = np.full((nT, nK2, 64, 8), -1, dtype=np.int64)
            g2 = np.full((nT, nK2, 64, 8), -1, dtype=np.int64)
            g3 = np.full((nOT, nK2, 64, 8), -1, dtype=np.int64)
            for T in range(nT):
                out_unit = su[16 * T + li]
                for X in range(nX2):
                    for e in range(8):
                        r_in = 32 * X + 8 * lg + e
                        feat = np.where(r_in < D, feat_of_rank[np.minimum(r_in, D - 1)], -1)
                        g0[T, X, :, e] = cidx("W0", out_unit, feat, D, M0)
                for K in range(nK2):
                    for e in range(8):
                        iu = in_slot(32 * K + 8 * lg + e)
                        g1[T, K, :, e] = cidx("W1", out_unit, iu, H, M1)
                        g2[T, K, :, e] = cidx("W2", out_unit, iu, H, M2)
            for O in range(nOT):
                orow = 16 * O + li
                r_out, sft = orow // self.n_out, orow % self.n_out
                crow = np.where(r_out < D, self.n_out * feat_of_rank[np.minimum(r_out, D - 1)] + sft, -1)
                for K in range(nK2):
                    for e in range(8):
                        g3[O, K, :, e] = cidx("W3", crow, in_slot(32 * K + 8 * lg + e), H, M3)
            b = t * L["per_transform"]
            for name, arr in (("g0", g0), ("g1", g1), ("g2", g2), ("g3", g3)):
                a = arr.reshape(-1)
                idx[b + L["off"][name]: b + L["off"][name] + a.size] = a
        return idx.astype(np.int32)

    # ------------------------------------------- bf16 training image (wide flows)
    def wide_layout(self):
        """Sizes of the row-major bf16 weight image of ``csrc/maf_train_bf16.hip`` (elements per transform):
        ``W0f [HK][DK]  W0b = W0f^T  W1f [HK][HK]  W1b  W2f  W2b  W3f [OK][HK]  W3b`` and of its float32 bias image
        ``b0 b1 b2 [HK]  b3 [OK]``; hidden units in slot order, features in canonical order, W3 rows ``2 feature + s``."""
        DK, HK = _ceil_to(self.n_dim, 32), _ceil_to(self.Hp, 32)
        OK = 2 * DK
        return dict(DK=DK, HK=HK, OK=OK, per_transform=2 * (HK * DK + 2 * HK * HK + OK * HK),
                    bias_per_transform=3 * HK + OK)

    def wide_index(self):
        """``(image_idx, bias_idx)`` int32: canonical index of every element of the two images, -1 where masked or
        padded.  ``image[i] = bf16(flat[idx[i]])``; the ``W?f`` parts double as the scatter map of the weight gradients."""
        if self.univariate != "affine":
            raise NotImplementedError("the bf16 training image is built for the affine flows")
        D, H = self.n_dim, self.hidden
        L = self.wide_layout()
        DK, HK, OK = L["DK"], L["HK"], L["OK"]
        su = np.full(HK, -1, dtype=np.int64)
        su[:self.Hp] = self.slot_unit
        hv = su >= 0
        u = np.where(hv, su, 0)
        feat = np.arange(DK)
        fv = feat < D
        fc = np.where(fv, feat, 0)
        orow = np.arange(OK)                                  # 2 feature + s
        ov = (orow // 2) < D
        oc = np.where(ov, orow, 0)
        img = np.full(self.n_transforms * L["per_transform"], -1, dtype=np.int64)
        bias = np.full(self.n_transforms * L["bias_per_transform"], -1, dtype=np.int64)
        for t in range(self.n_transforms):
            base = t * self.params_per_transform
            M0, M1, M2, M3 = self.masks(t)
            off = {k: base + v[0] for k, v in self.offsets.items()}
            w0 = np.where(hv[:, None] & fv[None, :] & M0[u][:, fc], off["W0"] + u[:, None] * D + fc[None, :], -1)
            w1 = np.where(hv[:, None] & hv[None, :] & M1[u][:, u], off["W1"] + u[:, None] * H + u[None, :], -1)
            w2 = np.where(hv[:, None] & hv[None, :] & M2[u][:, u], off["W2"] + u[:, None] * H + u[None, :], -1)
            w3 = np.where(ov[:, None] & hv[None, :] & M3[oc][:, u], off["W3"] + oc[:, None] * H + u[None, :], -1)
            parts = [w0, w0.T, w1, w1.T, w2, w2.T, w3, w3.T]
            o = t * L["per_transform"]
            for a in parts:
                img[o:o + a.size] = np.ascontiguousarray(a).reshape(-1)
                o += a.size
            ob = t * L["bias_per_transform"]
            for k, name in enumerate(("b0", "b1", "b2")):
                bias[ob + k * HK: ob + (k + 1) * HK] = np.where(hv, off[name] + u, -1)
            bias[ob + 3 * HK: ob + 3 * HK + OK] = np.where(ov, off["b3"] + oc, -1)
        return img.astype(np.int32), bias.astype(np.int32)

    # ------------------------------------------------------------- training
    def train_layout(self):
        """Sizes of the training-side device arrays, per transform.

        ``packedT`` (float): transposed weight fragments for the data-gradient products
        ``dh = W^T da``:  f0T [nXT][nT], f1T/f2T [nT][nT], f3T [nT][nOT], each ``[..][..][lane][4]``
        with ``A[i][k] = W[out 16*To + 4c + k][in 16*Ti + i]``.
        ``gmap`` (int32): canonical index of every element of a weight-gradient tile in MFMA C
        layout (lane (q, j) reg r <-> W[out 16*To + 4q + r][in 16*Ti + j]), -1 = masked / padding:
        g0 [nT][nXT], g1/g2 [nT][nT], g3 [nOT][nT], then the bias maps b0,b1,b2 [Hp], b3 [Op].
        """
        nT, nXT, nOT = self.nT, self.nXT, self.nOT
        szT = {"f0T": nXT * nT * 256, "f1T": nT * nT * 256, "f2T": nT * nT * 256, "f3T": nT * nOT * 256}
        szG = {"g0": nT * nXT * 256, "g1": nT * nT * 256, "g2": nT * nT * 256, "g3": nOT * nT * 256,
               "gb0": self.Hp, "gb1": self.Hp, "gb2": self.Hp, "gb3": self.Op}
        offT, o = {}, 0
        for k, v in szT.items():
            offT[k] = o
            o += v
        pkT = o
        offG, o = {}, 0
        for k, v in szG.items():
            offG[k] = o
            o += v
        return dict(szT=szT, offT=offT, pkT_per_transform=pkT, szG=szG, offG=offG, gmap_per_transform=o)

    def par_per_transform(self) -> int:
        """Floats of the hyper-network's outputs kept per row set and transform (``pmc_maf_train_t.par_scratch``):
        ``nOT`` tiles for the affine flows, ``nXT`` panels of ``n_out`` (= 3 bins - 1) tiles for the spline flows."""
        return 256 * (self.nXT * self.n_out if self.univariate == "rqs" else self.nOT)

    def train_tables(self, n_waves: int = 16) -> np.ndarray:
        """``int32`` tables of the chain kernel (``pmc_maf_train_t.tables``):

        ``[16][8]`` ownership of the hidden tiles of the triangular layers: row ``w`` lists the COST RANKS (0 = the
        tile with the longest contraction: ``nT - r`` K tiles when the degree groups fit a tile) wave ``w`` of the
        ``n_waves`` takes, most expensive first, -1 ends the row.  The four SIMDs of a compute unit each run
        ``n_waves / 4`` of the waves (wave ``w`` on SIMD ``w % 4``) and a layer is bound by its busiest f32 matrix pipe,
        so tiles go to the least-loaded SIMD first and to its least-loaded wave (longest processing time first).
        ``[T][D]`` the rank in transform ``t + 1`` of the feature at rank ``r`` of transform ``t`` (last row unused);
        ``[T][D]`` likewise for ``t - 1`` (first row unused)."""
        nT, D, T = self.nT, self.n_dim, self.n_transforms
        own = np.full((16, 8), -1, dtype=np.int64)
        if nT <= 8 * n_waves:
            simd_load = np.zeros(4)
            wave_load = np.zeros(n_waves)
            wave_cnt = np.zeros(n_waves, dtype=np.int64)
            for r in range(nT):
                cost = (nT - r) if self.tri_ok else nT
                live = [sd for sd in range(min(4, n_waves)) if any(wave_cnt[w] < 8 for w in range(sd, n_waves, 4))]
                sd = min(live, key=lambda k: (simd_load[k], k))
                w = min((w for w in range(sd, n_waves, 4) if wave_cnt[w] < 8), key=lambda k: (wave_load[k], k))
                own[w, wave_cnt[w]] = r
                wave_cnt[w] += 1
                wave_load[w] += cost
                simd_load[sd] += cost
        else:
            raise NotImplementedError("more than 8 hidden tiles per wave of the training workgroup")
        f_o_r = [np.argsort(o) for o in self.orders]
        nxt = np.zeros((T, D), dtype=np.int64)
        prv = np.zeros((T, D), dtype=np.int64)
        for t in range(T):
            if t + 1 < T:
                nxt[t] = np.asarray(self.orders[t + 1])[f_o_r[t]]
            if t > 0:
                prv[t] = np.asarray(self.orders[t - 1])[f_o_r[t]]
        return np.concatenate([own.reshape(-1), nxt.reshape(-1), prv.reshape(-1)]).astype(np.int32)

    def train_jobs(self) -> np.ndarray:
        """``int32 [n_jobs][8]``: the weight-gradient tiles of one minibatch, one workgroup of ``maf_dw_kernel`` each.

        ``{kind_a, off_a, kind_b, off_b, g_w, g_b, 0, 0}``: the tile ``dW[out tile][in tile] = sum over the batch's rows
        of delta[out] x activation[in]`` takes its A operand (delta, 16 x rows) from scratch array ``kind_a`` at float
        offset ``off_a`` inside a row set's block and its B operand (activation) likewise; ``g_w`` is the offset of
        the tile's 256 entries in ``gmap`` (or -1: bias only), ``g_b`` the offset of the 16 bias entries of the out
        tile (carried by ONE job per out tile, -1 elsewhere).  Kinds: 0 ``xt_scratch`` (transform inputs), 1
        ``act_scratch`` (h0, h1, h2), 2 ``delta_scratch`` (da0, da1, da2), 3 ``par_scratch`` (output gradients).
        Only tiles with at least one unmasked entry get a job.

        Order: workgroup ``b`` of a launch runs on XCD ``b % 8`` (observed on MI355X, used for speed only), and the eight
        L2s are not shared.  Jobs that read the same operand tiles -- the tiles of one layer of one transform -- are
        therefore placed in the same residue class mod 8 (pieces of a layer's job list, longest first onto the least
        loaded class), so that an operand crosses the fabric into ONE L2 instead of eight; classes are padded to equal
        length with no-op jobs (``kind_a = -1``)."""
        L = self.train_layout()
        _, gm = self.train_index()
        nT, nXT, nOT, Hp, Dp = self.nT, self.nXT, self.nOT, self.Hp, self.Dp
        G = L["gmap_per_transform"]
        ppt = self.par_per_transform()
        jobs, groups, g_start = [], [], 0
        for t in range(self.n_transforms):
            gt = t * G
            layers = (
                # (gmap weights, [out tiles][in tiles], gmap bias, A kind / base, B kind / base)
                ("g0", nT, nXT, "gb0", 2, (t * 3 + 0) * Hp * 16, 0, t * Dp * 16),
                ("g1", nT, nT, "gb1", 2, (t * 3 + 1) * Hp * 16, 1, (t * 3 + 0) * Hp * 16),
                ("g2", nT, nT, "gb2", 2, (t * 3 + 2) * Hp * 16, 1, (t * 3 + 1) * Hp * 16),
                ("g3", nOT, nT, "gb3", 3, t * ppt, 1, (t * 3 + 2) * Hp * 16),
            )
            for gname, n_out_t, n_in_t, bname, ka, base_a, kb, base_b in layers:
                gw0, gb0 = gt + L["offG"][gname], gt + L["offG"][bname]
                n_bias = L["szG"][bname]
                for To in range(n_out_t):
                    bias_live = 16 * To < n_bias and (gm[gb0 + 16 * To: gb0 + min(16 * To + 16, n_bias)] >= 0).any()
                    first = True
                    for Ti in range(n_in_t):
                        off = gw0 + (To * n_in_t + Ti) * 256
                        if not (gm[off: off + 256] >= 0).any():
                            continue
                        jobs.append((ka, base_a + To * 256, kb, base_b + Ti * 256, off,
                                     gb0 + 16 * To if (first and bias_live) else -1, 0, 0))
                        first = False
                    if first and bias_live:                  # an out tile whose weights are all masked still has biases
                        jobs.append((ka, base_a + To * 256, -1, 0, -1, gb0 + 16 * To, 0, 0))
                groups.append(jobs[g_start:])
                g_start = len(jobs)
        # ---- placement by XCD (docstring): pieces of <= piece jobs, longest processing time first
        piece = max(4, -(-len(jobs) // 32))
        pieces = [g[i:i + piece] for g in groups for i in range(0, len(g), piece)]
        buckets = [[] for _ in range(8)]
        for pc in sorted(pieces, key=len, reverse=True):
            min(buckets, key=len).extend(pc)
        buckets.sort(key=len, reverse=True)                  # (block 0, which also adds up the loss, gets a real job)
        depth = len(buckets[0])
        noop = (-1, 0, -1, 0, -1, -1, 0, 0)
        out = [buckets[b][k] if k < len(buckets[b]) else noop for k in range(depth) for b in range(8)]
        while out and out[-1] == noop:
            out.pop()
        return np.asarray(out, dtype=np.int32).reshape(-1, 8)

    def train_index(self):
        """``(packT_idx, gmap)``: gather map for ``packedT`` (like ``pack_index``) and the
        gradient scatter map, both int32, all transforms concatenated.  (Built once per spec: ``train_jobs`` reads it too.)"""
        if getattr(self, "_train_index", None) is None:
            self._train_index = self._build_train_index()
        return self._train_index

    def _build_train_index(self):
        D, H, Hp = self.n_dim, self.hidden, self.Hp
        nT, nXT, nOT = self.nT, self.nXT, self.nOT
        L = self.train_layout()
        pT = np.full(L["pkT_per_transform"] * self.n_transforms, -1, dtype=np.int64)
        gm = np.full(L["gmap_per_transform"] * self.n_transforms, -1, dtype=np.int64)
        lane = np.arange(64)
        li, lk = lane & 15, lane >> 4
        su = self.slot_unit
        for t in range(self.n_transforms):
            base_c = t * self.params_per_transform
            rank = self.orders[t]
            feat_of_rank = np.argsort(rank)
            M0, M1, M2, M3 = self.masks(t)

            def cidx(name, row, col, ncol, mask):
                off, _ = self.offsets[name]
                ok = (row >= 0) & (col >= 0)
                r = np.where(ok, row, 0)
                c = np.where(ok, col, 0)
                ok = ok & mask[r, c]
                return np.where(ok, base_c + off + r * ncol + c, -1)

            def feat(r_in):
                return np.where(r_in < D, feat_of_rank[np.minimum(r_in, D - 1)], -1)

            def orow(o):
                r_out, s = o // self.n_out, o % self.n_out
                return np.where(r_out < D, self.n_out * feat_of_rank[np.minimum(r_out, D - 1)] + s, -1)

            # ---- transposed fragments: A[i = lane&15][k = lane>>4], component c
            f0T = np.full((nXT, nT, 64, 4), -1, dtype=np.int64)
            f1T = np.full((nT, nT, 64, 4), -1, dtype=np.int64)
            f2T = np.full((nT, nT, 64, 4), -1, dtype=np.int64)
            f3T = np.full((nT, nOT, 64, 4), -1, dtype=np.int64)
            # (the inner tile index is broadcast: one numpy call per (tile, component) instead of one per tile pair)
            in_all = su[16 * np.arange(nT)[:, None] + li[None, :]]                 # [Ti][lane]: hidden unit of the in slot
            x_all = feat(16 * np.arange(nXT)[:, None] + li[None, :])               # [Xi][lane]: feature of the in rank
            for To in range(nT):
                for c in range(4):
                    out_u = su[16 * To + 4 * c + lk][None, :]
                    f0T[:, To, :, c] = cidx("W0", np.broadcast_to(out_u, x_all.shape), x_all, D, M0)
                    f1T[:, To, :, c] = cidx("W1", np.broadcast_to(out_u, in_all.shape), in_all, H, M1)
                    f2T[:, To, :, c] = cidx("W2", np.broadcast_to(out_u, in_all.shape), in_all, H, M2)
            for O in range(nOT):
                for c in range(4):
                    crow = orow(16 * O + 4 * c + lk)[None, :]
                    f3T[:, O, :, c] = cidx("W3", np.broadcast_to(crow, in_all.shape), in_all, H, M3)
            # ---- gradient scatter: lane (q = lane>>4, j = lane&15), reg r
            g0 = np.full((nT, nXT, 64, 4), -1, dtype=np.int64)
            g1 = np.full((nT, nT, 64, 4), -1, dtype=np.int64)
            g2 = np.full((nT, nT, 64, 4), -1, dtype=np.int64)
            g3 = np.full((nOT, nT, 64, 4), -1, dtype=np.int64)
            for To in range(nT):
                for r in range(4):
                    out_u = su[16 * To + 4 * lk + r][None, :]
                    g0[To, :, :, r] = cidx("W0", np.broadcast_to(out_u, x_all.shape), x_all, D, M0)
                    g1[To, :, :, r] = cidx("W1", np.broadcast_to(out_u, in_all.shape), in_all, H, M1)
                    g2[To, :, :, r] = cidx("W2", np.broadcast_to(out_u, in_all.shape), in_all, H, M2)
            for O in range(nOT):
                for r in range(4):
                    crow = orow(16 * O + 4 * lk + r)[None, :]
                    g3[O, :, :, r] = cidx("W3", np.broadcast_to(crow, in_all.shape), in_all, H, M3)

            def bidx(name):
                off, _ = self.offsets[name]
                return np.where(su >= 0, base_c + off + np.maximum(su, 0), -1)

            gb3 = np.full(self.Op, -1, dtype=np.int64)
            off3, _ = self.offsets["b3"]
            for r in range(D):
                for j in range(self.n_out):
                    gb3[self.n_out * r + j] = base_c + off3 + self.n_out * feat_of_rank[r] + j
            bT, bG = t * L["pkT_per_transform"], t * L["gmap_per_transform"]
            for name, arr in (("f0T", f0T), ("f1T", f1T), ("f2T", f2T), ("f3T", f3T)):
                a = arr.reshape(-1)
                pT[bT + L["offT"][name]: bT + L["offT"][name] + a.size] = a
            for name, arr in (("g0", g0), ("g1", g1), ("g2", g2), ("g3", g3), ("gb0", bidx("b0")),
                              ("gb1", bidx("b1")), ("gb2", bidx("b2")), ("gb3", gb3)):
                a = arr.reshape(-1)
                gm[bG + L["offG"][name]: bG + L["offG"][name] + a.size] = a
        return pT.astype(np.int32), gm.astype(np.int32)

    # ------------------------------------------------------------ accounting
    def flops_forward_dense(self) -> int:
        """``F_fwd = T*2*(3*D*H + 2*H*H)`` per particle for the affine flows;
        in general the output layer has ``n_out*D`` rows: ``T*2*((1 + n_out)*D*H + 2*H*H)``."""
        D, H = self.n_dim, self.hidden
        return self.n_transforms * 2 * ((1 + self.n_out) * D * H + 2 * H * H)

    def flops_inverse_naive(self) -> int:
        """``F_inv = (D+1) * F_fwd`` per particle."""
        return (self.n_dim + 1) * self.flops_forward_dense()

    def macs_masked(self) -> int:
        """Unmasked multiply-accumulates per particle (the work that is left
        once the masks are exploited)."""
        n = 0
        for t in range(self.n_transforms):
            n += sum(int(M.sum()) for M in self.masks(t))
        return n


SPEC_BY_NAME = {"maf3": 3, "maf6": 6, "maf12": 12}
NSF_BY_NAME = {"nsf3": 3, "nsf6": 6, "nsf12": 12}          # pocomc/flow.py:69-86


def spec_by_name(n_dim: int, name: str) -> "MAFSpec":
    """The six predefined flows of ``pocomc/flow.py:54-86``."""
    if name in SPEC_BY_NAME:
        return MAFSpec(n_dim, SPEC_BY_NAME[name])
    if name in NSF_BY_NAME:
        return MAFSpec(n_dim, NSF_BY_NAME[name], univariate="rqs", bins=8)
    raise KeyError(name)
