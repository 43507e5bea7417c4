"""Block decomposition of the global domain, as CICE defines it.

Mirrors the *semantics* of create_blocks
(cicecore/cicedyn/infrastructure/ice_blocks.F90:119-333): blocks of
block_size_x x block_size_y interior cells with one ghost ring, numbered west to
east then south to north; the last block in a direction is padded (its ihi/jhi
are smaller than nx_block-1/ny_block-1) when the size does not divide evenly.
Block -> rank assignment is a cartesian split (distribution_type='cartesian',
shared/ice_distribution.F90:91-119): the grid of blocks is cut into
px x py rectangles of whole blocks.

Also provides scatter/gather between global [ny][nx] arrays and block arrays
[nblocks][ny_block][nx_block] with ghost cells filled by the meaning of a ghost
cell (same global cell; cyclic wrap; closed: left at `fill`).
"""
from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

NGHOST = 1


@dataclass
class Block:
    gid: int          # 0-based global block id (reference: block_id - 1)
    iblock: int       # 1-based position in the block grid
    jblock: int
    gi0: int          # global index (1-based) of first interior cell
    gj0: int
    gnx: int          # interior extent
    gny: int
    owner: int = 0
    local: int = 0

    # local 1-based index ranges like type(block)
    @property
    def ilo(self): return NGHOST + 1
    @property
    def jlo(self): return NGHOST + 1
    @property
    def ihi(self): return NGHOST + self.gnx
    @property
    def jhi(self): return NGHOST + self.gny


@dataclass
class Decomp:
    nx_global: int
    ny_global: int
    block_size_x: int
    block_size_y: int
    ew: str = "cyclic"
    ns: str = "closed"
    nranks: int = 1
    proc_shape: tuple | None = None      # (px, py) ranks in x and y
    blocks: list = field(default_factory=list)
    nblocks_x: int = 0
    nblocks_y: int = 0

    def __post_init__(self):
        bx, by = self.block_size_x, self.block_size_y
        self.nx_block = bx + 2 * NGHOST
        self.ny_block = by + 2 * NGHOST
        self.nblocks_x = (self.nx_global - 1) // bx + 1
        self.nblocks_y = (self.ny_global - 1) // by + 1
        if self.proc_shape is None:
            self.proc_shape = _default_proc_shape(self.nranks, self.nblocks_x, self.nblocks_y)
        px, py = self.proc_shape
        assert px * py == self.nranks, "proc_shape must multiply to nranks"
        counts = [0] * self.nranks
        gid = 0
        for jb in range(1, self.nblocks_y + 1):
            js = (jb - 1) * by + 1
            je = min(js + by - 1, self.ny_global)
            for ib in range(1, self.nblocks_x + 1):
                is_ = (ib - 1) * bx + 1
                ie = min(is_ + bx - 1, self.nx_global)
                rx = (ib - 1) * px // self.nblocks_x
                ry = (jb - 1) * py // self.nblocks_y
                owner = ry * px + rx
                self.blocks.append(Block(gid, ib, jb, is_, js, ie - is_ + 1, je - js + 1, owner, counts[owner]))
                counts[owner] += 1
                gid += 1

    # ---- per-rank views ---------------------------------------------------
    def local_blocks(self, rank: int = 0):
        return sorted([b for b in self.blocks if b.owner == rank], key=lambda b: b.local)

    def shape(self, rank: int = 0):
        return (len(self.local_blocks(rank)), self.ny_block, self.nx_block)

    # ---- global <-> block arrays -------------------------------------------
    FOLD_OFFSETS = {"center": (0, 0), "NEcorner": (1, 1), "Eface": (1, 0), "Nface": (0, 1)}

    def scatter(self, g: np.ndarray, rank: int = 0, fill=0.0, fold=None) -> np.ndarray:
        """Global [ny][nx] -> block array of `rank`, ghost cells included.  fold = (location, sign) on a tripole grid:
        the ghost row beyond the fold takes sign * the mirrored row (u-fold, offsets per location as
        ice_boundary.F90:1626-1683: ghost(i, NY+1) <- g(NX - i + 1 - ioff, NY - joff)); without it that row is `fill`."""
        out = self._scatter_plain(g, rank, fill)
        if fold is None or self.ns != "tripole":
            return out
        loc, sign = fold
        ioff, joff = self.FOLD_OFFSETS[loc]
        NX, NY = self.nx_global, self.ny_global
        for b in self.local_blocks(rank):
            if b.gj0 + b.gny - 1 != NY:
                continue
            ii = (np.arange(b.gi0 - NGHOST, b.gi0 + b.gnx + NGHOST) - 1) % NX + 1      # global i of the local columns
            src = (NX - ii + 1 - ioff - 1) % NX                                         # 0-based mirrored column
            out[b.local][NGHOST + b.gny, :len(ii)] = sign * g[NY - joff - 1, src]
        return out

    def _scatter_plain(self, g: np.ndarray, rank: int = 0, fill=0.0) -> np.ndarray:
        blks = self.local_blocks(rank)
        out = np.full((len(blks), self.ny_block, self.nx_block), fill, dtype=g.dtype)
        NX, NY = self.nx_global, self.ny_global
        for b in blks:
            jj = np.arange(b.gj0 - NGHOST, b.gj0 + b.gny + NGHOST)    # global j of local rows 0..gny+1
            ii = np.arange(b.gi0 - NGHOST, b.gi0 + b.gnx + NGHOST)
            jv = (jj >= 1) & (jj <= NY)
            iv = (ii >= 1) & (ii <= NX)
            if self.ew == "cyclic":
                ii = (ii - 1) % NX + 1
                iv[:] = True
            if self.ns == "cyclic":
                jj = (jj - 1) % NY + 1
                jv[:] = True
            sub = g[np.ix_(jj[jv] - 1, ii[iv] - 1)]
            rows = np.nonzero(jv)[0]
            cols = np.nonzero(iv)[0]
            out[b.local][np.ix_(rows, cols)] = sub
        return out

    def gather(self, parts: dict, dtype=np.float64) -> np.ndarray:
        """{rank: block array} -> global [ny][nx] from interior cells."""
        g = np.zeros((self.ny_global, self.nx_global), dtype=dtype)
        for b in self.blocks:
            a = parts[b.owner][b.local]
            g[b.gj0 - 1:b.gj0 - 1 + b.gny, b.gi0 - 1:b.gi0 - 1 + b.gnx] = \
                a[NGHOST:NGHOST + b.gny, NGHOST:NGHOST + b.gnx]
        return g


def _default_proc_shape(nranks, nbx, nby):
    """Most-square px x py with px | nranks, preferring cuts along the longer block axis."""
    best = (nranks, 1)
    score = None
    for px in range(1, nranks + 1):
        if nranks % px:
            continue
        py = nranks // px
        if px > nbx or py > nby:
            continue
        s = abs(nbx / px - nby / py)
        if score is None or s < score:
            best, score = (px, py), s
    return best


def single_block(nx_global, ny_global, ew="cyclic", ns="closed") -> Decomp:
    return Decomp(nx_global, ny_global, nx_global, ny_global, ew, ns, 1)


def per_rank_blocks(nx_global, ny_global, nranks, ew="cyclic", ns="closed", proc_shape=None) -> Decomp:
    """One block per rank (one contiguous sub-domain per GPU): the layout the
    multi-GPU benchmark uses.  Block size = ceil(n / p) in each direction."""
    if proc_shape is None:
        proc_shape = _default_proc_shape(nranks, nx_global, ny_global) if nranks > 1 else (1, 1)
        # prefer splitting the longer global axis more often
        cands = [(px, nranks // px) for px in range(1, nranks + 1) if nranks % px == 0]
        proc_shape = min(cands, key=lambda s: abs(nx_global / s[0] - ny_global / s[1]))
    px, py = proc_shape
    bx = -(-nx_global // px)
    by = -(-ny_global // py)
    return Decomp(nx_global, ny_global, bx, by, ew, ns, nranks, proc_shape)
