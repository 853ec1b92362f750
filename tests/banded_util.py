"""Test helper: turn a dense (P, q, A, l, u) of the oracle into the banded/ELL arrays of pqp_banded_qp.hpp under a
variable permutation, run the host emulation of the core, and map the solution back."""
import ctypes as C

import numpy as np

import emu_util as E

RMAX, CMAX = 4, 6


def interleave_tension2(n):
    """reference order [x(n) y(n) theta(n) k(n-1)] -> position in the point-interleaved order"""
    pos = np.zeros(4 * n - 1, dtype=np.int64)
    for i in range(n):
        pos[i] = 4 * i; pos[n + i] = 4 * i + 1; pos[2 * n + i] = 4 * i + 2
        if i < n - 1:
            pos[3 * n + i] = 4 * i + 3
    return pos


def interleave3(n):
    """[a(n) b(n) c(n)] -> (a_i, b_i, c_i) per point"""
    pos = np.zeros(3 * n, dtype=np.int64)
    for i in range(n):
        pos[i] = 3 * i; pos[n + i] = 3 * i + 1; pos[2 * n + i] = 3 * i + 2
    return pos


def to_banded(P, q, A, lo, up, pos):
    nv, nc = P.shape[0], A.shape[0]
    Pp = np.zeros_like(P); Pp[np.ix_(pos, pos)] = P
    Ap = np.zeros_like(A); Ap[:, pos] = A
    qp = np.zeros(nv); qp[pos] = q
    pbw = 0
    ii, jj = np.nonzero(Pp)
    if len(ii):
        pbw = int(np.max(np.abs(ii - jj)))
    bw = pbw
    acol = -np.ones((nc, RMAX), dtype=np.int32); aval = np.zeros((nc, RMAX))
    trow = -np.ones((nv, CMAX), dtype=np.int32); tslot = np.zeros((nv, CMAX), dtype=np.int32)
    fill = np.zeros(nv, dtype=np.int64)
    for r in range(nc):
        cs = np.nonzero(Ap[r])[0]
        assert len(cs) <= RMAX
        if len(cs):
            bw = max(bw, int(cs.max() - cs.min()))
        for s, c in enumerate(cs):
            acol[r, s] = c; aval[r, s] = Ap[r, c]
            trow[c, fill[c]] = r; tslot[c, fill[c]] = s; fill[c] += 1
    assert fill.max() <= CMAX
    pband = np.zeros((pbw + 1, nv))
    for d in range(pbw + 1):
        for j in range(nv - d):
            pband[d, j] = Pp[j + d, j]
    return dict(nv=nv, nc=nc, bw=bw, pbw=pbw, pband=pband, q=qp, acol=acol, aval=aval, trow=trow, tslot=tslot, lo=lo.copy(), up=up.copy())


def emu_solve(prm, b):
    lib = E.load()
    vp = lambda a: np.ascontiguousarray(a).ctypes.data_as(C.c_void_p)
    x = np.zeros(b["nv"]); y = np.zeros(b["nc"]); st = np.zeros(1, dtype=np.int32); it = np.zeros(1, dtype=np.int32); info = np.zeros(8)
    arrs = {k: np.ascontiguousarray(b[k]) for k in ("pband", "q", "acol", "aval", "trow", "tslot", "lo", "up")}
    lib.pqp_emu_banded_solve(C.byref(prm), 1, b["nv"], b["nc"], b["bw"], b["pbw"], vp(arrs["pband"]), vp(arrs["q"]), vp(arrs["acol"]),
                             vp(arrs["aval"]), vp(arrs["trow"]), vp(arrs["tslot"]), vp(arrs["lo"]), vp(arrs["up"]), vp(x), vp(y), vp(st), vp(it), vp(info))
    return dict(x=x, y=y, status=int(st[0]), iters=int(it[0]), info=info)
