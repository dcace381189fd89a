"""How many 64-voxel words stay undecided after the box x block-grid test, by word shape (host analysis of a
4-layer slab's lookup table fetched from the device): y-lines (current), oriented y-lines, 4x4x4 cubes, 2x8x4..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import voxcarve, fixtures_util as fx

G = 1024
cams, masks = fx.golden_cameras(), fx.golden_masks()
H, W = masks[0].shape
eng = voxcarve.CarveEngine(0)
eng.set_grid(G, G, G); eng.set_cameras(cams, H, W); eng.upload_masks(masks)
SH = 2
bs = 1 << SH
gh, gw = (H + bs - 1) // bs, (W + bs - 1) // bs
any_g, all_g = [], []
for m in masks:
    fg = np.zeros((gh * bs, gw * bs), bool); fg[:H, :W] = m > 0
    valid = np.zeros((gh * bs, gw * bs), bool); valid[:H, :W] = True
    blk = fg.reshape(gh, bs, gw, bs)
    vblk = valid.reshape(gh, bs, gw, bs)
    any_g.append(blk.any(axis=(1, 3)))
    all_g.append((blk | ~vblk).all(axis=(1, 3)))
# summed-area tables for O(1) box queries
def sat(a):
    s = np.zeros((a.shape[0] + 1, a.shape[1] + 1), np.int64); s[1:, 1:] = np.cumsum(np.cumsum(a, 0), 1); return s
any_s = [sat(a) for a in any_g]; notall_s = [sat(~a) for a in all_g]
# variant: "all" grid at twice the block size (block (i, j) of it is all-foreground iff its four children are)
def coarsen(a):
    h2, w2 = (a.shape[0] + 1) // 2, (a.shape[1] + 1) // 2
    p_ = np.ones((h2 * 2, w2 * 2), bool); p_[:a.shape[0], :a.shape[1]] = a
    return p_.reshape(h2, 2, w2, 2).all(axis=(1, 3))
notall2_s = [sat(~coarsen(a)) for a in all_g]
COARSE_ALL = len(sys.argv) > 1 and sys.argv[1] == "coarse_all"
def boxsum(s, v0, u0, v1, u1):
    return s[v1 + 1, u1 + 1] - s[v0, u1 + 1] - s[v1 + 1, u0] + s[v0, u0]

for z0 in (400, 512, 640):
    eng.set_slab(z0, z0 + 4); eng.build_lut()
    n = G * G * 4
    lut = np.stack([eng.fetch_lut(c) for c in range(4)])            # [4][n], i = izl*nx*ny + ix*ny + iy
    lut = lut.reshape(4, 4, G, G)                                   # cam, izl, ix, iy
    pv, pu = lut // W, lut % W
    inside = lut >= 0
    def classify(shape):
        dz, dx, dy = shape
        # regroup to [cam, words, 64]
        def grp(a):
            a = a.reshape(4, 4 // dz, dz, G // dx, dx, G // dy, dy).transpose(0, 1, 3, 5, 2, 4, 6)
            return a.reshape(4, -1, dz * dx * dy)
        PV, PU, IN = grp(pv), grp(pu), grp(inside)
        big = 1 << 20
        v0 = np.where(IN, PV, big).min(2); v1 = np.where(IN, PV, -1).max(2)
        u0 = np.where(IN, PU, big).min(2); u1 = np.where(IN, PU, -1).max(2)
        empty = v1 < 0
        allin = IN.all(2)
        bv0, bv1, bu0, bu1 = np.minimum(v0, H - 1) >> SH, np.maximum(v1, 0) >> SH, np.minimum(u0, W - 1) >> SH, np.maximum(u1, 0) >> SH
        dead = np.zeros(PV.shape[1], bool); need = np.zeros(PV.shape[1], bool)
        area = []
        for c in range(4):
            na = boxsum(any_s[c], bv0[c], bu0[c], bv1[c], bu1[c])
            nn = (boxsum(notall2_s[c], bv0[c] >> 1, bu0[c] >> 1, bv1[c] >> 1, bu1[c] >> 1) if COARSE_ALL
                  else boxsum(notall_s[c], bv0[c], bu0[c], bv1[c], bu1[c]))
            rej = empty[c] | (na == 0)
            acc = ~rej & allin[c] & (nn == 0)
            dead |= rej
            need |= ~rej & ~acc
            area.append(((bv1[c] - bv0[c] + 1) * (bu1[c] - bu0[c] + 1))[~empty[c]].mean())
        und = ~dead & need
        return dead.mean(), (~dead & ~need).mean(), und.mean(), np.mean(area)
    for shape in ((1, 1, 64), (1, 4, 16)):
        d, a, u, ar = classify(shape)
        print("z0 %4d shape dz,dx,dy=%s  dead %.4f accepted %.4f undecided %.4f  mean box area %.1f blocks" % (z0, shape, d, a, u, ar), flush=True)
