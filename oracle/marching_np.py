"""Marching cubes over a dense ON/OFF volume: case table + numpy restatement.  TEST INFRASTRUCTURE (see oracle/__init__.py).

Reference consumer: voxel_reconstruction.py:127-163 ``plot_marching_cubes`` -> ``skimage.measure.marching_cubes(
voxels_status, 0)`` on the boolean volume ``np.reshape(statuses, (width, height*2, depth))`` (assignment.py:143-146; the
call site sits inside a string literal, the function itself is live).  scikit-image is not importable here and the
reference holds no mesh output, so this is PARITY UNPINNED against skimage's Lewiner variant: what is restated is the
classic Lorensen-Cline algorithm -- one case per 8 corner bits, vertices on the cube edges that join an ON and an OFF
corner, placed at ``off + level * (on - off)`` (the reference passes level 0: on the OFF corner; 0.5: the midpoint).

The 256-case table is GENERATED here rather than copied from anywhere: for every face of the cube the boundary segments
follow from its four corner bits alone (an ambiguous face -- two ON corners on a diagonal -- always separates the ON
corners), so the cube on the other side of a face draws the same segments and the surface is closed by construction;
the segments of the six faces chain into closed loops, each loop is fan-triangulated and oriented from ON to OFF.
``scripts/gen_mc_table.py`` writes the same table as csrc/mc_table.h for the HIP kernels; tests compare the two, check the
table's invariants (watertight, consistently oriented, Euler characteristic, enclosed volume) and the device mesh against
``extract`` below, index for index.

Conventions shared with the device: corner k of a cell has offsets (a, b, c) = (k >> 2 & 1, k >> 1 & 1, k & 1) along the
volume's axes 0, 1, 2 (C order: axis 2 fastest); edge e = axis * 4 + m is the cube edge along ``axis`` whose other two
offsets are the bits of m (high bit = the lower-numbered axis).  Vertices are numbered per 64 consecutive volume elements
(a "word"): within a word first the crossings along axis 0 in element order, then axis 1, then axis 2; an edge belongs to
its lower corner.  Faces come cell by cell in linear order, triangles of a cell in table order.
"""
import numpy as np


def corner_offsets(k):
    return (k >> 2) & 1, (k >> 1) & 1, k & 1


def edge_corners(e):
    """(lower corner, upper corner) of cube edge e = axis * 4 + m."""
    axis, m = divmod(e, 4)
    others = [ax for ax in range(3) if ax != axis]
    off = [0, 0, 0]
    off[others[0]] = (m >> 1) & 1
    off[others[1]] = m & 1
    lo = (off[0] << 2) | (off[1] << 1) | off[2]
    off[axis] = 1
    hi = (off[0] << 2) | (off[1] << 1) | off[2]
    return lo, hi


def _edge_between(k0, k1):
    for e in range(12):
        if set(edge_corners(e)) == {k0, k1}:
            return e
    raise ValueError((k0, k1))


# the six faces as cyclic corner quadruples
def _faces():
    out = []
    for axis in range(3):
        others = [ax for ax in range(3) if ax != axis]
        for side in (0, 1):
            quad = []
            for (p, q) in ((0, 0), (0, 1), (1, 1), (1, 0)):
                off = [0, 0, 0]
                off[axis] = side
                off[others[0]] = p
                off[others[1]] = q
                quad.append((off[0] << 2) | (off[1] << 1) | off[2])
            out.append(quad)
    return out


def build_tables():
    """-> (ntri uint8 [256], tri uint8 [256, 15]) : edges of the triangles of each case, 255-padded."""
    faces = _faces()
    ntri = np.zeros(256, np.uint8)
    tri = np.full((256, 15), 255, np.uint8)
    corner_pos = np.array([corner_offsets(k) for k in range(8)], dtype=np.float64)
    for case in range(256):
        on = [(case >> k) & 1 for k in range(8)]
        segs = []                                  # undirected segments between cube edges
        for quad in faces:
            b = [on[k] for k in quad]
            edges = [_edge_between(quad[i], quad[(i + 1) % 4]) for i in range(4)]     # edge i joins corner i and i+1
            crossed = [i for i in range(4) if b[i] != b[(i + 1) % 4]]
            if len(crossed) == 2:
                segs.append((edges[crossed[0]], edges[crossed[1]]))
            elif len(crossed) == 4:
                # two ON corners on a diagonal: each ON corner is cut off on its own (fixed rule, the same from both sides)
                for i in range(4):
                    if b[i]:
                        segs.append((edges[(i - 1) % 4], edges[i]))
        # chain the segments into closed loops
        loops = []
        left = list(segs)
        while left:
            a, bnd = left.pop()
            loop = [a, bnd]
            while loop[-1] != loop[0]:
                for i, (p, q) in enumerate(left):
                    if p == loop[-1] or q == loop[-1]:
                        loop.append(q if p == loop[-1] else p)
                        left.pop(i)
                        break
                else:
                    raise AssertionError("open loop in case %d" % case)
            loops.append(loop[:-1])
        tris = []
        for loop in loops:
            mid = np.array([(corner_pos[edge_corners(e)[0]] + corner_pos[edge_corners(e)[1]]) / 2 for e in loop])
            nrm = np.zeros(3)
            for i in range(len(loop)):
                nrm += np.cross(mid[i], mid[(i + 1) % len(loop)])             # Newell
            d = np.zeros(3)
            for e in loop:
                k0, k1 = edge_corners(e)
                d += (corner_pos[k1] - corner_pos[k0]) * (1 if on[k0] else -1)   # from the ON end to the OFF end
            assert abs(nrm @ d) > 1e-9, case
            if nrm @ d < 0:
                loop = loop[::-1]
            for i in range(1, len(loop) - 1):
                tris.append((loop[0], loop[i], loop[i + 1]))
        assert len(tris) <= 5, (case, len(tris))
        ntri[case] = len(tris)
        for t, (e0, e1, e2) in enumerate(tris):
            tri[case, 3 * t:3 * t + 3] = (e0, e1, e2)
    return ntri, tri


NTRI, TRI = build_tables()


def extract(volume, level=0.0):
    """(verts float32 [V, 3] in index coordinates (axis 0, 1, 2), faces uint32 [F, 3]) of a boolean volume."""
    vol = np.ascontiguousarray(volume).astype(bool)
    d0, d1, d2 = vol.shape
    flat = vol.reshape(-1)
    n = flat.size
    idx = np.arange(n, dtype=np.int64)
    a, b, c = idx // (d1 * d2), (idx // d2) % d1, idx % d2
    strides = (d1 * d2, d2, 1)
    limits = (a < d0 - 1, b < d1 - 1, c < d2 - 1)
    # crossed edges, owned by their lower corner
    keys, pos = [], []
    for axis in range(3):
        ok = limits[axis]
        lo = idx[ok]
        cr = lo[flat[lo] != flat[lo + strides[axis]]]
        keys.append(np.stack([cr // 64, np.full(cr.size, axis), cr % 64, cr], axis=1))
    keys = np.concatenate(keys) if keys else np.zeros((0, 4), np.int64)
    order = np.lexsort((keys[:, 2], keys[:, 1], keys[:, 0]))
    keys = keys[order]
    vid = {}
    verts = np.zeros((keys.shape[0], 3), np.float32)
    for i, (_, axis, _, lo) in enumerate(keys):
        vid[(int(lo), int(axis))] = i
        p = np.array([lo // (d1 * d2), (lo // d2) % d1, lo % d2], dtype=np.float64)
        on_lo = bool(flat[lo])
        t = (1.0 - level) if on_lo else level                  # distance from the lower corner: off + level * (on - off)
        p[axis] += t
        verts[i] = p
    faces = []
    cells = idx[limits[0] & limits[1] & limits[2]]
    case = np.zeros(cells.size, np.int64)
    for k in range(8):
        oa, ob, oc = corner_offsets(k)
        case |= flat[cells + oa * strides[0] + ob * strides[1] + oc].astype(np.int64) << k
    for cell, cs in zip(cells[(case != 0) & (case != 255)], case[(case != 0) & (case != 255)]):
        for t in range(NTRI[cs]):
            f = []
            for e in TRI[cs, 3 * t:3 * t + 3]:
                k0, _ = edge_corners(int(e))
                oa, ob, oc = corner_offsets(k0)
                f.append(vid[(int(cell + oa * strides[0] + ob * strides[1] + oc), int(e) // 4)])
            faces.append(f)
    return verts, np.array(faces, dtype=np.uint32).reshape(-1, 3)


def mesh_invariants(verts, faces):
    """(every edge shared by exactly two triangles with opposite directions, Euler characteristic V - E + F, signed volume)."""
    f = faces.astype(np.int64)
    he = np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]])
    und = np.sort(he, axis=1)
    _, inv, cnt = np.unique(und, axis=0, return_inverse=True, return_counts=True)
    closed = bool(np.all(cnt == 2))
    # opposite directions: each directed half-edge's reverse must exist exactly once
    key = he[:, 0] * (f.max() + 1 if f.size else 1) + he[:, 1]
    rkey = he[:, 1] * (f.max() + 1 if f.size else 1) + he[:, 0]
    oriented = bool(np.array_equal(np.sort(key), np.sort(rkey)) and np.unique(key).size == key.size)
    used = np.unique(f).size
    chi = used - cnt.size + f.shape[0]
    v = verts.astype(np.float64)
    vol = float(np.einsum("ij,ij->i", v[f[:, 0]], np.cross(v[f[:, 1]], v[f[:, 2]])).sum() / 6.0)
    return closed, oriented, int(chi), vol
