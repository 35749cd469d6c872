"""Scene construction on our side: the reference's ``Triangle.from_points`` /
``to_points`` (src/tracer.hpp:442-506, generalised cross product src/geometry.hpp:858-906), the prototypes
(``TrianglePrototype``, ``SolidPrototype``; src/tracer.hpp:1363-1439) and ``build_kdtree`` /
``build_composite_scene`` (src/tracer.hpp:1965-2455).

Host-side, runs once per scene.  The plane / edge-normal records are computed in float64 and rounded to fp32
(the reference runs an fp32 LU; values agree to ~1e-7 relative).  The k-d builder is NOT a restatement of the
reference's: nearest-hit results do not depend on the tree, so this one is a plain SAH sweep
over primitive bounding boxes with conservative (bounding-box) overlap -- a primitive may land in a cell it only
touches, which costs a test but never a hit.  The reference's cost constants and limits are kept
(KD_DEFAULT_MAX_DEPTH = 25, split threshold 2; tracer.hpp:41-44).
"""
import numpy as np

f32 = np.float32

KD_DEFAULT_MAX_DEPTH = 25       # tracer.hpp:41 (BATCH_SIZE > 1)
KD_DEFAULT_SPLIT_THRESHOLD = 2  # tracer.hpp:44
TRAVERSAL_COST = 1.0
INTERSECTION_COST = 1.0


def cross(vectors):
    """Generalised cross product of n-1 vectors of dimension n (geometry.hpp:858-882):
    r[i] = f_i * det(M_i), M_i = the vectors' components without coordinate i, f alternating from
    +1 (odd n) / -1 (even n)."""
    vs = np.asarray([list(v) for v in vectors], dtype=np.float64)
    n = vs.shape[1]
    if vs.shape[0] != n - 1:
        raise ValueError("cross product of dimension %d needs %d vectors" % (n, n - 1))
    r = np.zeros(n)
    f = 1.0 if n % 2 else -1.0
    for i in range(n):
        m = np.delete(vs, i, axis=1).T          # rows: remaining coordinates, columns: vectors
        r[i] = f * np.linalg.det(m)
        f = -f
    return r


def from_points_record(points):
    """(p1, face_normal, edge_normals) of the (n-1)-simplex with the given n vertices (tracer.hpp:442-462)."""
    pts = np.asarray([list(p) for p in points], dtype=np.float64)
    n = pts.shape[1]
    if pts.shape[0] != n:
        raise ValueError("a simplex in %d dimensions has %d vertices" % (n, n))
    p1 = pts[0]
    vsides = pts[1:] - p1
    N = cross(vsides)
    square = float(N @ N)
    if square == 0.0:
        raise ValueError("the points are not linearly independent")
    edges = []
    for i in range(n - 1):
        vs = vsides.copy()
        vs[i] = N
        edges.append(cross(vs) / square)
    return p1.astype(f32), N.astype(f32), np.asarray(edges, f32)


def to_points_array(p1, face_normal, edge_normals):
    """Triangle.to_points as the reference computes it (tracer.hpp:490-506): p1 + cross(edge normals with the
    i-th replaced by the face normal).  NOTE: because of the alternating sign in the generalised cross product
    this returns the true vertices only in odd dimensions; in even dimensions the reference (and therefore this
    function) returns p1 - (p_k - p1).  The reference's own test covers n = 5 only.  Use vertices_of() for the
    geometry."""
    p1 = np.asarray(p1, np.float64)
    fn = np.asarray(face_normal, np.float64)
    en = np.asarray(edge_normals, np.float64)
    n = len(p1)
    pts = [p1]
    for i in range(n - 1):
        vs = en.copy()
        vs[i] = fn
        pts.append(cross(vs) + p1)
    return np.asarray(pts, f32)


def vertices_of(p1, face_normal, edge_normals):
    """The true vertices of a simplex record, by solving  E_j . s_k = -delta_jk, N . s_k = 0  for the edge
    vectors s_k = p_k - p1 (the barycentric reading of tracer.hpp:426-431).  Unlike to_points_array this is
    correct in every dimension."""
    p1 = np.asarray(p1, np.float64)
    n = len(p1)
    m = np.vstack([np.asarray(edge_normals, np.float64).reshape(n - 1, n), np.asarray(face_normal, np.float64)])
    rhs = np.vstack([-np.eye(n - 1), np.zeros((1, n - 1))])
    sk = np.linalg.solve(m, rhs)          # columns are the edge vectors
    return np.vstack([p1, p1 + sk.T]).astype(f32)


def solid_bounds(type_cube, position, orientation):
    """World-space bounding box of a Solid: x = orientation * (u + position), u in [-1,1]^n (cube) or
    |u| <= 1 (sphere) -- solid::intersects (tracer.hpp:257-260) read backwards."""
    o = np.asarray(orientation, np.float64)
    c = o @ np.asarray(position, np.float64)
    ext = np.abs(o).sum(axis=1) if type_cube else np.sqrt((o * o).sum(axis=1))
    return (c - ext).astype(f32), (c + ext).astype(f32)


class _Item(object):
    __slots__ = ("prim", "lo", "hi")

    def __init__(self, prim, lo, hi):
        self.prim = prim
        self.lo = np.asarray(lo, np.float64)
        self.hi = np.asarray(hi, np.float64)


def group_batches(tri_items, batch_size, make_batch):
    """Pack triangles into batches of `batch_size` spatial neighbours; leftovers stay unbatched.  (The reference
    sorts by centre along the widest axis and packs greedily by a distance metric, tracer.hpp:2395-2427; here the
    centres are split recursively at the median of their widest axis until groups of `batch_size` remain, which
    keeps each batch's bounding box small in every axis.)"""
    if len(tri_items) < batch_size:
        return [], list(tri_items)
    centres = np.asarray([(it.lo + it.hi) * 0.5 for it in tri_items])
    groups = []

    def split(idx):
        if len(idx) <= batch_size:
            groups.append(idx)
            return
        c = centres[idx]
        axis = int(np.argmax(c.max(axis=0) - c.min(axis=0)))
        order = idx[np.argsort(c[:, axis], kind="stable")]
        # cut at a multiple of batch_size so that at most one group is incomplete
        half = (len(order) // 2 + batch_size - 1) // batch_size * batch_size
        if half >= len(order):
            half = len(order) - batch_size if len(order) > batch_size else len(order) // 2
        split(order[:half])
        split(order[half:])

    split(np.arange(len(tri_items)))
    batches, loose = [], []
    for grp in groups:
        if len(grp) == batch_size:
            members = [tri_items[i] for i in grp]
            lo = np.min([g.lo for g in members], axis=0)
            hi = np.max([g.hi for g in members], axis=0)
            batches.append(_Item(make_batch([g.prim for g in members]), lo, hi))
        else:
            loose.extend(tri_items[i] for i in grp)
    return batches, loose


def build_tree(items, make_leaf, make_branch, max_depth=KD_DEFAULT_MAX_DEPTH, split_threshold=KD_DEFAULT_SPLIT_THRESHOLD):
    """SAH k-d tree over item bounding boxes.  Returns (lo, hi, root)."""
    if not items:
        raise ValueError("cannot build a k-d tree from zero primitives")
    los = np.asarray([it.lo for it in items])
    his = np.asarray([it.hi for it in items])
    lo = los.min(axis=0)
    hi = his.max(axis=0)
    n = len(lo)

    def area(l, h):
        # surface measure of an n-box up to a constant: sum over axes of the product of the other extents
        e = np.maximum(h - l, 0.0)
        if n == 1:
            return 1.0
        tot = 0.0
        for a in range(n):
            tot += float(np.prod(np.delete(e, a)))
        return tot

    def rec(idx, l, h, depth):
        if len(idx) <= split_threshold or depth >= max_depth:
            return make_leaf([items[i].prim for i in idx])
        base = area(l, h)
        best = None
        leaf_cost = INTERSECTION_COST * len(idx)
        if base > 0.0:
            for axis in np.argsort(-(h - l))[:min(n, 3)]:
                axis = int(axis)
                if h[axis] - l[axis] <= 0.0:
                    continue
                s = los[idx, axis]
                e = his[idx, axis]
                cands = np.unique(np.concatenate([s, e]))
                cands = cands[(cands > l[axis]) & (cands < h[axis])]
                if len(cands) == 0:
                    continue
                if len(cands) > 256:
                    cands = cands[np.linspace(0, len(cands) - 1, 256).astype(int)]
                flat = ((s[None, :] == cands[:, None]) & (e[None, :] == cands[:, None])).sum(axis=1)
                nl = (s[None, :] < cands[:, None]).sum(axis=1) + flat
                nr = (e[None, :] > cands[:, None]).sum(axis=1) + flat
                # area of a box is linear in its extent along `axis`: A0 + A1*x
                ext = np.maximum(h - l, 0.0)
                others = np.delete(ext, axis)
                a0 = float(np.prod(others))
                a1 = sum(float(np.prod(np.delete(others, k))) for k in range(len(others))) if len(others) > 1 else 1.0
                cost = TRAVERSAL_COST + INTERSECTION_COST * ((a0 + a1 * (cands - l[axis])) * nl + (a0 + a1 * (h[axis] - cands)) * nr) / base
                cost = np.where((nl == 0) | (nr == 0), cost * 0.8, cost)      # cutting off empty space is worth more
                k = int(np.argmin(cost))
                if best is None or cost[k] < best[0]:
                    best = (float(cost[k]), axis, float(cands[k]), int(nl[k]), int(nr[k]))
        if best is None or best[0] >= leaf_cost or (best[3] == len(idx) and best[4] == len(idx)):
            return make_leaf([items[i].prim for i in idx])
        _, axis, c, _, _ = best
        c32 = float(f32(c))
        s = los[idx, axis]
        e = his[idx, axis]
        # conservative membership with the fp32 split the kernels will see; flat primitives in the plane go both ways
        left = idx[(s < c32) | ((s <= c32) & (e <= c32) & (s == e))]
        right = idx[(e > c32) | ((s >= c32) & (e >= c32) & (s == e))]
        both_flat = idx[(s == c32) & (e == c32)]
        left = np.union1d(left, both_flat)
        right = np.union1d(right, both_flat)
        if len(left) == len(idx) and len(right) == len(idx):
            return make_leaf([items[i].prim for i in idx])
        hl = h.copy()
        hl[axis] = c32
        lr = l.copy()
        lr[axis] = c32
        ln = rec(left, l, hl, depth + 1) if len(left) else None
        rn = rec(right, lr, h, depth + 1) if len(right) else None
        if ln is None and rn is None:
            return make_leaf([items[i].prim for i in idx])
        return make_branch(axis, c32, ln, rn)

    root = rec(np.arange(len(items)), lo.copy(), hi.copy(), 0)
    return lo.astype(f32), hi.astype(f32), root
