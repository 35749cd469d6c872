"""Scene construction on our side: the reference's ``Triangle.from_points`` /
``to_points`` (src/tracer.hpp:442-506, generalised cross product src/geometry.hpp:858-906), the prototypes
(``TrianglePrototype``, ``SolidPrototype``; src/tracer.hpp:1363-1439) and ``build_kdtree`` /
``build_composite_scene`` (src/tracer.hpp:1965-2455).

Host-side, runs once per scene.  The plane / edge-normal records are computed in float64 and rounded to fp32
(the reference runs an fp32 LU; values agree to ~1e-7 relative).  The k-d builder is NOT a restatement of the
reference's: nearest-hit results do not depend on the tree.  It is a SAH sweep whose split
candidates come from each simplex's EDGES clipped to the cell (so a long thin simplex crossing a cell diagonally
counts where it actually is, not over its whole bounding box), and whose membership test is a set of separating
axes -- bounding box, the simplex's hyperplane against the cell, and the simplex's shadow against the cell's in
every coordinate plane.  Every axis test is a proof of disjointness, so a primitive may land in a cell it only
nearly touches (costs a test) but is never left out of one it enters (would cost a hit; the kernels' early exit
and pruning rely on this, like the reference's own walk).  The reference's cost constants and limits are kept
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


def from_points_records(simplices):
    """from_points_record for many simplices at once: (count, n, n) points -> (p1 (count,n), face normals (count,n),
    edge normals (count,n-1,n)), all float32.  Same formulas, batched determinants."""
    pts = np.asarray(simplices, np.float64)
    cnt, n, n2 = pts.shape
    if n != n2:
        raise ValueError("a simplex in %d dimensions has %d vertices" % (n2, n2))

    def cross_many(vs):                         # vs: (count, n-1, n)
        r = np.empty((cnt, n))
        f = 1.0 if n % 2 else -1.0
        for i in range(n):
            m = np.delete(vs, i, axis=2)        # (count, n-1 vectors, n-1 coordinates)
            r[:, i] = f * np.linalg.det(np.swapaxes(m, 1, 2))
            f = -f
        return r

    p1 = pts[:, 0, :]
    vsides = pts[:, 1:, :] - p1[:, None, :]
    N = cross_many(vsides)
    square = (N * N).sum(axis=1)
    if (square == 0.0).any():
        raise ValueError("the points are not linearly independent")
    edges = np.empty((cnt, n - 1, n))
    for i in range(n - 1):
        vs = vsides.copy()
        vs[:, i, :] = N
        edges[:, i, :] = cross_many(vs) / square[:, None]
    return p1.astype(f32), N.astype(f32), edges.astype(f32)


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


def vertices_of_many(recs, n):
    """vertices_of for an array of records (count, >= 1+n+n+(n-1)*n) in the device layout
    [d, face_normal[n], p1[n], edge_normals[n-1][n]] -> (count, n, n) float64."""
    r = np.asarray(recs, np.float64)
    fn = r[:, 1:1 + n]
    p1 = r[:, 1 + n:1 + 2 * n]
    en = r[:, 1 + 2 * n:1 + 2 * n + (n - 1) * n].reshape(-1, n - 1, n)
    m = np.concatenate([en, fn[:, None, :]], axis=1)                     # rows: E_0..E_{n-2}, N
    rhs = np.zeros((len(r), n, n - 1))
    rhs[:, :n - 1, :] = -np.eye(n - 1)
    sk = np.linalg.solve(m, rhs)                                         # columns: edge vectors
    out = np.empty((len(r), n, n))
    out[:, 0, :] = p1
    out[:, 1:, :] = p1[:, None, :] + np.swapaxes(sk, 1, 2)
    return out


def solid_bounds(type_cube, position, orientation):
    """World-space bounding box of a Solid: x = orientation * (u + position), u in [-1,1]^n (cube) or
    |u| <= 1 (sphere) -- solid::intersects (tracer.hpp:257-260) read backwards."""
    o = np.asarray(orientation, np.float64)
    c = o @ np.asarray(position, np.float64)
    ext = np.abs(o).sum(axis=1) if type_cube else np.sqrt((o * o).sum(axis=1))
    return (c - ext).astype(f32), (c + ext).astype(f32)


class _Item(object):
    """A primitive (or batch) for the builder: its bounding box and, for simplices, the vertex arrays
    (count, n, n); None for solids, which are placed by bounding box alone."""
    __slots__ = ("prim", "lo", "hi", "simplices")

    def __init__(self, prim, lo, hi, simplices=None):
        self.prim = prim
        self.lo = np.asarray(lo, np.float64)
        self.hi = np.asarray(hi, np.float64)
        self.simplices = None if simplices is None else np.asarray(simplices, np.float64).reshape(-1, len(self.lo), len(self.lo))


def group_batches(tri_items, batch_size, make_batch):
    """Pack triangles into batches of `batch_size` spatial neighbours; leftovers form a padded batch.  (The reference
    sorts by centre along the widest axis and packs greedily by a distance metric, tracer.hpp:2395-2427; here the
    centres are split recursively at the median of their widest axis until groups of `batch_size` remain, which
    keeps each batch's bounding box small in every axis.)"""
    if not tri_items:
        return [], []
    if len(tri_items) < batch_size:
        return [_pad_batch(list(tri_items), batch_size, make_batch)], []
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
            simp = None
            if all(g.simplices is not None for g in members):
                simp = np.concatenate([g.simplices for g in members])
            batches.append(_Item(make_batch([g.prim for g in members]), lo, hi, simp))
        else:
            loose.extend(tri_items[i] for i in grp)
    if loose:
        # the leftovers (fewer than a batch) become one more batch, its last triangle repeated: a duplicate lane can
        # never win (strict <), and a scene made of batches only is eligible for the packet kernel.  (The reference
        # leaves them as single triangles, tracer.hpp:2395-2427.)
        batches.append(_pad_batch(loose, batch_size, make_batch))
        loose = []
    return batches, loose


def _pad_batch(members, batch_size, make_batch):
    lo = np.min([g.lo for g in members], axis=0)
    hi = np.max([g.hi for g in members], axis=0)
    simp = np.concatenate([g.simplices for g in members]) if all(g.simplices is not None for g in members) else None
    prims = [g.prim for g in members]
    prims += [prims[-1]] * (batch_size - len(prims))
    return _Item(make_batch(prims), lo, hi, simp)


def build_tree(items, make_leaf, make_branch, max_depth=KD_DEFAULT_MAX_DEPTH, split_threshold=KD_DEFAULT_SPLIT_THRESHOLD,
               traversal_cost=0.0, intersection_cost=0.0):
    """SAH k-d tree over the items, built by the native builder (csrc/nt_builder.cpp, nt_kdtree_build): exact
    clipping of every simplex to every cell it is tested against.  Returns (lo, hi, root) with the nodes made
    through make_leaf(list of prims) / make_branch(axis, split, left, right)."""
    lo, hi, axis, split, left, right, leaf_items, root = build_tree_arrays(items, max_depth, split_threshold, traversal_cost,
                                                                           intersection_cost)
    made = {}
    # children before parents, without recursion (trees are up to max_depth deep but can be wide)
    order, stack = [], [root]
    while stack:
        k = stack.pop()
        order.append(k)
        if axis[k] >= 0:
            for c in (left[k], right[k]):
                if c >= 0:
                    stack.append(c)
    for k in reversed(order):
        if axis[k] < 0:
            made[k] = make_leaf([items[i].prim for i in leaf_items[left[k]:left[k] + right[k]]])
        else:
            made[k] = make_branch(int(axis[k]), float(split[k]), made.get(left[k]) if left[k] >= 0 else None,
                                  made.get(right[k]) if right[k] >= 0 else None)
    return lo, hi, made[root]


def build_tree_arrays(items, max_depth=KD_DEFAULT_MAX_DEPTH, split_threshold=KD_DEFAULT_SPLIT_THRESHOLD,
                      traversal_cost=0.0, intersection_cost=0.0):
    """The same tree as flat arrays: (lo, hi, node_axis, node_split, node_left, node_right, leaf_items, root); leaf:
    axis -1, left = first entry of leaf_items, right = count; leaf_items holds indices into `items`."""
    import ctypes as C
    from . import _lib
    if not items:
        raise ValueError("cannot build a k-d tree from zero primitives")
    n = len(items[0].lo)
    los = np.ascontiguousarray([it.lo for it in items], f32)
    his = np.ascontiguousarray([it.hi for it in items], f32)
    first = np.zeros(len(items) + 1, np.int32)
    verts = []
    for k, it in enumerate(items):
        if it.simplices is not None:
            verts.append(it.simplices)
            first[k + 1] = first[k] + len(it.simplices)
        else:
            first[k + 1] = first[k]
    sv = np.ascontiguousarray(np.concatenate(verts) if verts else np.zeros((0, n, n)), f32)
    out = _lib.NtKdTree()
    params = _lib.NtKdTreeParams(int(max_depth), int(split_threshold), float(traversal_cost), float(intersection_cost))
    _lib.check(_lib.lib().nt_kdtree_build(n, len(items), los.ctypes.data_as(_lib.f32p), his.ctypes.data_as(_lib.f32p),
                                          first.ctypes.data_as(_lib.i32p), sv.ctypes.data_as(_lib.f32p),
                                          C.byref(params), C.byref(out)))
    try:
        nn = out.n_nodes
        axis = np.ctypeslib.as_array(out.node_axis, (nn,)).copy()
        split = np.ctypeslib.as_array(out.node_split, (nn,)).copy()
        left = np.ctypeslib.as_array(out.node_left, (nn,)).copy()
        right = np.ctypeslib.as_array(out.node_right, (nn,)).copy()
        out_leaf_items = out.n_leaf_items
        leaf_items = np.ctypeslib.as_array(out.leaf_items, (max(out.n_leaf_items, 1),)).copy()
        box = np.ctypeslib.as_array(out.aabb, (2 * n,)).copy()
        root = out.root
    finally:
        _lib.lib().nt_kdtree_free(C.byref(out))
    return box[:n].astype(f32), box[n:].astype(f32), axis, split, left, right, leaf_items[:out_leaf_items], root
