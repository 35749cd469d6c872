"""Regular polytopes from their Schläfli symbol, as simplex scenes.

What the reference's ``scripts/polytope.py`` produces with its facet-propagation construction (:135-479, driver
:557-578) is generated here a different way -- a Wythoff / Coxeter-group construction:

  * the symbol {p1, p2, ...} (entries may be fractions p/q for star polytopes) fixes the angles pi*q/p between
    consecutive mirrors of a linear Coxeter diagram; the mirror normals are written down in lower-triangular form;
  * the vertices are the orbit of the point lying on every mirror but the first; the k-faces are the orbits of the
    base flag's faces, kept as sets of vertex indices (no fuzzy point matching, no hash-order dependence: the
    output is deterministic, unlike the script's);
  * every facet is cut into simplices: a convex face by pulling its lowest vertex (fan triangulation, recursively),
    a star face -- whose own symbol has a fractional entry -- by coning its sub-faces from its centre; those cones
    overlap where the star's density exceeds 1, which a nearest-hit ray caster does not see (same plane, same
    material, same distance).

Position, scale and orientation equal the reference's: the polytope is centred on the origin, its 2-faces have
inradius 1 (``Polygon.apothem``, :136), and the base flag's centres step along +e_0, +e_1, ... (:88-89, :472).
The reference additionally inflates every propagated facet by 1.00001 (:319-322); this module does not, so
silhouettes can differ by that much.

Only numpy; runs once per scene on the host.
"""
import fractions
import math

import numpy as np


def schlafli_component(x):
    """'5', '5/2' or a Fraction -> Fraction, with the reference's argument checks (scripts/polytope.py:29-41)."""
    if isinstance(x, fractions.Fraction):
        f = x
    elif isinstance(x, int):
        f = fractions.Fraction(x)
    else:
        num, _, den = str(x).partition("/")
        p = int(num, 10)
        if p < 3:
            raise ValueError("a component cannot be less than 3")
        if not den:
            return fractions.Fraction(p)
        s = int(den, 10)
        if s < 1:
            raise ValueError("for component p/q: q cannot be less than 1")
        if s >= p:
            raise ValueError("for component p/q: q must be less than p")
        if math.gcd(s, p) != 1:
            raise ValueError("for component p/q: p and q must be co-prime")
        return fractions.Fraction(p, s)
    if f.numerator < 3 or f.denominator < 1 or f.denominator >= f.numerator:
        raise ValueError("invalid Schläfli component %s" % f)
    return f


def star_component(x):
    """scripts/polytope.py:113-114"""
    return (x.numerator - 1) > x.denominator > 1


def is_hypercube(schlafli):
    """{4,3,...,3}: the driver renders these with BoxScene instead of a mesh (scripts/polytope.py:557-559)."""
    return len(schlafli) >= 2 and schlafli[0] == 4 and all(c == 3 for c in schlafli[1:])


class RegularPolytope(object):
    """The regular polytope {schlafli} of rank r = len(schlafli)+1, embedded in max(r,3) dimensions."""

    def __init__(self, schlafli, max_vertices=200000):
        self.schlafli = [schlafli_component(c) for c in schlafli]
        if not self.schlafli:
            raise ValueError("at least one Schläfli component is required")
        r = len(self.schlafli) + 1
        self.rank = r
        self.dimension = max(r, 3)
        # ---- mirrors: m_0 = e_0, m_i in span(e_{i-1}, e_i), m_i . m_{i-1} = -cos(pi/p_i)
        m = np.zeros((r, r))
        m[0, 0] = 1.0
        for i in range(1, r):
            p = self.schlafli[i - 1]
            a = -math.cos(math.pi * p.denominator / p.numerator) / m[i - 1, i - 1]
            rest = 1.0 - a * a
            if rest <= 1e-12:
                raise ValueError("Component #%d (%s) is invalid because the angles of the parts add up to 360° or "
                                 "more and thus can't be folded inward" % (i, p))
            m[i, i - 1] = a
            m[i, i] = math.sqrt(rest)
        self.mirrors = m
        # ---- the base vertex lies on every mirror but the first
        v0 = np.zeros(r)
        v0[0] = 1.0
        for i in range(1, r):
            v0[i] = -m[i, i - 1] * v0[i - 1] / m[i, i]
        # ---- vertex orbit, and what every generator does to the vertices
        verts = [v0]
        perms = [[] for _ in range(r)]
        head = 0
        while head < len(verts):
            x = verts[head]
            for g in range(r):
                y = x - 2.0 * (m[g] @ x) * m[g]
                d = np.abs(np.asarray(verts) - y).max(axis=1)
                k = int(np.argmin(d))
                if d[k] > 1e-6:
                    verts.append(y)
                    k = len(verts) - 1
                    if k >= max_vertices:
                        raise ValueError("the symbol does not describe a finite polytope (more than %d vertices)" % max_vertices)
                perms[g].append(k)
            head += 1
        self._perms = np.asarray(perms, dtype=np.int64)          # [generator][vertex]
        verts = np.asarray(verts)
        # ---- faces: orbits of the base flag's faces (as vertex sets)
        self.faces = [None] * (r + 1)                             # faces[k]: list of sorted vertex tuples
        self.faces[0] = [(i,) for i in range(len(verts))]
        for k in range(1, r + 1):
            base = self._orbit_of_vertex(range(min(k, r)))
            self.faces[k] = self._orbit_of_set(base) if k < r else [tuple(sorted(base))]
        # ---- scale: the 2-faces have inradius 1 (distance between the base edge's and the base 2-face's centres)
        c1 = verts[list(self.faces[1][0])].mean(axis=0)
        c2 = verts[list(self._orbit_of_vertex(range(2)))].mean(axis=0)
        self.vertices = np.zeros((len(verts), self.dimension))
        self.vertices[:, :r] = verts / np.linalg.norm(c1 - c2)
        self._sub = {}
        self._simp = {}

    # ---- group bookkeeping
    def _orbit_of_vertex(self, gens):
        seen = {0}
        todo = [0]
        while todo:
            v = todo.pop()
            for g in gens:
                w = int(self._perms[g][v])
                if w not in seen:
                    seen.add(w)
                    todo.append(w)
        return frozenset(seen)

    def _orbit_of_set(self, base):
        first = tuple(sorted(base))
        seen = {first: None}
        todo = [first]
        while todo:
            f = todo.pop()
            for g in range(self.rank):
                h = tuple(sorted(self._perms[g][list(f)].tolist()))
                if h not in seen:
                    seen[h] = None
                    todo.append(h)
        return sorted(seen)

    # ---- geometry
    def circumradius(self):
        return float(np.linalg.norm(self.vertices[0]))

    def circumradius_square(self):
        return float(self.vertices[0] @ self.vertices[0])

    def _face_is_star(self, k):
        return any(star_component(c) for c in self.schlafli[:max(k - 1, 0)])

    def _subfaces(self, k, f):
        key = (k, f)
        if key not in self._sub:
            fs = set(f)
            self._sub[key] = [g for g in self.faces[k - 1] if fs.issuperset(g)]
        return self._sub[key]

    def _simplices(self, k, f):
        """The k-face f as a list of k-simplices, each a list of k+1 points."""
        key = (k, f)
        if key in self._simp:
            return self._simp[key]
        V = self.vertices
        if k == 1:
            out = [[V[f[0]], V[f[1]]]]
        elif self._face_is_star(k):
            c = V[list(f)].mean(axis=0)
            out = [[c] + s for g in self._subfaces(k, f) for s in self._simplices(k - 1, g)]
        else:
            v = f[0]
            out = [[V[v]] + s for g in self._subfaces(k, f) if v not in g for s in self._simplices(k - 1, g)]
        self._simp[key] = out
        return out

    def is_star(self):
        return any(star_component(c) for c in self.schlafli)

    def simplices(self, max_edge=None):
        """The boundary as (count, n, n) float32: n-1-simplices with n vertices each, n = self.dimension.  For a
        polygon (rank 2, shown in 3-D) the polygon itself is returned, as in the reference (Polygon.hull, :177-181).

        max_edge: bisect simplices until no edge OPPOSITE THE FIRST VERTEX is longer than this.  The cones of a
        star facet (first vertex = the facet's centre) span the whole facet; cut into needles they can be told
        apart by a k-d tree.  Default: half the circumradius for star polytopes, no subdivision otherwise."""
        k = self.rank - 1 if self.rank >= 3 else 2
        out = []
        for f in self.faces[k]:
            out.extend(self._simplices(k, f))
        s = np.asarray(out, np.float64)
        if max_edge is None and self.is_star():
            max_edge = 0.5 * self.circumradius()
        if max_edge:
            s = bisect_bases(s, float(max_edge))
        return s.astype(np.float32)

    def hull(self, nt, material, max_edge=None):
        """-> list of nt.TrianglePrototype, what ``p.hull()`` is in the reference's driver (:569)."""
        if nt.dimension != self.dimension:
            raise ValueError("the polytope needs an NTracer of dimension %d" % self.dimension)
        from . import tracern
        return tracern.triangle_prototypes(self.simplices(max_edge), material)


def bisect_bases(simplices, max_edge):
    """Longest-edge bisection of the face opposite vertex 0 of every simplex ((count, m, n) array) until none of
    its edges exceeds max_edge; the pieces tile the original simplices exactly."""
    m = simplices.shape[1]
    done = []
    work = np.asarray(simplices, np.float64)
    while len(work):
        base = work[:, 1:, :]
        d = base[:, :, None, :] - base[:, None, :, :]
        length = np.sqrt((d * d).sum(-1)).reshape(len(work), -1)
        big = length.max(axis=1) > max_edge
        done.append(work[~big])
        w = work[big]
        if not len(w):
            break
        i, j = np.divmod(length[big].argmax(axis=1), m - 1)
        rows = np.arange(len(w))
        mid = (w[rows, i + 1] + w[rows, j + 1]) * 0.5
        a = w.copy()
        a[rows, i + 1] = mid
        b = w.copy()
        b[rows, j + 1] = mid
        work = np.concatenate([a, b])
    return np.concatenate(done)


def build_scene(schlafli, material=None, cam_dist=4.0, nt=None):
    """What the reference's driver builds for a symbol (scripts/polytope.py:557-586): returns
    (nt, scene, cam_distance); the camera is NOT set.  {4,3,..,3} gives a BoxScene, as there."""
    from . import Material, NTracer
    comps = [schlafli_component(c) for c in schlafli]
    n = max(len(comps) + 1, 3)
    nt = nt or NTracer(n)
    if is_hypercube(comps):
        return nt, nt.BoxScene(), -math.sqrt(n) * cam_dist
    p = RegularPolytope(comps)
    scene = nt.build_composite_scene(p.hull(nt, material or Material((1, 0.5, 0.5))))
    return nt, scene, -math.sqrt(p.circumradius_square()) * cam_dist


def rotating_cameras(nt, cam_distance, frames=160, jitter=True):
    """The camera path of the reference's animation / benchmark loop (RotatingCamera, scripts/polytope.py:522-556;
    first camera :586-587): yields ``frames`` cameras (nt.Camera copies)."""
    n = nt.dimension
    cam = nt.Camera()
    j = nt.Vector([0, 0, 0] + [0.0001] * (n - 3)) if jitter else nt.Vector([0] * n)
    cam.translate(nt.Vector.axis(2, cam_distance) + j)
    incr = 2 * math.pi / frames
    h = 1 / math.sqrt(n - 1)
    for f in range(frames):
        c = nt.Camera()
        c.origin = cam.origin
        for i in range(n):
            c.axes[i] = cam.axes[i]
        yield c
        a2 = cam.axes[0] * h + cam.axes[1] * h
        for i in range(n - 3):
            a2 = a2 + cam.axes[i + 3] * h
        cam.transform(nt.Matrix.rotation(cam.axes[2], a2, incr))
        cam.normalize()
        cam.origin = cam.axes[2] * cam_distance
