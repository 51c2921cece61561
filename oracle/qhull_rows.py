"""TEST INFRASTRUCTURE ONLY (imported by tests/, profiles/ probes; never by the product path).

Restatement of what `scipy.spatial.Delaunay(points2d).simplices` IS, row for row, for the call sites
/root/reference/src/scale_calculator.py:257,:266 (also src/rescale.py:124,:136): SciPy's bundled Qhull — a THIRD-PARTY
dependency absent from /root/reference and un-pinned there; here qhull_r 7.3.2 (2019.1.r 2019/06/21) inside SciPy 1.15.3, run
as `qhull d Qbb Qc Qz Q12 Qt` — builds the convex hull of the sites lifted to a paraboloid with Quickhull's beneath-beyond,
and SciPy emits one row per lower facet, in facet-list order, as the facet's vertices by decreasing vertex id (= reverse
insertion order) with the first two swapped where needed to make the row counter-clockwise.  The reference's `check_triangle`
(src/scale_calculator.py:105-119) reads the rotation of each row, so its vote is a function of Qhull's INSERTION ORDER.

This file restates the published algorithm (Barber, Dobkin, Huhdanpaa, "The Quickhull algorithm for convex hulls", ACM TOMS
1996, and Qhull's documented options) for this one configuration: 2-d sites, general position (no merged facets, no coplanar
points, no narrow initial simplex).  A frame that leaves that regime raises `Declined` — the same contract the device
triangulation has.  Pinned: `tests/test_qhull_rows_oracle.py` checks rows == SciPy's rows (order and rotation) on
seeded frames, and `profiles/qhull_trace.py` checks the insertion order itself against Qhull's own trace output.

Steps (each is a decision the order depends on):
  1. lift: z = x*x + y*y; the point "at infinity" of 'Qz' = (mean x, mean y, 1.1 * max z), id n;
  2. extreme points per coordinate (first strict maximum / minimum in input order, maximum tested first);
  3. 'Qbb': the last coordinate scaled to [0, max |x or y|];
  4. roundoff constants from the coordinate ranges (DISTround; MINvisible = 2 DISTround; MINoutside = 4 DISTround);
  5. initial simplex: the extreme points of min x and max x, then twice the extreme point with the largest |determinant|;
  6. four facets, each omitting one vertex, orientation alternating, flipped as a whole if the centre lies outside the first;
  7. every other point to the FIRST facet (list order) it lies above by 8 DISTround; a facet keeps its furthest point last;
  8. the facet with the furthest point of all moves to the head of the list;
  9. loop: first facet in list order with an outside set -> its furthest point; visible facets by breadth-first search in
     neighbour order (neighbour i is opposite vertex i); for each visible facet in that order, for each horizon neighbour in
     neighbour order, a new facet (apex first) at the END of the list; the visible facets' points go to the first new facet
     that a directed walk from the visible facet's replacement finds them MINoutside above; visible facets deleted.
"""
import math

import numpy as np

EPS = 2.220446049250313e-16


class Declined(Exception):
    """The frame leaves the general-position regime this restatement covers (Qhull would merge facets or joggle)."""


class _Facet:
    __slots__ = ("id", "v", "nb", "top", "n0", "n1", "n2", "off", "upper", "out", "fdist", "visible", "new", "visit", "replace",
                 "prev", "next", "dead")

    def __init__(self, fid):
        self.id = fid
        self.v = None
        self.nb = []
        self.top = False
        self.out = None
        self.fdist = 0.0
        self.visible = False
        self.new = False
        self.visit = 0
        self.replace = None
        self.prev = None
        self.next = None
        self.upper = False
        self.dead = False


class QhullDelaunay2D:
    def __init__(self, points2d, record=False):
        P = np.ascontiguousarray(points2d, dtype=np.float64)
        if P.ndim != 2 or P.shape[1] != 2 or len(P) < 3:
            raise Declined("need (n >= 3, 2) points")
        self.n = n = len(P)
        self.record = record
        self.events = []
        # 1. lift
        xs, ys = P[:, 0].tolist(), P[:, 1].tolist()
        zs = [x * x + y * y for x, y in zip(xs, ys)]
        sx = sy = 0.0
        maxb = -1.797e308
        for i in range(n):
            sx += xs[i]
            sy += ys[i]
            if zs[i] > maxb:
                maxb = zs[i]
        xs.append(sx / n)
        ys.append(sy / n)
        zs.append(maxb * 1.1)
        self.x, self.y, self.z = xs, ys, zs
        m = n + 1
        # 2. extreme points, ranges
        coords = (xs, ys, zs)
        maxpoints = []
        maxabs = 0.0
        maxwidth = 0.0
        maxsum = 0.0
        self.nearzero = []
        for k in range(3):
            c = coords[k]
            lo = hi = 0
            for i in range(m):
                if c[hi] < c[i]:
                    hi = i
                elif c[lo] > c[i]:
                    lo = i
            if k == 2:
                zlow, zhigh = c[lo], c[hi]
                maxcoord = maxabs
            else:
                maxcoord = max(c[hi], -c[lo])
                maxwidth = max(maxwidth, c[hi] - c[lo])
            maxabs = max(maxabs, maxcoord)
            maxsum += maxcoord
            maxpoints += [lo, hi]
            self.nearzero.append(80 * maxsum * EPS)
        if maxwidth <= 0.0:
            raise Declined("zero width")
        # 3. Qbb
        scale = maxabs / (zhigh - zlow)
        shift = 0.0 - zlow * scale
        self.z = zs = [z * scale + shift for z in zs]
        self.scale, self.shift = scale, shift
        # 4. roundoff
        maxdistsum = min(math.sqrt(3.0) * maxabs, maxsum)
        self.distround = EPS * (3 * maxdistsum * 1.01 + maxabs)
        self.anground = 1.01 * 3 * EPS
        self.minvisible = 2 * self.distround
        self.maxcoplanar = self.minvisible
        self.minoutside = 2 * self.minvisible
        self.distoutside = 2 * self.minoutside
        self.maxwidth = maxwidth
        self.guard = 64 * self.distround          # decisions closer than this to a threshold: not covered
        # 5. initial simplex
        simplex = self._maxsimplex(maxpoints)
        self.simplex = simplex
        # 6. vertices, facets
        self.vpoint = [None] + list(simplex)       # vertex id -> point id (ids from 1)
        self.pvertex = {p: i + 1 for i, p in enumerate(simplex)}
        self.order = list(simplex)                 # insertion order pi (initial simplex first)
        self.nfacets = 0
        self.visit_id = 0
        self.tail = _Facet(0)
        self.head = self.tail
        self.facet_next = self.tail
        verts = [4, 3, 2, 1]
        fs = []
        top = True
        for i in range(4):
            f = self._newfacet()
            f.v = [v for j, v in enumerate(verts) if j != i]
            f.top = top
            top = not top
            self._append(f)
            fs.append(f)
        for f in fs:
            f.nb = [g for g in fs if g is not f]
        cx = sum(xs[p] for p in simplex[::-1]) / 4
        cy = sum(ys[p] for p in simplex[::-1]) / 4
        cz = sum(zs[p] for p in simplex[::-1]) / 4
        # interior point: Qhull sums in vertex-set order (descending id)
        cx = cy = cz = 0.0
        for v in verts:
            p = self.vpoint[v]
            cx += xs[p]
            cy += ys[p]
            cz += zs[p]
        cx, cy, cz = cx / 4, cy / 4, cz / 4
        self._plane(fs[0])
        d = fs[0].off + cx * fs[0].n0 + cy * fs[0].n1 + cz * fs[0].n2
        if d > self.distround:
            for f in fs:
                f.top = not f.top
            self._plane(fs[0])
        elif d > -self.distround:
            raise Declined("flat initial simplex")
        for f in fs[1:]:
            self._plane(f)
        # narrow hull test (cosine between facet normals)
        minangle = 2.0
        for f in fs:
            for g in f.nb:
                a = f.n0 * g.n0 + f.n1 * g.n1 + f.n2 * g.n2
                minangle = min(minangle, a)
        if minangle < -0.99999999:
            raise Declined("narrow initial simplex")
        # 7. partition all
        self._partition_all()
        # 8. furthest facet first
        best, bestd = None, -1.797e308
        f = self.head
        while f is not self.tail:
            if f.out is not None and f.fdist > bestd:
                best, bestd = f, f.fdist
            f = f.next
        self.facet_next = self.head
        if best is not None:
            self._remove(best)
            self._prepend(best, self.facet_next)
            self.facet_next = best
            self.head = best if best.prev is None else self.head
        # 9. build
        self._build()

    # ---- list plumbing (Qhull's facet list: order is part of the algorithm) ----
    def _newfacet(self):
        self.nfacets += 1
        return _Facet(self.nfacets)

    def _append(self, f):
        t = self.tail
        f.prev = t.prev
        f.next = t
        if t.prev is not None:
            t.prev.next = f
        else:
            self.head = f
        t.prev = f
        if self.facet_next is t:
            self.facet_next = f

    def _remove(self, f):
        if f is self.facet_next:
            self.facet_next = f.next
        if f.prev is not None:
            f.prev.next = f.next
        else:
            self.head = f.next
        f.next.prev = f.prev
        f.prev = f.next = None

    def _prepend(self, f, before):
        f.prev = before.prev
        f.next = before
        if before.prev is not None:
            before.prev.next = f
        else:
            self.head = f
        before.prev = f

    # ---- geometry, in Qhull's order of operations ----
    def _plane(self, f):
        x, y, z = self.x, self.y, self.z
        p0, p1, p2 = (self.vpoint[v] for v in f.v)
        dx1, dy1, dz1 = x[p1] - x[p0], y[p1] - y[p0], z[p1] - z[p0]
        dx2, dy2, dz2 = x[p2] - x[p0], y[p2] - y[p0], z[p2] - z[p0]
        n0 = dy2 * dz1 - dz2 * dy1
        n1 = dx1 * dz2 - dz1 * dx2
        n2 = dx2 * dy1 - dy2 * dx1
        norm = math.sqrt(n0 * n0 + n1 * n1 + n2 * n2)
        if not norm > 1e-290:
            raise Declined("degenerate facet")
        if not f.top:
            norm = -norm
        n0, n1, n2 = n0 / norm, n1 / norm, n2 / norm
        f.n0, f.n1, f.n2 = n0, n1, n2
        f.off = -(x[p0] * n0 + y[p0] * n1 + z[p0] * n2)
        gauss = False
        for p in (p2, p1):
            d = f.off + (x[p] * n0 + y[p] * n1 + z[p] * n2)
            if d > self.distround or d < -self.distround:
                gauss = True
                break
        if gauss:
            # qh_sethyperplane_gauss: Gaussian elimination with partial pivoting on the two edge vectors, back substitution
            # from normal[2] = -+1, positive normalisation; the sign follows the row swaps and the signs of the pivots
            r0 = [dx1, dy1, dz1]
            r1 = [dx2, dy2, dz2]
            sign = bool(f.top)
            if abs(r1[0]) > abs(r0[0]):
                r0, r1 = r1, r0
                sign = not sign
            if abs(r0[0]) <= self.nearzero[0]:
                raise Declined("near-zero pivot")
            q = r1[0] / r0[0]
            r1[1] -= q * r0[1]
            r1[2] -= q * r0[2]
            if abs(r1[1]) <= self.nearzero[1]:
                raise Declined("near-zero pivot")
            if r1[1] < 0:
                sign = not sign
            if r0[0] < 0:
                sign = not sign
            n2 = -1.0 if sign else 1.0
            n1 = 0.0
            n1 -= r1[2] * n2
            n1 /= r1[1]
            n0 = 0.0
            n0 -= r0[1] * n1
            n0 -= r0[2] * n2
            n0 /= r0[0]
            norm = math.sqrt(n0 * n0 + n1 * n1 + n2 * n2)
            n0, n1, n2 = n0 / norm, n1 / norm, n2 / norm
            f.n0, f.n1, f.n2 = n0, n1, n2
            off = -(x[p0] * n0)
            off -= y[p0] * n1
            off -= z[p0] * n2
            f.off = off
        f.upper = n2 > -self.anground * 2.0

    def _dist(self, p, f):
        return f.off + self.x[p] * f.n0 + self.y[p] * f.n1 + self.z[p] * f.n2

    def _det(self, apex, pts, dim):
        x, y, z = self.x, self.y, self.z
        if dim == 2:
            a, b = pts[0], pts[1]
            r00, r01 = x[a] - x[apex], y[a] - y[apex]
            r10, r11 = x[b] - x[apex], y[b] - y[apex]
            det = r00 * r11 - r01 * r10
            return det, abs(det) < 10 * self.nearzero[1]
        a, b, c = pts[0], pts[1], pts[2]
        a1, a2, a3 = x[a] - x[apex], y[a] - y[apex], z[a] - z[apex]
        b1, b2, b3 = x[b] - x[apex], y[b] - y[apex], z[b] - z[apex]
        c1, c2, c3 = x[c] - x[apex], y[c] - y[apex], z[c] - z[apex]
        det = a1 * (b2 * c3 - b3 * c2) - b1 * (a2 * c3 - a3 * c2) + c1 * (a2 * b3 - a3 * b2)
        return det, abs(det) < 10 * self.nearzero[2]

    def _maxsimplex(self, maxpoints):
        x = self.x
        maxc, minc = -1.797e308, 1.797e308
        maxx = minx = None
        for p in maxpoints:
            if maxc < x[p]:
                maxc, maxx = x[p], p
            if minc > x[p]:
                minc, minx = x[p], p
        simplex = [minx]
        if maxx not in simplex:
            simplex.append(maxx)
        if len(simplex) < 2:
            raise Declined("one extreme point")
        maxdet = maxc - minc
        for i in range(2, 4):
            prevdet = maxdet
            maxpoint, maxdet, maxnear = None, -1.0, False
            for p in maxpoints:
                if p not in simplex:
                    det, near = self._det(p, simplex, i)
                    det = abs(det)
                    if det > maxdet:
                        maxdet, maxpoint, maxnear = det, p, near
            targetdet = prevdet * self.maxwidth
            if maxpoint is None or maxnear or (maxdet > 0.0 and maxdet / targetdet < 0.02):
                # Qhull searches all points here (qh_RATIOmaxsimplex); outside the regime this restatement is pinned on
                raise Declined("initial simplex needs the all-points search")
            simplex.append(maxpoint)
        return simplex

    # ---- partition ----
    def _partition_all(self):
        n = self.n
        pointset = [p for p in range(n + 1) if p not in self.pvertex]
        f = self.head
        while f is not self.tail:
            rest = []
            out = []
            best, bestd = None, 0.0
            for p in pointset:
                d = self._dist(p, f)
                if d < self.distoutside:
                    rest.append(p)
                    if d > -self.guard and d > self.distoutside - 2 * self.guard:
                        raise Declined("point within roundoff of an initial facet")
                else:
                    if best is None:
                        best, bestd = p, d
                    elif d > bestd:
                        out.append(best)
                        best, bestd = p, d
                    else:
                        out.append(p)
            if best is not None:
                out.append(best)
                f.out = out
                f.fdist = bestd
            pointset = rest
            f = f.next
        if pointset:
            raise Declined("points inside the initial simplex")

    def _add_outside(self, f, p, d):
        if f.out is None or not f.out:
            f.out = [p]
            f.fdist = d
        elif f.fdist < d:
            f.out.append(p)
            f.fdist = d
        else:
            f.out.insert(len(f.out) - 1, p)

    def _partition_point(self, p, start):
        """qh_partitionpoint -> qh_findbest(isnewfacets): a directed walk over the new facets; the first facet found
        MINoutside above wins.  A walk that ends below its best facet asks whether the cone is 'sharp' (normals in more than
        one orthant): then this point and every later one of this insertion scan the new facets in list order instead."""
        if self.findbestnew:
            return self._findbestnew(p, start)
        self.visit_id += 1
        vid = self.visit_id
        d = self._dist(p, start)
        self._band(d)
        if d >= self.minoutside:
            return start, d
        bestd = d
        best = None if start.upper else start
        start.visit = vid
        f = start
        while f is not None:
            nxt = None
            for g in f.nb:
                if not g.new or g.visit == vid:
                    continue
                g.visit = vid
                d = self._dist(p, g)
                self._band(d)
                if d > bestd:
                    if d >= self.minoutside:
                        return g, d
                    if not g.upper:
                        best, bestd = g, d
                        nxt = g
                        break
                    elif best is None:
                        bestd = d
                        nxt = g
                        break
            f = nxt
        testhorizon = True
        if best is None:
            return self._findbestnew(p, self.newlist)
        if not self.notsharp and bestd < -self.distround:
            if self._sharp():
                self.findbestnew = True
                return self._findbestnew(p, best)
            self.notsharp = True
        best, bestd = self._findbesthorizon(p, best, bestd)
        if bestd < self.minoutside:
            raise Declined("point above no facet")
        return best, bestd

    def _band(self, d):
        if -self.guard < d < self.guard:
            raise Declined("partition decision within roundoff")

    def _sharp(self):
        f = self.newlist
        q = (f.n0 > 0, f.n1 > 0, f.n2 > 0)
        f = f.next
        while f is not self.tail:
            if q != (f.n0 > 0, f.n1 > 0, f.n2 > 0):
                return True
            f = f.next
        return False

    def _findbestnew(self, p, start):
        """qh_findbestnew: the new facets in list order from `start` (wrapping to the first new facet); the first one
        2 MINoutside above wins."""
        self.visit_id += 1
        vid = self.visit_id
        best, bestd = None, -1.797e308
        for i in range(2):
            f = start if i == 0 else self.newlist
            while f is not self.tail:
                if f is start and i:
                    break
                f.visit = vid
                d = self._dist(p, f)
                self._band(d)
                if d > bestd and (not f.upper or d >= self.minoutside):
                    best = f
                    if d >= self.distoutside:
                        return f, d
                    bestd = d
                f = f.next
        best, bestd = self._findbesthorizon(p, best if best is not None else start, bestd)
        if bestd < self.minoutside:
            raise Declined("point above no facet")
        return best, bestd

    def _findbesthorizon(self, p, start, bestd):
        """qh_findbesthorizon(!ischeckmax, noupper = False): climb through ALL neighbours (old facets too) from `start`."""
        self.visit_id += 1
        vid = self.visit_id
        best = start
        searchdist = 4 * self.distround
        minsearch = bestd - searchdist
        stack = []
        start.visit = vid
        f = start
        nextfacet = None
        while True:
            for g in f.nb:
                if g.visit == vid:
                    continue
                g.visit = vid
                d = self._dist(p, g)
                self._band(d)
                if d > bestd:
                    if not g.upper or d >= self.minoutside:
                        minsearch = d - searchdist
                        if d > bestd + searchdist:
                            stack = []
                        best, bestd = g, d
                elif d < minsearch:
                    continue
                if nextfacet is not None:
                    stack.append(nextfacet)
                nextfacet = g
            f = nextfacet
            if f is not None:
                nextfacet = None
            elif not stack:
                break
            else:
                f = stack.pop() if len(stack) > 1 else stack.pop(0)
        return best, bestd

    # ---- the loop ----
    def _build(self):
        tail = self.tail
        x, y, z = self.x, self.y, self.z
        while True:
            f = self.facet_next
            while f is not tail and not f.out:
                f.out = None
                f = f.next
            self.facet_next = f
            if f is tail:
                break
            p = f.out.pop()
            if self.record:
                self.events.append((p, f.id, f.fdist))
            # horizon
            self._remove(f)
            self._append(f)
            f.visible = True
            f.replace = None
            visible = [f]
            self.visit_id += 1
            vid = self.visit_id
            i = 0
            while i < len(visible):
                vis = visible[i]
                i += 1
                vis.visit = vid
                for g in vis.nb:
                    if g.visit == vid:
                        continue
                    g.visit = vid
                    d = self._dist(p, g)
                    if d > self.minvisible:
                        if d < self.guard:
                            raise Declined("visibility within roundoff")
                        self._remove(g)
                        self._append(g)
                        g.visible = True
                        g.replace = None
                        visible.append(g)
                    elif d >= -self.guard:
                        raise Declined("coplanar horizon")
            # cone
            apex = len(self.vpoint)
            self.vpoint.append(p)
            self.pvertex[p] = apex
            self.order.append(p)
            newf = []
            self.newlist = None
            for vis in visible:
                last = None
                for g in vis.nb:
                    if g.visible:
                        continue
                    skip = 0 if g.nb[0] is vis else (1 if g.nb[1] is vis else 2)
                    nf = self._newfacet()
                    nf.v = [apex] + [v for j, v in enumerate(g.v) if j != skip]
                    nf.top = bool(skip & 1) if g.top else not (skip & 1)
                    nf.nb = [g, None, None]
                    nf.new = True
                    self._append(nf)
                    g.nb[skip] = nf
                    newf.append(nf)
                    last = nf
                if last is not None:
                    vis.replace = last
            self.newlist = newf[0]
            # match: neighbour k (k = 1, 2) shares the ridge without vertex k
            ridge = {}
            for nf in newf:
                for k in (1, 2):
                    key = nf.v[3 - k]                     # the horizon vertex that stays in the ridge {apex, key}
                    other = ridge.pop(key, None)
                    if other is None:
                        ridge[key] = (nf, k)
                    else:
                        nf.nb[k] = other[0]
                        other[0].nb[other[1]] = nf
            if ridge:
                raise Declined("open cone")
            for nf in newf:
                self._plane(nf)
            # convexity of the cone (Qhull would merge): the vertex of each neighbour opposite the shared ridge must lie below
            for nf in newf:
                for k in range(3):
                    g = nf.nb[k]
                    j = 0 if g.nb[0] is nf else (1 if g.nb[1] is nf else 2)
                    q = self.vpoint[g.v[j]]
                    if self._dist(q, nf) > -self.guard:
                        raise Declined("cone not strictly convex (merge)")
            # partition the visible facets' points
            self.findbestnew = False
            self.notsharp = False
            for vis in visible:
                if not vis.out:
                    continue
                start = vis.replace if vis.replace is not None else self.newlist
                for q in vis.out:
                    g, d = self._partition_point(q, start)
                    if not g.out and not g.new:            # an old facet takes a point: Qhull moves it behind facet_next
                        self._remove(g)
                        self._append(g)
                    self._add_outside(g, q, d)
            for vis in visible:
                self._remove(vis)
                vis.dead = True
            for nf in newf:
                nf.new = False

    # ---- output ----
    def simplices(self):
        """SciPy's rows: lower facets in list order; vertices by decreasing vertex id, first two swapped when NOT top-oriented."""
        rows = []
        f = self.head
        while f is not self.tail:
            if not f.upper:
                a, b, c = (self.vpoint[v] for v in f.v)
                rows.append((a, b, c) if f.top else (b, a, c))
            f = f.next
        return np.array(rows, dtype=np.int32).reshape(-1, 3)


def delaunay_rows(points2d):
    return QhullDelaunay2D(points2d).simplices()
