"""Design prototype (CPU, pure Python) of the grid-based per-point Delaunay stars of mvosr_delaunay.hip.

Not product code and not an oracle: it exists to pin the certification logic (which completions a lane may
finish from its 3x3 cell block, which points must go to the wave-parallel "hard" pass) and to measure the
statistics the kernel's design rests on — share of hard points, candidates examined per completion — before
the logic is written in HIP.  Checked against scipy.spatial.Delaunay (triangle sets).

    python profiles/micro/dt_proto.py [n] [frames] [points_per_cell]
"""
import math
import sys

import numpy as np

BLOCK_R = 1
TIE = 1e-9
COL = 1e-12
INF = float("inf")


class Grid:
    def __init__(self, P, per_cell):
        n = len(P)
        self.lo = P.min(0)
        self.hi = P.max(0)
        W, H = self.hi - self.lo
        s = math.sqrt(W * H * per_cell / n)
        self.gx = max(1, min(256, int(math.ceil(W / s))))
        self.gy = max(1, min(256, int(math.ceil(H / s))))
        self.ix = self.gx / W
        self.iy = self.gy / H
        self.sx = W / self.gx
        self.sy = H / self.gy
        cx = np.minimum(self.gx - 1, ((P[:, 0] - self.lo[0]) * self.ix).astype(int))
        cy = np.minimum(self.gy - 1, ((P[:, 1] - self.lo[1]) * self.iy).astype(int))
        cell = cy * self.gx + cx
        order = np.lexsort((np.arange(n), cell))
        self.oid = order                       # sorted index -> original id
        self.S = P[order]                      # sorted points
        self.cx, self.cy = cx[order], cy[order]
        cnt = np.bincount(cell, minlength=self.gx * self.gy)
        self.start = np.concatenate([[0], np.cumsum(cnt)])
        self.margin = 1e-9 * max(W, H)

    def rows(self, cxa, cxb, cya, cyb):
        """index ranges (in sorted order) of the cell box, one per cell row"""
        out = []
        for y in range(cya, cyb + 1):
            out.append((self.start[y * self.gx + cxa], self.start[y * self.gx + cxb + 1]))
        return out

    def cell_of(self, x, y):
        cx = int((x - self.lo[0]) * self.ix)
        cy = int((y - self.lo[1]) * self.iy)
        return cx, cy


def complete(S, ranges, i, iq, stats):
    """best apex left of the directed edge S[i] -> S[iq] among the points of `ranges`:
    returns (index or -1, tie, collinear_flag)"""
    px, py = S[i]
    ax, ay = S[iq][0] - px, S[iq][1] - py
    a2 = ax * ax + ay * ay
    n1 = c1 = n2 = c2 = 0.0
    b1 = -1
    have2 = False
    flag = False
    for (j0, j1) in ranges:
        for j in range(j0, j1):
            if j == i or j == iq:
                continue
            stats["cand"] += 1
            bx, by = S[j][0] - px, S[j][1] - py
            cr = ax * by - ay * bx
            b2 = bx * bx + by * by
            if cr * cr <= COL * COL * a2 * b2:
                if bx * ax + by * ay > 0.0 or b2 == 0.0:
                    flag = True
                continue
            if cr <= 0.0:
                continue
            num = bx * (bx - ax) + by * (by - ay)
            if b1 < 0 or num * c1 < n1 * cr:
                if b1 >= 0:
                    n2, c2, have2 = n1, c1, True
                n1, c1, b1 = num, cr, j
            elif not have2 or num * c2 < n2 * cr:
                n2, c2, have2 = num, cr, True
    tie = False
    if b1 >= 0 and have2:
        # t2 - t1 <= TIE * (|t1| + 1)   with t = n / c, c > 0
        tie = (n2 * c1 - n1 * c2) <= TIE * (abs(n1) + c1) * c2
    return b1, tie, flag


def circle(S, i, iq, ic):
    px, py = S[i]
    ax, ay = S[iq][0] - px, S[iq][1] - py
    bx, by = S[ic][0] - px, S[ic][1] - py
    cr = ax * by - ay * bx
    a2, b2 = ax * ax + ay * ay, bx * bx + by * by
    ox = (by * a2 - ay * b2) / (2.0 * cr)
    oy = (ax * b2 - bx * a2) / (2.0 * cr)
    return px + ox, py + oy, math.sqrt(ox * ox + oy * oy)


def star_lane(G, i, max_box_cells, stats, verify):
    """the lane's attempt: returns (status, neighbours CCW) with status 'ok' | 'hard:<why>' | 'degenerate'"""
    S = G.S
    px, py = S[i]
    cx, cy = G.cx[i], G.cy[i]
    R = BLOCK_R
    cxa, cxb = max(cx - R, 0), min(cx + R, G.gx - 1)
    cya, cyb = max(cy - R, 0), min(cy + R, G.gy - 1)
    X0 = G.lo[0] + (cx - R) * G.sx + G.margin if cx - R >= 0 else -INF
    X1 = G.lo[0] + (cx + R + 1) * G.sx - G.margin if cx + R <= G.gx - 1 else INF
    Y0 = G.lo[1] + (cy - R) * G.sy + G.margin if cy - R >= 0 else -INF
    Y1 = G.lo[1] + (cy + R + 1) * G.sy - G.margin if cy + R <= G.gy - 1 else INF
    block = G.rows(cxa, cxb, cya, cyb)
    # nearest neighbour
    best, bj = INF, -1
    for (j0, j1) in block:
        for j in range(j0, j1):
            if j == i:
                continue
            d2 = (S[j][0] - px) ** 2 + (S[j][1] - py) ** 2
            if d2 < best:
                best, bj = d2, j
    if bj < 0:
        return "hard:alone", []
    if best == 0.0:
        return "degenerate", []
    if not math.sqrt(best) <= min(px - X0, X1 - px, py - Y0, Y1 - py):
        return "hard:nn", []
    q0 = bj
    iq = q0
    nb = [q0]
    while True:
        stats["compl"] += 1
        c, tie, flag = complete(S, block, i, iq, stats)
        if flag:
            return "degenerate", []
        if c < 0:
            return "hard:open", []
        ox, oy, r = circle(S, i, iq, c)
        if not (ox - r >= X0 and ox + r <= X1 and oy - r >= Y0 and oy + r <= Y1):
            # the circle leaves the block: continue optimistically, a wave verifies the completion afterwards
            stats["ext"] += 1
            verify.append((i, iq, c, tie))
            tie = False
        if tie:
            return "degenerate", []
        iq = c
        if iq == q0:
            break
        nb.append(iq)
        if len(nb) > 24:
            return "hard:degree", []
    return "ok", nb


def star_full(S, i):
    """robust wave-pass equivalent: every completion scans all points; handles open stars"""
    n = len(S)
    allr = [(0, n)]
    st = {"cand": 0, "compl": 0, "ext": 0}
    px, py = S[i]
    d2 = (S[:, 0] - px) ** 2 + (S[:, 1] - py) ** 2
    d2[i] = INF
    q0 = int(np.argmin(d2))
    if d2[q0] == 0.0:
        return "degenerate", [], False
    ccw = [q0]
    iq = q0
    closed = False
    while True:
        c, tie, flag = complete(S, allr, i, iq, st)
        if flag or tie:
            return "degenerate", [], False
        if c < 0:
            break
        if c == q0:
            closed = True
            break
        ccw.append(c)
        iq = c
    if closed:
        return "ok", ccw, False
    # clockwise from q0: mirror (swap the roles: apex right of p -> q)
    cw = []
    iq = q0
    while True:
        c, tie, flag = complete_right(S, i, iq)
        if flag or tie:
            return "degenerate", [], False
        if c < 0:
            break
        cw.append(c)
        iq = c
    return "ok", cw[::-1] + ccw, True


def complete_right(S, i, iq):
    px, py = S[i]
    ax, ay = S[iq][0] - px, S[iq][1] - py
    a2 = ax * ax + ay * ay
    n1 = c1 = n2 = c2 = 0.0
    b1 = -1
    have2 = False
    flag = False
    for j in range(len(S)):
        if j == i or j == iq:
            continue
        bx, by = S[j][0] - px, S[j][1] - py
        cr = -(ax * by - ay * bx)
        b2 = bx * bx + by * by
        if cr * cr <= COL * COL * a2 * b2:
            if bx * ax + by * ay > 0.0 or b2 == 0.0:
                flag = True
            continue
        if cr <= 0.0:
            continue
        num = bx * (bx - ax) + by * (by - ay)
        if b1 < 0 or num * c1 < n1 * cr:
            if b1 >= 0:
                n2, c2, have2 = n1, c1, True
            n1, c1, b1 = num, cr, j
        elif not have2 or num * c2 < n2 * cr:
            n2, c2, have2 = num, cr, True
    tie = b1 >= 0 and have2 and (n2 * c1 - n1 * c2) <= TIE * (abs(n1) + c1) * c2
    return b1, tie, flag


def triangulate(P, per_cell=3.0, max_box_cells=25):
    n = len(P)
    G = Grid(P, per_cell)
    stats = {"cand": 0, "compl": 0, "ext": 0}
    why = {}
    rows = set()
    star_tris = 0
    hull = 0
    for i in range(n):
        verify = []
        st, nb = star_lane(G, i, max_box_cells, stats, verify)
        opened = False
        if st == "ok":
            for (vi, viq, vc, vtie) in verify:
                ox, oy, r = circle(G.S, vi, viq, vc)
                r *= 1.0 + 1e-12
                bxa, bya = G.cell_of(ox - r, oy - r)
                bxb, byb = G.cell_of(ox + r, oy + r)
                bxa, bxb = max(bxa, 0), min(bxb, G.gx - 1)
                bya, byb = max(bya, 0), min(byb, G.gy - 1)
                stats["vcells"] = stats.get("vcells", 0) + (bxb - bxa + 1) * (byb - bya + 1)
                stats["vreq"] = stats.get("vreq", 0) + 1
                vst = {"cand": 0, "compl": 0, "ext": 0}
                c2, tie2, flag2 = complete(G.S, G.rows(bxa, bxb, bya, byb), vi, viq, vst)
                stats["vcand"] = stats.get("vcand", 0) + vst["cand"]
                if flag2 or tie2:
                    st = "degenerate"
                    break
                if c2 != vc:
                    st = "hard:verify"
                    break
        if st != "ok":
            why[st] = why.get(st, 0) + 1
            if st == "degenerate":
                return None, stats, why, G
            st, nb, opened = star_full(G.S, i)
            if st != "ok":
                return None, stats, why, G
        k = len(nb)
        hull += opened
        for t in range(k if not opened else k - 1):
            a, b = nb[t], nb[(t + 1) % k]
            star_tris += 1
            tri = tuple(sorted((int(G.oid[i]), int(G.oid[a]), int(G.oid[b]))))
            if tri[0] == G.oid[i]:
                rows.add(tri)
    ok = len(rows) == 2 * n - 2 - hull and star_tris == 3 * len(rows)
    stats["hull"] = hull
    stats["euler"] = ok
    return rows, stats, why, G


def main():
    from scipy.spatial import Delaunay
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    per_cell = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
    box = 25
    global BLOCK_R
    BLOCK_R = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    for f in range(frames):
        rng = np.random.default_rng(100 + f)
        P = np.stack([rng.uniform(0, 1241, n), rng.uniform(186, 376, n)], axis=1)
        rows, stats, why, G = triangulate(P, per_cell, box)
        ref = set(tuple(sorted(map(int, r))) for r in Delaunay(P).simplices)
        hard = sum(v for k, v in why.items())
        print("n=%d cells=%dx%d same_set=%s euler=%s hull=%d hard=%d (%.1f%%) %s  cand/compl=%.1f compl/pt=%.2f ext/compl=%.3f vreq=%d vcells/req=%.1f vcand/req=%.1f" % (
            n, G.gx, G.gy, rows == ref, stats["euler"], stats["hull"], hard, 100.0 * hard / n, why,
            stats["cand"] / max(stats["compl"], 1), stats["compl"] / n, stats["ext"] / max(stats["compl"], 1), stats.get("vreq", 0), stats.get("vcells", 0) / max(stats.get("vreq", 0), 1), stats.get("vcand", 0) / max(stats.get("vreq", 0), 1)))


if __name__ == "__main__":
    main()
