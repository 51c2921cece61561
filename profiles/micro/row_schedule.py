import numpy as np

def bank_cost(tri, G=64):
    """sum over groups of 64 consecutive rows and the three columns of max(4, 2*max bank multiplicity)"""
    T = (len(tri) // G) * G
    b = (tri[:T] >> 1) & 63
    cost = 0
    for c in range(3):
        g = b[:, c].reshape(-1, G)
        for row in g:
            cost += max(4, 2 * np.bincount(row, minlength=64).max())
    return cost / (3 * (T // G))

def schedule_rows(tri, G=64, cand=24, seed=0):
    """Greedy: place rows one by one into the group (of `cand` random open ones) where they raise the bank
    loads least.  Returns a permutation of the rows."""
    rng = np.random.default_rng(seed)
    T = len(tri)
    ng = (T + G - 1) // G
    cap = np.full(ng, G); cap[-1] = T - G * (ng - 1)
    load = np.zeros((ng, 3, 64), dtype=np.int16)
    fill = np.zeros(ng, dtype=np.int32)
    members = [[] for _ in range(ng)]
    banks = (tri >> 1) & 63
    order = rng.permutation(T)
    for t in order:
        open_g = np.nonzero(fill < cap)[0]
        if len(open_g) > cand:
            open_g = rng.choice(open_g, cand, replace=False)
        b = banks[t]
        l = load[open_g, 0, b[0]] + load[open_g, 1, b[1]] + load[open_g, 2, b[2]]
        # prefer groups where no bank would exceed 2, then the fullest (to close groups)
        worst = np.maximum(np.maximum(load[open_g, 0, b[0]], load[open_g, 1, b[1]]), load[open_g, 2, b[2]])
        key = worst * 1000 + l * 10 - (fill[open_g] > 0)
        g = open_g[np.argmin(key)]
        members[g].append(t); fill[g] += 1
        load[g, 0, b[0]] += 1; load[g, 1, b[1]] += 1; load[g, 2, b[2]] += 1
    return np.concatenate([np.array(m, dtype=np.int64) for m in members])

if __name__ == "__main__":
    import sys; sys.path.insert(0, "/root/repo")
    from mvoscalerecovery_amd import synth
    from scipy.spatial import Delaunay
    import time
    f3, f2 = synth.synth_frame(0, 2000, base_seed=2024)
    tri = Delaunay(f2).simplices.astype(np.int32)
    print("rows", len(tri), "cost before %.2f" % bank_cost(tri))
    t = time.time(); p = schedule_rows(tri); dt = time.time() - t
    print("cost after %.2f (%.2f s)" % (bank_cost(tri[p]), dt), "perm ok", len(np.unique(p)) == len(tri))
