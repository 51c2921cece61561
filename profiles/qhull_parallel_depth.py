#!/usr/bin/env python3
"""How sequential is Qhull's run really?  (Groundwork for a parallel replay: SURVEY §8 row f / DESIGN §6.)
Qhull's facet list is a FIFO — new facets go to the tail, the next insertion takes the first facet with an outside set — so the
run proceeds generation by generation (a facet's generation = its creator's + 1), and two insertions of one generation whose
regions (visible + horizon facets) share no facet commute: their relative order only decides vertex ids and list positions, which
follow from the creators' order afterwards.  This script replays the run with oracle/qhull_rows.py, records every insertion's
region, and counts what a deterministic-reservations schedule would need: generations, and per generation the longest chain of
insertions that conflict pairwise in list order (rounds).  Build-container tool (imports oracle/).
   python profiles/qhull_parallel_depth.py [points] [frames]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mvoscalerecovery_amd import synth      # noqa: E402

src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "qhull_rows.py")).read()
# record, per insertion: the processed facet, the visible facets, the horizon facets, the new facets
src = src.replace("            # partition the visible facets' points\n",
                  "            self.regions.append((f.id, [v.id for v in visible], sorted({g.id for v in visible for g in v.nb if not g.visible}), [nf.id for nf in newf]))\n"
                  "            # partition the visible facets' points\n")
src = src.replace("        self.events = []\n", "        self.events = []\n        self.regions = []\n", 1)
ns = {}
exec(compile(src, "qhull_rows_instrumented", "exec"), ns)


def depth(points):
    d = ns["QhullDelaunay2D"](points)
    gen = {}                       # facet id -> generation (initial facets: 0)
    rounds_of_gen = {}
    last_round = {}                # facet id -> round (within ITS generation's processing) of the last insertion that touched it
    cur_gen = -1
    order_ok = True
    prev_gen = 0
    per_gen_entries = {}
    for fid, vis, hz, new in d.regions:
        g = gen.get(fid, 0)
        if g < prev_gen:
            order_ok = False       # (would contradict the FIFO argument)
        prev_gen = g
        if g != cur_gen:
            cur_gen = g
            last_round = {}
        region = vis + hz
        r = 1 + max((last_round.get(x, 0) for x in region), default=0)
        for x in region:
            last_round[x] = r
        for x in new:
            gen[x] = g + 1
            last_round[x] = r      # (a later insertion of this generation that reaches the new cone depends on this one)
        rounds_of_gen[g] = max(rounds_of_gen.get(g, 0), r)
        per_gen_entries[g] = per_gen_entries.get(g, 0) + 1
    return len(d.regions), order_ok, rounds_of_gen, per_gen_entries


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for s in range(frames):
        p = synth.synth_frame(s, n, base_seed=999)[1]
        ins, ok, rounds, entries = depth(p)
        steps = sum(rounds.values())
        print("frame %d: %d insertions, %d generations (list order is generation order: %s), %d rounds in all = %.1f insertions per round; "
              "largest generation %d insertions in %d rounds" % (s, ins, len(rounds), ok, steps, ins / steps,
                                                               max(entries.values()), rounds[max(entries, key=entries.get)]))
        if s == 0:
            print("   generation: insertions / rounds  " + "  ".join("%d: %d/%d" % (g, entries[g], rounds[g]) for g in sorted(rounds)))


if __name__ == "__main__":
    main()
