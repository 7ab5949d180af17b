"""Restatement (on top of the C oracle's primitives) of the orchestration in
ACS_Rank::searchBestPathOfPoints (ACSRank_3D.hpp:427-504), ACS_GTSP::readFromGraphFile
(ACS_GTSP.hpp:224-253) and read_all_segments (:286-298).  Test infrastructure only."""
import numpy as np

import oracle_lib as O


def read_points_file(path):
    """ACSRank_3D.hpp:444-455: `%d` then P x `%f %f %f`."""
    tok = open(path).read().split()
    n = int(tok[0])
    return np.array([float(t) for t in tok[1:1 + 3 * n]], np.float32).reshape(n, 3)


def graph_file_bytes(n_points, costs):
    """ACSRank_3D.hpp:469-503 incl. the header overwrite: "0 0\\n", one "%.3f\\n" per
    pair, then rewind and "%d %d\\r" on top of whatever is there."""
    body = b"0 0\n" + b"".join(("%.3f\n" % c).encode() for c in costs)
    head = ("%d %d\r" % (n_points, len(costs))).encode()
    return head + body[len(head):]


def parse_graph_text(text):
    """What fscanf("%d %d") + N(N-1)/2 x fscanf("%lf") sees (ACS_GTSP.hpp:229,243)."""
    tok = text.split()
    n, cnt = int(tok[0]), int(tok[1])
    vals = [float(t) for t in tok[2:2 + n * (n - 1) // 2]]
    d = np.zeros((n, n), np.float64)
    k = 0
    for i in range(n):
        for j in range(i + 1, n):
            d[i, j] = d[j, i] = vals[k]
            k += 1
    return d, cnt


def search_best_path_of_points(grid, pts, predict, seed, iters=150):
    rng = O.srand(seed)  # initFromGridMap :327
    acs = O.Acs(grid)
    P = len(pts)
    cost = np.zeros((P, P), np.float32)
    paths = {}
    costs_in_order = []
    prev_path = np.zeros(0, np.int32)
    for i in range(P):
        for j in range(i + 1, P):
            sid, eid = grid.resolve(pts[i]), grid.resolve(pts[j])
            if sid < 0 or eid < 0:
                return None
            acs.solve(sid, eid, iters, predict, mode=O.REF, rng=rng)
            acs.reset()
            L = acs.best_L
            ids, _ = acs.best_path()  # persists when no ant arrived (Q9)
            prev_path = ids
            cost[i, j] = cost[j, i] = L
            paths[(i, j)] = paths[(j, i)] = ids
            costs_in_order.append(float(L))
    return dict(cost=cost, paths=paths, graph=graph_file_bytes(P, costs_in_order), rng=rng, acs=acs)


def read_all_segments(grid, tour_edges, paths):
    xs, ys, zs = [], [], []
    nx, ny = grid.nx, grid.ny
    for a, b in tour_edges[:-1]:
        ids = paths[(int(a), int(b))].astype(np.int64)
        xs.append(grid.cx[ids % nx])
        ys.append(grid.cy[(ids // nx) % ny])
        zs.append(grid.cz[ids // (nx * ny)])
    return np.concatenate(xs), np.concatenate(ys), np.concatenate(zs)
