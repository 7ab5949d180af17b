import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    P, G, lazy, gens = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3] == "lazy", int(sys.argv[4])
    import numpy as np
    import bench
    from welding_robot_amd import api, synth
    ctx = api.Context(0)
    n, ants = 128, 256
    free, cx, cy, cz, prec, wall = synth.synth_grid(n, seed=2024, occ_prob=0.10)
    grid = api.Grid.from_occupancy(ctx, free, cx, cy, cz, prec, wall)
    ids = grid.resolve(np.array([[0, 0, 0], [n - 1, n - 1, n - 1]], np.float32))
    for (g, lz) in ((1, False), (0, False), (1, True), (0, True)):
        out, hist, steps, _ = bench.multi_start_run(ctx, grid, ids, n, ants, P, g, gens, lz, 5)
        print(round(out["problem_generations_per_s"]), flush=True)
    print(ctx.cache_stats())
    GiB = 1 << 30
    ctx.trim()
    import time; time.sleep(0.5)
    f, t = ctx.memory_info(); print("after trim free %.1f of %.1f GiB" % (f / GiB, t / GiB))
    sys.exit(0)
for args, env in [("2 0 seq 30", {}), ("8 0 seq 100", {}), ("8 0 seq 100", {"WA_DEV_POISON": "1"})]:
    r = subprocess.run([sys.executable, os.path.abspath(__file__)] + args.split(), capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
    print("==", args, env, "rc", r.returncode, (r.stdout[-700:] + r.stderr[-300:]).replace("\n", " | "), flush=True)
