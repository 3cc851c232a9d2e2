"""A/B of the launch groups under library options taken at server creation (spiral_gpu_set_option): tools/stage_ab.py "fold_pair=0,fold_blocks=300" "fold_chain=0" ...
(an empty string = defaults).  Prints wall us per replay of run_pre / first_dim / run_post / run_query (hipGraph replays back to
back, 40 each after warm-up) at config 2 (or --nu1/--nu2).  SPIRAL_LIB=<path> selects another build."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spiral_amd as sa

args = [a for a in sys.argv[1:] if not a.startswith("--")]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
nu1, nu2 = int(opts.get("nu1", 8)), int(opts.get("nu2", 7))
reps = int(opts.get("reps", 40))
kw = {k: int(opts[k]) for k in ("t_gsw", "t_conv", "t_exp", "t_exp_right") if k in opts}
for cfg in args or [""]:
    env = {k.replace("SPIRAL_", "").lower(): int(v) for k, v in (kv.split("=") for kv in cfg.split(",") if kv)}
    old = {k: sa.get_option(k) for k in env}
    for k, v in env.items():
        sa.set_option(k, v)
    pg = sa.make_params(nu1, nu2, **kw)
    s = sa.get_shape(pg)
    srv = sa.Server(pg)
    srv.fill_db_random(3)
    rng = np.random.default_rng(1)
    mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
    srv.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    srv.set_query(mk((1, 2)))
    srv.use_graphs(True)
    if "overlap" in opts:
        srv.set_overlap(int(opts["overlap"]))
    out = {}
    for name, fn in (("pre", srv.run_pre), ("sweep", srv.first_dim), ("post", srv.run_post), ("query", srv.run_query)):
        for _ in range(5):
            fn()
        srv.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            srv.sync()
            best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        out[name] = round(best, 1)
    print(f"{os.environ.get('SPIRAL_LIB', 'product'):28s} {cfg or 'defaults':44s} {out}", flush=True)
    srv.close()
    for k, v in old.items():
        sa.set_option(k, v)
