"""Producer-fused updates at 64 x 1e6 (bench.py's fused_updates_n1e6) as a table: python tools/bench_fused.py [f64|f32]"""
import sys
sys.path.insert(0, '.')
import tramp_amd as T
import bench
ctx = T.default_context()
for dt in (sys.argv[1:] or ["f64", "f32"]):
    res = bench.fused_updates_n1e6(T, ctx, dt)
    for k, r in res.items():
        if "error" in r:
            print(f"{k:9s} {dt}: {r['error']}")
            continue
        print(f"{k:9s} {dt}: update {r['us_per_update']:7.1f} us = consumer pass {r['us_consumer_pass']:7.1f} + producer "
              f"{r['us_added_to_producer']:+7.1f} (mad {r['us_added_to_producer_mad']:.1f}); {r['frac_of_hbm_peak']:.3f} of HBM peak; "
              f"standalone {r['us_standalone']:7.1f} us = {r['standalone_frac_of_hbm_peak']:.3f}", flush=True)
