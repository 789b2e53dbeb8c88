import sys, os
sys.path.insert(0, os.getcwd())
from tools.closed_loop import device_closed_loop
from trajtrack_mpcndqn_rlboost_amd import MpcConfig
for warm in (False, True):
    for order, tp in (("as_given", 0), ("as_given", None), ("longest_first", None), ("longest_first", 512)):
        r = device_closed_loop(MpcConfig(), 8192, 30, 5, 4, warm, order, tail_promotion=tp)
        print(f"{'warm' if warm else 'cold'} {order:14s} tail_promotion={tp}: {r['ms_per_tick']:.1f} ms/tick, worst {r['ms_per_tick_min_max'][1]:.1f}, {r['value']:.0f} solves/s", flush=True)
