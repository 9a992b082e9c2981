#!/usr/bin/env python3
"""GPU box: where does the time of a sharded render go on ONE rank?  The KITCHEN workload of bench.py rendered through the torch.distributed
(RCCL) reducer on a one-rank communicator — every exchange the identity — under cProfile, next to the plain render: the host-side cost of
the exchanges (staging copies, host synchronisations, the whole-key sort of the records a round hook hands back), which a sharded render
pays per rank whatever the number of ranks.

    python tools/dist_overhead_probe.py [passes]"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "practical-path-guiding_amd")); sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist
import ppg_host
from ppg_host.distributed import TorchReducer
from bench import KITCHEN_FILE, scene_props

passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_ADDR", "127.0.0.1"), ("MASTER_PORT", "29544")):
    os.environ.setdefault(k, v)
torch.cuda.set_device(0)
dist.init_process_group("nccl")
scene = ppg_host.load_scene_file(KITCHEN_FILE)
props = scene_props(KITCHEN_FILE, dict(budgetType="spp", seed=1234))


def render(with_reducer, profile=False):
    e = ppg_host.Engine.hip(budget=float(passes), **props)
    e.set_scene(scene)
    g = ppg_host.GuidedPathTracer(engine=e, reducer=TorchReducer(dist, torch.device("cuda", 0)) if with_reducer else None)
    torch.cuda.synchronize()
    pr = cProfile.Profile() if profile else None
    t0 = time.perf_counter()
    if pr:
        pr.enable()
    g.render()
    torch.cuda.synchronize()
    if pr:
        pr.disable()
    dt = time.perf_counter() - t0
    e.close()
    return dt, pr


render(True); render(False)
plain = min(render(False)[0] for _ in range(3))
red = min(render(True)[0] for _ in range(3))
print("plain %.1f ms, with the reducer on one rank %.1f ms (+%.1f ms)" % (plain * 1e3, red * 1e3, (red - plain) * 1e3))
dt, pr = render(True, profile=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print("\n".join(l for l in s.getvalue().splitlines() if l.strip())[:6000])
dist.destroy_process_group()
