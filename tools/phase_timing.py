import sys, time, os
sys.path.insert(0, '/root/repo/practical-path-guiding_amd')
import torch, ppg_host
props = dict(budgetType="spp", sppPerPass=4, maxDepth=10, rrDepth=10, strictNormals=1, seed=1234, budget=4.0*255)
scene = ppg_host.cbox_scene(1280, 720)
WORLD = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for rep in range(2):
    e = ppg_host.Engine.hip(**props); e.set_scene(scene)
    if WORLD > 1: e.set_shard(0, WORLD, 32)
    t = {}
    def T(name, f, *a):
        torch.cuda.synchronize(); t0=time.perf_counter(); r=f(*a); torch.cuda.synchronize(); t[name]=t.get(name,0)+time.perf_counter()-t0; return r
    t0=time.perf_counter()
    T('begin_render', e.begin_render)
    passes=[1,2,4,8,16,32,64,128]
    for it,p in enumerate(passes):
        T('begin_iteration', e.begin_iteration, it==len(passes)-1)
        T('passes', e.render_passes_nostat, p)
        T('finish', e.finish_passes)
        T('build', e.build_sdtree)
        T('end_it', e.end_iteration)
    T('end_render', e.end_render)
    tot=time.perf_counter()-t0
    print(rep, 'total %.1f ms'%(tot*1e3), {k:round(v*1e3,2) for k,v in t.items()}, 'Msamples/s', 1280*720*4*255/tot/1e6/WORLD)
