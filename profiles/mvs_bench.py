import sys, time, json, os
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np
from stereoreconstruction_amd import capi, synthetic
W,H,D,NV = 1280,960,128,8
if len(sys.argv)>1 and sys.argv[1]=='small': W,H,D,NV=320,240,64,4
cams3 = synthetic.semicircle_rig(NV, W, H, radius=10.0, step_deg=22.5, focal=float(W))
rgba, masks, depth = synthetic.render_sphere_views(cams3, W, H, 0x5EED0004, sphere_radius=2.0, tex_size=1024)
cams=[capi.camera_from_krt(K,R,t) for (K,R,t) in cams3]
zmin,zmax=8.0,12.0
p=capi.params_mvs(min_depth=zmin,max_depth=zmax,num_depth_levels=D,cross_check_threshold=2*(zmax-zmin)/(D-1))
neigh=capi.mvs_neighbours(cams,p)
ctx=capi.Context(0)
for v in range(NV): ctx.upload_view(v, rgba[v], masks[v], cams[v])
def run():
    t0=time.perf_counter()
    ne=0
    for v in range(NV):
        ctx.mvs_initial_estimate(v, neigh[v], p)
    ctx.synchronize(); t1=time.perf_counter()
    for v in range(NV): ctx.mvs_cross_check(list(range(NV)), v, p)
    ctx.synchronize(); t2=time.perf_counter()
    return t1-t0, t2-t1
run()
ctx.profile_enable(True)
a,b=run()
st=ctx.stats()
links=sum(len(n) for n in neigh)
print('initial %.1f ms cross %.2f ms  nominal Mhyp/s %.1f  n_eval(last view) %d  masked px %d'%(a*1e3,b*1e3, W*H*D*links/a/1e6, st['n_eval'], st['n_pixels']))
print(ctx.profile())
d0=ctx.download_depth(0); print('finite frac in mask', np.isfinite(d0[masks[0]==1]).mean(), 'neg', (d0[masks[0]==1]==-1).mean())
