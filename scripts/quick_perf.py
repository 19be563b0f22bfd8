import sys, time
sys.path.insert(0,'pbrt-v3-rs_amd'); sys.path.insert(0,'tests')
import numpy as np, pbrt_hip
b=pbrt_hip.default_binding(); h=pbrt_hip.Host(b)
for n,res,spp in [(100000,512,64),(1000000,512,16)]:
    t0=time.time()
    s=pbrt_hip.Scene()
    spec=pbrt_hip.SceneSpec(n_tris=n,xres=res,yres=res,spp=spp)
    pbrt_hip.capture_spec(spec,s,h)
    t1=time.time()
    xyz,wt,st=s.render_path()
    t2=time.time()
    xyz,wt,st=s.render_path()
    t3=time.time()
    rgb=s.film_to_rgb(xyz,wt)
    d=st.as_dict()
    rays=d['regular_rays']+d['shadow_rays']
    print(n,res,spp,'setup %.2fs first %.2fs second %.2fs'%(t1-t0,t2-t1,t3-t2),'mean',rgb.mean(),'min',rgb.min(),'max',rgb.max())
    print(d, 'Mrays/s %.1f'%(rays/d['render_seconds']/1e6))
