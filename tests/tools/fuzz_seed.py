import sys; sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np, importlib.util
from oracle import ptref as R
from pathtracer_amd import binding
spec=importlib.util.spec_from_file_location("tf","tests/test_gpu_fuzz.py"); m=importlib.util.module_from_spec(spec); spec.loader.exec_module(m)
seed=int(sys.argv[1])
sc,depth,thin=m.random_scene(seed)
osc=R.OracleScene(sc)
opt=R.options(R.RNG_COUNTER,R.MATH_DET,max_depth=depth,camera_model=thin)
ids=osc.primary_ids(opt,1)
for accel in (1,2):
    with binding.PtGpu(accel=accel,max_depth=depth,camera_model=thin) as gpu:
        gpu.upload_scene(sc); got=gpu.primary_ids(1)
    bad=np.argwhere(got!=ids)
    print('accel',accel,'bad',len(bad))
    for y,x in bad[:10]:
        print((y,x),'gpu',got[y,x],'oracle',ids[y,x])
print('depth',depth,'thin',thin,'jitter',sc.image_settings.sub_pixel_sampling)
print('n_sph',len(sc.spheres),'n_pl',len(sc.planes),'n_es',len(sc.emissive_spheres),'n_ep',len(sc.emissive_planes))
allp=list(sc.planes)+list(sc.emissive_planes)
for i,p in enumerate(sc.spheres+sc.emissive_spheres): print('S',i,p)
for i,p in enumerate(allp): print('P',len(sc.spheres)+len(sc.emissive_spheres)+i,p.origin,p.u,p.v,p.material.kind)
