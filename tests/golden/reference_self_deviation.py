"""Evidence behind the bars of tests/test_gpu_atsize.py: how far the UNMODIFIED reference moves away from ITSELF when a drop tile is
embedded in a bigger world (the batch slot of a contact depends on how many contacts precede it in tag order, nudge.cpp:4206-4339).
124 seeds x (a union of two 3600-box tiles vs the second tile alone), 1000 steps, exact-math build: 446,400 body samples.
Result committed as profiles/r02_reference_self_deviation.json (max relative position error 9.1e-5, 99.99 % quantile 4.8e-5,
max |dv| 2.0e-4).  Takes ~1 minute on 8 cores; needs oracle/_ref.   usage: python tests/golden/reference_self_deviation.py [out.json]"""
import sys, time, json; import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests'))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from nudge_amd import scenes as S
from oracle import refworld
import parity_util as P
def one(seed):
    scene=S.grid_tiles(2, side=60, seed=seed)
    u=refworld.RefWorld(scene, max_contacts=6*len(scene["body_transforms"])); u.step(1000); ub=u.bodies()
    sub,bm=S.extract_tile(scene,1)
    r=refworld.RefWorld(sub, max_contacts=6*len(sub["body_transforms"])); r.step(1000); rb=r.bodies()
    origin=scene["box_transforms"]["position"][1].astype(np.float64)
    err=P.rel_err(ub["transforms"]["position"][bm[1:]], rb["transforms"]["position"][1:], origin=origin)
    dv=np.abs(ub["momentum"]["velocity"][bm[1:]]-rb["momentum"]["velocity"][1:]).max(axis=1)
    return err, dv
t=time.time()
with ThreadPoolExecutor(8) as ex:
    res=list(ex.map(one, range(100,224)))
err=np.concatenate([r[0] for r in res]); dv=np.concatenate([r[1] for r in res])
out=dict(samples=int(len(err)), max_rel_pos=float(err.max()), q9999=float(np.quantile(err,0.9999)), q999=float(np.quantile(err,0.999)), frac_over_1e4=float((err>1e-4).mean()),
         max_dv=float(dv.max()), dv_q9999=float(np.quantile(dv,0.9999)), seconds=time.time()-t)
print(json.dumps(out))
open(sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT,'profiles','r02_reference_self_deviation.json'),'w').write(json.dumps(out))
