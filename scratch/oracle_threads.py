import sys, time, os
sys.path.insert(0,'oracle')
import numpy as np, oracle as O
rng=np.random.default_rng(0)
n=100000
sc=rng.integers(0,1<<62,size=(n,4),dtype=np.uint64); sc[:,3]&=np.uint64((1<<60)-1)
G=O.ec_to_affine("g1",O.ec_generator("g1"))
pts=O.fixed_base_mul("g1",G,sc[::-1].copy())
t=time.time(); r=O.msm("g1",sc,pts); t1=time.time()-t
t=time.time(); r=O.msm("g1",sc,pts); t2=time.time()-t
x=np.concatenate([sc]*4)[:3*(1<<17)]
t=time.time(); y=O.fr_ntt(x, False, batch=3); t3=time.time()-t
print("threads", O.num_threads(), "msm %.3f %.3f s  ntt3x2^17 %.3f s" % (t1,t2,t3), flush=True)
