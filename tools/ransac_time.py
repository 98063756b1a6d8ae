#!/usr/bin/env python3
"""akz_remove_outliers (host trials) by match count on the cores this process may use: python tools/ransac_time.py
(taskset -c 0-3 python tools/ransac_time.py for fewer cores)"""
import sys, time, os, numpy as np
sys.path.insert(0,'akaze-rust_amd/python')
import akaze_amd as A
rng=np.random.default_rng(1)
def run(nm, trials, reps=20):
    k0=np.zeros(nm, dtype=A.KEYPOINT_DTYPE); k1=np.zeros(nm, dtype=A.KEYPOINT_DTYPE)
    k0['x']=rng.uniform(0,3840,nm); k0['y']=rng.uniform(0,2160,nm)
    k1['x']=k0['x']+17+rng.normal(0,0.5,nm); k1['y']=k0['y']+9+rng.normal(0,0.5,nm)
    m=np.zeros(nm, dtype=A.MATCH_DTYPE); m['index_0']=np.arange(nm); m['index_1']=np.arange(nm)
    A.remove_outliers(k0,k1,m,trials,0.05,3.0)
    t=time.perf_counter()
    for _ in range(reps): out=A.remove_outliers(k0,k1,m,trials,0.05,3.0)
    return round((time.perf_counter()-t)/reps*1e3,3)
print("cpus", len(os.sched_getaffinity(0)))
for nm in (8, 1000, 8264):
    print(nm, "matches, 1000 trials:", run(nm,1000), "ms")
