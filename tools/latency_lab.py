import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np
from aprilslam_amd import _lib, synth
w,h=1280,720
rng=np.random.default_rng(20250621)
frame,_=synth.render_frame(w,h,synth.random_scene(w,h,20,rng),18.0)
K=synth.camera_matrix(w,h)
det=_lib.Detector(id_limit=0)
for B in (1,2,4,8):
    fr=np.stack([frame]*B)
    for prof in (False, True):
        det.set_profiling(prof)
        for i in range(20): det.detect_host(fr, K=K, dist=np.zeros(4), tag_size=10.0)
        t0=time.perf_counter()
        for i in range(100): det.detect_host(fr, K=K, dist=np.zeros(4), tag_size=10.0)
        dt=(time.perf_counter()-t0)/100*1e3
        if prof:
            st=det.stage_times(); ks=sum(v for k,v in st.items() if k.startswith('k_'))
            print("B=%d profiled: %.3f ms/call, kernels sum %.3f, host %s" % (B, dt, ks, {k:round(v,3) for k,v in st.items() if k.startswith('host')}))
            if B==1: print({k:round(v,4) for k,v in st.items() if k.startswith('k_')})
        else:
            print("B=%d: %.3f ms/call" % (B, dt))
