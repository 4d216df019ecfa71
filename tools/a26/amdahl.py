"""an Amdahl bound for serving the K-pulse searches of the pvq_theta calls
the batched band stage cannot serve (bands with a reference vector: keyframe chroma, neighbour-predicted
luma).  The plain C reference encoder on two 1080p frames with an interposer (time_search.c: test
infrastructure) that times pvq_theta and pvq_search_rdo_double, split by whether the call has a reference.
No GPU.  profiles/r5_a26_amdahl.txt is its output."""
import ctypes, os, subprocess, sys, time, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/odhip_a26_time.so"
subprocess.run(["gcc", "-O2", "-fPIC", "-shared", "-o", so, os.path.join(HERE, "time_search.c"), "-ldl"], check=True)
cnt = ctypes.CDLL(so, mode=ctypes.RTLD_GLOBAL)
import encode_job as S
r = ctypes.CDLL(os.path.join(ROOT,'oracle','_ref','libdaalaref.so'))
cnt.cnt_set_ref(ctypes.c_void_p(r._handle))
w,h=1920,1080
frames=[S.frame_yuv(i,w,h) for i in range(2)]
t0=time.perf_counter()
pk=S.encode_frames(r,[0,1],frames,w,h)
dt=time.perf_counter()-t0
ts=(ctypes.c_double*2).in_dll(cnt,'t_search'); ns=(ctypes.c_long*2).in_dll(cnt,'n_search')
tt=(ctypes.c_double*2).in_dll(cnt,'t_theta'); nt=(ctypes.c_long*2).in_dll(cnt,'n_theta')
print('2 frames of 1080p, plain C encoder with timers: %.2f s (%.2f s/frame)' % (dt, dt/2))
for i,nm in enumerate(['keyframe luma bands with a null reference (what the batch serves)','bands with a reference vector / chroma (left to the reference)']):
    print('%-75s pvq_theta calls %8d  %.3f s/frame   of which pvq_search_rdo_double: calls %8d  %.3f s/frame' % (nm, nt[i]//2, tt[i]/2, ns[i]//2, ts[i]/2))
