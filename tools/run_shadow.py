import os, sys, ctypes as C
import numpy as np
ROOT="/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd()
sys.path.insert(0, ROOT)
import torch, ntracer_amd
from ntracer_amd import _lib, tracern
g = np.load(os.path.join(ROOT,"tests","golden","cell120_n4.npz"))
sc = tracern.CompositeScene.from_flat(4, g)
sc.add_light(tracern.PointLight(tracern.Vector(4, (8.0, 9.0, -7.0, 3.0)), (60.0, 60.0, 60.0)))
sc.add_light(tracern.GlobalLight(tracern.Vector(4, (0.2, -1.0, 0.3, 0.1)).unit(), (0.5, 0.5, 0.5)))
sc.set_shadows(os.environ.get("SHADOWS", "1") != "0")
W,H=960,540
fmt = ntracer_amd.ImageFormat(W,H,[ntracer_amd.Channel(8,1,0,0),ntracer_amd.Channel(8,0,1,0),ntracer_amd.Channel(8,0,0,1),ntracer_amd.Channel(8,0,0,0)])
fst=fmt._as_struct()
sel=[0,40,80,120]
o=np.ascontiguousarray(g["origins"][sel],np.float32); a=np.ascontiguousarray(g["axes"][sel],np.float32)
fb=torch.empty((4,fmt.pitch*H),dtype=torch.uint8,device="cuda"); st=torch.cuda.current_stream()
for rep in range(2):
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True); e0.record()
    _lib.check(_lib.lib().nt_render_frames_device(sc._handle,C.c_void_p(fb.data_ptr()),fmt.pitch*H,4,o.ctypes.data_as(_lib.f32p),a.ctypes.data_as(_lib.f32p),C.byref(fst),None,C.c_void_p(st.cuda_stream)))
    e1.record(); torch.cuda.synchronize()
    print("shadow scene: %.3f ms/frame"%(e0.elapsed_time(e1)/4))
print("checksum", int(fb.to(torch.int64).sum().item()))
