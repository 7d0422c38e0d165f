import sys, os, tempfile
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
import numpy as np
import synth_tflite as st
import rs_face_detection_tflite_amd as mi
from oracle import pyoracle
name = sys.argv[1]
mk,h,w = st.CASES[name]
p = os.path.join(tempfile.mkdtemp(), name+'.tflite'); open(p,'wb').write(mk())
x = np.random.RandomState(105).uniform(-1,1,(5,h,w,3)).astype(np.float32)
refs = pyoracle.Model(p).run(x, nthreads=5)
for opts in ({}, {"heads":0},):
    m = mi.Model(p); m.set_option("fuse", 5)
    print(m.describe().splitlines()[0])
    outs = m.run(x)
    for o,r in zip(outs, refs):
        r = r.reshape(o.shape); d = np.abs(o-r)
        print(o.shape, "max err", d.max(), "first bad row per frame:", [int(np.argmax(d[f].max(axis=-1) > 1e-3)) if (d[f].max()>1e-3) else -1 for f in range(o.shape[0])],
              "bad rows:", [int((d[f].max(axis=-1) > 1e-3).sum()) for f in range(o.shape[0])])
    m.close()
    break
