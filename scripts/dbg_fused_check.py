import numpy as np, sys
sys.path.insert(0, '.')
from oracle import resample as R
H,W=1080,1920
frame=np.fromfile('gpurun_out/dbg_frame.bin',dtype=np.uint8).reshape(H,W,3)
d=np.fromfile('gpurun_out/dbg_fused.bin',dtype=np.uint8)
m=d[:96].view(np.int32)
dy0,dy1,ry0,ry1,ty0,ty1,p0,p1,off1,off2,total,rb,sw,sh,rw,rh,px,py,nh,nv,dd,mode,sx0,sy0=[int(x) for x in m]
lds=d[256:256+total]
n0=ty1-ty0; n2=ry1-ry0
sl=frame[sy0:sy0+sh, sx0:sx0+sw]
T1=R._resample_axis1(sl, rw)
T2=R._resample_axis1(T1.transpose(1,0,2), rh).transpose(1,0,2)
B1=np.stack([lds[off1+y*p1:off1+y*p1+rw*3] for y in range(n0)])
B2=np.stack([lds[off2+y*p1:off2+y*p1+rw*3] for y in range(n2)])
print('B1 ok', np.array_equal(B1.reshape(n0,rw,3), T1[ty0:ty1]), 'B2 ok', np.array_equal(B2.reshape(n2,rw,3), T2[ry0:ry1]))
bad=np.argwhere(B2.reshape(n2,rw*3)!=T2[ry0:ry1].reshape(n2,rw*3)); print(len(bad), np.bincount(bad[:,1]%4, minlength=4) if len(bad) else '')
