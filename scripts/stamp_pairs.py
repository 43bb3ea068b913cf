import struct, numpy as np, collections, sys
raw=open(sys.argv[1],'rb').read()
grid,bm,M,chunk=struct.unpack("4i",raw[:16])
a=np.frombuffer(raw[16:],dtype=np.uint64).reshape(grid,6).astype(np.int64)
t0=a[:,0].min()
st,l0,l1,en=[(a[:,i]-t0)*0.01 for i in range(4)]
hw,xcc=a[:,4],a[:,5]&0xf
slot=hw&0xf
print("span %.1f"%en.max(), "slot0 loop-end med %.1f, slot1 loop-end med %.1f"%(np.median(l1[slot==0]), np.median(l1[slot==1])))
