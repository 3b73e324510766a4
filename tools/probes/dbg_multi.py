import numpy as np, sys
sys.path.insert(0,'/root/repo')
import libsdr_amd as sa
from oracle import pyoracle as orc
ctx=sa.Context(0)
FSr=1e6; order=127; decim=8; Fc=100e3; C=1
rng=np.random.default_rng(1)
lut, inc = orc.freqshift_lut_i16(), orc.freqshift_inc(Fc, FSr)
taps = orc.iqbb_design(Fc, 12.5e3, FSr, order)
node = sa.IQBaseBandI16(ctx, taps, lut, inc, False, decim, channels=C, max_in=70000, epilogue=sa.EPI_FM)
x=rng.integers(-32768,32768,(C,4*16384,2),dtype=np.int16)
y,counts=node.process_multi(x,4)
bb=orc.IQBaseBandI16(taps,lut,inc,False,decim); fm=orc.FMDemodI16()
rs=[]; raw=[]
for j in range(4):
    r0=bb.process(x[0,j*16384:(j+1)*16384]); raw.append(r0); rs.append(fm.process(r0))
r=np.concatenate(rs)
print(counts)
for q in (2047,4095):
    print("q",q,"gpu",y[0][q-1:q+6],"ref",r[q-1:q+3], "raw re", np.concatenate(raw)[q-1:q+2,0])
# long call for comparison
node2 = sa.IQBaseBandI16(ctx, taps, lut, inc, False, decim, channels=C, max_in=70000, epilogue=sa.EPI_FM)
yl=node2.process(x)
print("long", yl[0][2046:2050])
