import os
import sys, ctypes as C, numpy as np
sys.path.insert(0,'.')
import bliss_amd
from tests.oracle_py import Oracle
orc=Oracle()
n=44100*2*30
corpus=bliss_amd.DeviceCorpus([n],2,30)
corpus.synth(1000,44100)
corpus.analyze()
got=corpus.fetch()
lib=bliss_amd.load()
nbf=int(got[0]['nb_frames'])
en=np.zeros(nbf,dtype=np.float32)
print('copied',lib.bl_amd_last_energies(en.ctypes.data_as(C.POINTER(C.c_float)), nbf))
pcm=corpus.pcm.cpu().numpy()[:n]
ref,ren=orc.envelope(pcm,30)
ren=ren[:nbf]
nw=nbf-2
bad=np.nonzero(en[:nw]!=ren[:nw])[0]
print('windows',nw,'mismatch',bad.size, bad[:40])
if bad.size:
    rel=np.abs(en[bad]-ren[bad])/ren[bad]
    print('rel err max',rel.max(),'median',np.median(rel))
    print('bad mod 28:',np.bincount(bad%28,minlength=28))
    print('bad mod 4:',np.bincount(bad%4,minlength=4))
