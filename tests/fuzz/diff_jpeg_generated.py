#!/usr/bin/env python3
"""Generator-driven differential sweep of the encode direction against the reference binary: coefficient-domain JPEGs
(tests/jpeg_writer.py) with random geometry, 1..4 components, sampling factors 1..4, restart intervals, 8- and 16-bit quantisation
tables, dense and sparse blocks, sequential multi-scan layouts -- both sides write the same .lep or refuse with the same code.
python tests/fuzz/diff_jpeg_generated.py <seed> <trials>"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import jpeg_writer as jw
import oracle_binding as ob
from lepton_amd.codec import JpegImage, LeptonError
REF='/root/repo/oracle/_ref/lepton'
CODES={'ASSERTION_FAILURE':1,'CODING_ERROR':2,'SHORT_READ':3,'UNSUPPORTED_4_COLORS':4,'THREAD_PROTOCOL_ERROR':5,'COEFFICIENT_OUT_OF_RANGE':6,'STREAM_INCONSISTENT':7,'PROGRESSIVE_UNSUPPORTED':8,'FILE_NOT_FOUND':9,'SAMPLING_BEYOND_TWO_UNSUPPORTED':10,'SAMPLING_BEYOND_FOUR_UNSUPPORTED':11,'THREADING_PARTIAL_MCU':12,'VERSION_UNSUPPORTED':13,'ONLY_GARBAGE_NO_JPEG':14,'OS_ERROR':33,'HEADER_TOO_LARGE':34,'DIMENSIONS_TOO_LARGE':35,'MALLOCED_NULL':36,'OOM':37,'TOO_MUCH_MEMORY_NEEDED':38,'EARLY_EXIT':40,'ROUNDTRIP_FAILURE':41,'UNSUPPORTED_JPEG':42,'UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0':43}
seed=int(sys.argv[1]); N=int(sys.argv[2])
jp,lp='/tmp/jg%d.jpg'%seed,'/tmp/jg%d.lep'%seed
same=refused=bad=skipped=0
for trial in range(N):
    rng=np.random.default_rng(seed*7919+trial)
    nc=int(rng.choice([1,3,3,3,4,2]))
    comps=[]
    for i in range(nc):
        h,v=(int(rng.choice([1,1,2,2,4,3])),int(rng.choice([1,1,2,2,4,3]))) if (i==0 or rng.random()<0.25) else (1,1)
        comps.append((int(rng.choice([i+1,i,200+i])) if rng.random()<0.2 else i+1,h,v,min(i,1) if rng.random()<0.8 else 0,min(i,1),min(i,1)))
    w,h=int(rng.integers(1,400)),int(rng.integers(1,300))
    kw=dict(quality=int(rng.choice([1,10,50,85,95,100])),density=float(rng.choice([0.02,0.25,0.7,3.0])),amp=float(rng.choice([1,20,150,600])),
            restart_interval=int(rng.choice([0,0,1,2,5,33])))
    try:
        if nc>1 and rng.random()<0.2:
            order=list(range(nc)); k=int(rng.integers(1,nc)); scans=[order[:k],order[k:]]
            jpg=jw.write_sequential_scans(w,h,comps,rng,scans,**kw)[0]
        else:
            if rng.random()<0.15: kw['dqt16']=True
            jpg=jw.write_baseline(w,h,comps,rng,**kw)[0]
    except Exception as e:
        skipped+=1; continue
    open(jp,'wb').write(jpg)
    if os.path.exists(lp): os.unlink(lp)
    try:
        r=subprocess.run([REF,'-unjailed','-skipverify',jp,lp],capture_output=True,timeout=120); rc=r.returncode
        named=[l.strip() for l in r.stderr.decode('latin1').split('\n') if l.strip() in CODES]
        if named: rc=CODES[named[-1]]
        want=open(lp,'rb').read() if rc==0 and os.path.exists(lp) and os.path.getsize(lp)>0 else None
    except subprocess.TimeoutExpired: want=None; rc='timeout'
    try:
        img=JpegImage(jpg); sg=img.plan(); streams,_=ob.oracle_encode(img.desc,sg); got=img.write_lep(streams); code=0
    except LeptonError as e: got=None; code=e.code
    except RuntimeError as e:
        got=None; code=str(e)
        if 'exit code' in code: code=int(code.rsplit(' ',1)[1])
    if got!=want or (got is None and code!=rc and isinstance(rc,int) and rc>=0):
        bad+=1; print('DIFF',trial,w,h,comps,kw,'ref',rc,None if want is None else len(want),'ours',code,None if got is None else len(got),flush=True); open('/tmp/jgdiff_%d_%d.jpg'%(seed,trial),'wb').write(jpg)
    elif got is None: refused+=1
    else: same+=1
print('same',same,'refused',refused,'bad',bad,'skipped',skipped)
