#!/usr/bin/env python3
"""Byte-level differential fuzz of the decode direction against the reference binary (bit flips, overwrites, insertions,
truncation of reference-written .lep files): either both sides refuse a file or both restore the same bytes.  A tool, run by hand
where oracle/_ref/lepton exists:  python tests/fuzz/diff_lep_bytes.py <seed> <trials>.  Found with it: the two-row ring of the
reference's baseline decoder behind a truncation point (tests/test_fuzz_host.py).  Known to differ: a damaged stream whose garbage
tail lies behind the point where the byte bound cuts the output -- the reference decodes rows lazily and never gets there (one
segment) or races its worker thread's exit(7) (several), here whole segments are decoded first and the file is refused."""
import os, sys, random, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden, golden_cases
import oracle_binding as ob
from lepton_amd.codec import LepFile, LeptonError
REF='/root/repo/oracle/_ref/lepton'
seed=int(sys.argv[1]); N=int(sys.argv[2])
rnd=random.Random(seed)
names=[n for n in golden_cases() if len(golden(n)[1])<40000]
src=lambda n: golden(n)[1]
if os.environ.get('LEP_FUZZ_DIR'):     # any other reference-written .lep files (e.g. larger ones with 4 or 8 thread segments)
    D=os.environ['LEP_FUZZ_DIR']; names=sorted(n[:-4] for n in os.listdir(D) if n.endswith('.lep'))
    src=lambda n: open(os.path.join(D,n+'.lep'),'rb').read()
lp,jp='/tmp/b%d.lep'%seed,'/tmp/b%d.jpg'%seed
same=refused=bad=0
for trial in range(N):
    name=rnd.choice(names)
    b=bytearray(src(name))
    kind=rnd.choice(["flip_stream","flip_hdr","trunc","flip_any","insert","insert_stream"])
    if kind=="flip_stream":
        for _ in range(rnd.randint(1,3)): b[rnd.randrange(len(b)//2,len(b))]^=1<<rnd.randrange(8)
    elif kind=="flip_hdr": b[rnd.randrange(0,min(len(b),40))]^=1<<rnd.randrange(8)
    elif kind=="trunc": b=b[:rnd.randrange(10,len(b))]
    elif kind=="flip_any": b[rnd.randrange(len(b))]=rnd.randrange(256)
    elif kind=="insert":
        i=rnd.randrange(len(b)); b[i:i]=bytes(rnd.randrange(256) for _ in range(rnd.randint(1,4)))
    else:
        i=rnd.randrange(len(b)//2,len(b)); b[i:i]=bytes(rnd.randrange(256) for _ in range(rnd.randint(1,4)))
    b=bytes(b); open(lp,'wb').write(b)
    if os.path.exists(jp): os.unlink(jp)
    try:
        r=subprocess.run([REF,'-unjailed',lp,jp],capture_output=True,timeout=60); rc=r.returncode
        want=open(jp,'rb').read() if rc==0 and os.path.exists(jp) else None
    except subprocess.TimeoutExpired: want=None; rc='timeout'
    try:
        f=LepFile(b); ob.oracle_decode(f.desc,f.segments,f.streams); got=f.recode(); code=0
    except LeptonError as e: got=None; code=e.code
    except RuntimeError as e: got=None; code=str(e)
    if got!=want:
        bad+=1; print('DIFF',trial,kind,name,'ref',rc,None if want is None else len(want),'ours',code,None if got is None else len(got)); open('/tmp/bdiff_%d_%d.lep'%(seed,trial),'wb').write(b)
    elif got is None: refused+=1
    else: same+=1
print('same',same,'refused',refused,'bad',bad)
