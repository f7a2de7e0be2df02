#!/usr/bin/env python3
"""Byte-level differential fuzz of the encode direction against the reference binary: damaged JPEGs (bit flips in header and scan,
insertions, truncation, stray 0xFF, Huffman tables with other code counts / symbols) -- either both sides refuse the file with the same exit code or both write the same .lep.
The reference ends a failing run with syscall(SYS_exit) on a worker thread: the process status stays 0 and the code's NAME is the
last line on stderr, which is what is compared.  python tests/fuzz/diff_jpeg_bytes.py <seed> <trials>; 800 mutants: no difference."""
import os, sys, random, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden, golden_cases
import oracle_binding as ob
from lepton_amd.codec import JpegImage, LeptonError
REF='/root/repo/oracle/_ref/lepton'
seed=int(sys.argv[1]); N=int(sys.argv[2])
rnd=random.Random(seed)
names=[n for n in golden_cases() if len(golden(n)[0])<30000]
src=lambda n: golden(n)[0]
if os.environ.get('JPEG_FUZZ_DIR'):    # any other JPEGs (e.g. larger ones that take 4 or 8 thread segments)
    D=os.environ['JPEG_FUZZ_DIR']; names=sorted(n[:-4] for n in os.listdir(D) if n.endswith('.jpg'))
    src=lambda n: open(os.path.join(D,n+'.jpg'),'rb').read()
jp,lp='/tmp/j%d.jpg'%seed,'/tmp/j%d.lep'%seed
same=refused=bad=0
for trial in range(N):
    name=rnd.choice(names)
    b=bytearray(src(name))
    kind=os.environ.get('JPEG_FUZZ_KIND') or rnd.choice(["flip_scan","flip_hdr","trunc","flip_any","insert","ff","dht"])
    if kind=="flip_scan":
        for _ in range(rnd.randint(1,3)): b[rnd.randrange(len(b)//2,len(b))]^=1<<rnd.randrange(8)
    elif kind=="dht":                  # a Huffman table's code counts or symbols (tables that no longer follow T.81 Annex C)
        at=[i for i in range(len(b)-20) if b[i]==0xff and b[i+1]==0xc4]
        i=rnd.choice(at) if at else 0; ln=(b[i+2]<<8)|b[i+3]
        for _ in range(rnd.randint(1,3)):
            j=i+5+rnd.randrange(0,16) if rnd.random()<0.6 else i+4+rnd.randrange(0,max(ln-2,1))
            b[j]=rnd.choice([b[j]^(1<<rnd.randrange(8)),b[j]+1&255,b[j]-1&255,rnd.randrange(256)])
    elif kind=="flip_hdr": b[rnd.randrange(0,min(len(b),700))]^=1<<rnd.randrange(8)
    elif kind=="trunc": b=b[:rnd.randrange(100,len(b))]
    elif kind=="flip_any": b[rnd.randrange(len(b))]=rnd.randrange(256)
    elif kind=="insert":
        i=rnd.randrange(len(b)); b[i:i]=bytes(rnd.randrange(256) for _ in range(rnd.randint(1,4)))
    else:
        i=rnd.randrange(len(b)//2,len(b)); b[i]=0xff
    b=bytes(b); open(jp,'wb').write(b)
    if os.path.exists(lp): os.unlink(lp)
    try:
        r=subprocess.run([REF,'-unjailed','-skipverify',jp,lp],capture_output=True,timeout=60); rc=r.returncode
        codes={'ASSERTION_FAILURE':1,'CODING_ERROR':2,'SHORT_READ':3,'UNSUPPORTED_4_COLORS':4,'THREAD_PROTOCOL_ERROR':5,'COEFFICIENT_OUT_OF_RANGE':6,'STREAM_INCONSISTENT':7,'PROGRESSIVE_UNSUPPORTED':8,'FILE_NOT_FOUND':9,'SAMPLING_BEYOND_TWO_UNSUPPORTED':10,'SAMPLING_BEYOND_FOUR_UNSUPPORTED':11,'THREADING_PARTIAL_MCU':12,'VERSION_UNSUPPORTED':13,'ONLY_GARBAGE_NO_JPEG':14,'OS_ERROR':33,'HEADER_TOO_LARGE':34,'DIMENSIONS_TOO_LARGE':35,'MALLOCED_NULL':36,'OOM':37,'TOO_MUCH_MEMORY_NEEDED':38,'EARLY_EXIT':40,'ROUNDTRIP_FAILURE':41,'UNSUPPORTED_JPEG':42,'UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0':43}
        named=[l.strip() for l in r.stderr.decode('latin1').split('\n') if l.strip() in codes]
        if named: rc=codes[named[-1]]      # a failing run names its exit code on stderr whatever the process status says (and may leave a partial file)
        want=open(lp,'rb').read() if rc==0 and os.path.exists(lp) and os.path.getsize(lp)>0 else None
    except subprocess.TimeoutExpired: want=None; rc='timeout'
    try:
        img=JpegImage(b); segs=img.plan(); streams,_=ob.oracle_encode(img.desc,segs); got=img.write_lep(streams); code=0
    except LeptonError as e: got=None; code=e.code
    except RuntimeError as e:
        got=None; code=str(e)
        if 'exit code' in code: code=int(code.rsplit(' ',1)[1])
    if got!=want or (got is None and code!=rc and isinstance(rc,int) and rc>=0):
        bad+=1; print('DIFF',trial,kind,name,'ref',rc,None if want is None else len(want),'ours',code,None if got is None else len(got)); open('/tmp/jdiff_%d_%d.jpg'%(seed,trial),'wb').write(b)
    elif got is None: refused+=1
    else: same+=1
print('same',same,'refused',refused,'bad',bad)
