#!/usr/bin/env python3
"""Structure-aware differential fuzz of the encode direction against the reference binary: the marker segments of a golden JPEG
are duplicated, deleted, swapped, resized, given other lengths, and new ones (DRI, COM, APPn, a second SOI / EOI, DNL, DHT / DQT
copies between scans) are inserted -- either both sides refuse the file with the same exit code or both write the same .lep.
python tests/fuzz/diff_jpeg_structured.py <seed> <trials>"""
import os, sys, random, subprocess
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from conftest import golden, golden_cases
import oracle_binding as ob
from lepton_amd.codec import JpegImage, LeptonError
REF='/root/repo/oracle/_ref/lepton'
CODES={'ASSERTION_FAILURE':1,'CODING_ERROR':2,'SHORT_READ':3,'UNSUPPORTED_4_COLORS':4,'THREAD_PROTOCOL_ERROR':5,'COEFFICIENT_OUT_OF_RANGE':6,'STREAM_INCONSISTENT':7,'PROGRESSIVE_UNSUPPORTED':8,'FILE_NOT_FOUND':9,'SAMPLING_BEYOND_TWO_UNSUPPORTED':10,'SAMPLING_BEYOND_FOUR_UNSUPPORTED':11,'THREADING_PARTIAL_MCU':12,'VERSION_UNSUPPORTED':13,'ONLY_GARBAGE_NO_JPEG':14,'OS_ERROR':33,'HEADER_TOO_LARGE':34,'DIMENSIONS_TOO_LARGE':35,'MALLOCED_NULL':36,'OOM':37,'TOO_MUCH_MEMORY_NEEDED':38,'EARLY_EXIT':40,'ROUNDTRIP_FAILURE':41,'UNSUPPORTED_JPEG':42,'UNSUPPORTED_JPEG_WITH_ZERO_IDCT_0':43}

def split(jpg):
    """[(marker, payload-with-length-bytes or b'' , entropy bytes after it)]; SOI first"""
    out=[]; i=2; n=len(jpg)
    assert jpg[:2]==b'\xff\xd8'
    out.append([0xd8,b'',b''])
    while i+4<=n and jpg[i]==0xff:
        m=jpg[i+1]
        if m==0xd9: out.append([m,b'',jpg[i+2:]]); return out
        ln=(jpg[i+2]<<8)|jpg[i+3]
        seg=jpg[i+2:i+2+ln]; i+=2+ln
        ent=b''
        if m==0xda:
            j=i
            while j+1<n and not (jpg[j]==0xff and jpg[j+1]!=0 and not 0xd0<=jpg[j+1]<=0xd7): j+=1
            if j+1>=n: j=n
            ent=jpg[i:j]; i=j
        out.append([m,seg,ent])
    out.append([None,b'',jpg[i:]])
    return out
def join(segs):
    b=bytearray()
    for m,seg,ent in segs:
        if m is not None: b+=bytes([0xff,m])
        b+=seg+ent
    return bytes(b)

seed=int(sys.argv[1]); N=int(sys.argv[2])
rnd=random.Random(seed)
names=[n for n in golden_cases() if len(golden(n)[0])<30000]
jp,lp='/tmp/js%d.jpg'%seed,'/tmp/js%d.lep'%seed
same=refused=bad=0
for trial in range(N):
    name=rnd.choice(names)
    segs=split(golden(name)[0])
    for _ in range(rnd.randint(1,2)):
        kind=rnd.choice(["dup","del","swap","len","insert","grow","shrink","body"])
        k=rnd.randrange(1,len(segs))
        if kind=="dup": segs.insert(rnd.randrange(1,len(segs)),[segs[k][0],segs[k][1],b'' if rnd.random()<0.7 else segs[k][2]])
        elif kind=="del" and len(segs)>3: del segs[k]
        elif kind=="swap":
            j=rnd.randrange(1,len(segs)); segs[k],segs[j]=segs[j],segs[k]
        elif kind=="len" and len(segs[k][1])>=2:
            s=bytearray(segs[k][1]); v=((s[0]<<8)|s[1])+rnd.choice([-3,-2,-1,1,2,3,17,256]); v=max(0,min(65535,v)); s[0]=v>>8; s[1]=v&255; segs[k][1]=bytes(s)
        elif kind=="insert":
            m=rnd.choice([0xdd,0xfe,0xe0,0xe1,0xee,0xd8,0xd9,0xdc,0xc4,0xdb,0xc0,0xc2,0xda,0x01,0xd0,0xf0,0xc8])
            if m in (0xd8,0xd9,0x01,0xd0): new=[m,b'',b'']
            elif m==0xdd: new=[m,bytes([0,4,0,rnd.choice([0,1,2,5,200])]),b'']
            elif m in (0xc4,0xdb,0xc0,0xc2,0xda):
                src=[s for s in segs if s[0]==m]
                new=[m,src[0][1],b''] if src else [m,bytes([0,2]),b'']
            else:
                body=bytes(rnd.randrange(256) for _ in range(rnd.choice([0,1,5,60,700])))
                new=[m,bytes([(len(body)+2)>>8,(len(body)+2)&255])+body,b'']
            segs.insert(rnd.randrange(1,len(segs)+1),new)
        elif kind=="grow" and len(segs[k][1])>=2:
            extra=bytes(rnd.randrange(256) for _ in range(rnd.randint(1,9))); s=bytearray(segs[k][1]+extra); v=len(s); s[0]=v>>8; s[1]=v&255; segs[k][1]=bytes(s)
        elif kind=="shrink" and len(segs[k][1])>4:
            s=bytearray(segs[k][1][:-rnd.randint(1,min(6,len(segs[k][1])-3))]); v=len(s); s[0]=v>>8; s[1]=v&255; segs[k][1]=bytes(s)
        elif kind=="body" and len(segs[k][1])>2:
            s=bytearray(segs[k][1]); s[rnd.randrange(2,len(s))]=rnd.randrange(256); segs[k][1]=bytes(s)
    b=join(segs); open(jp,'wb').write(b)
    if os.path.exists(lp): os.unlink(lp)
    try:
        r=subprocess.run([REF,'-unjailed','-skipverify',jp,lp],capture_output=True,timeout=60); rc=r.returncode
        named=[l.strip() for l in r.stderr.decode('latin1').split('\n') if l.strip() in CODES]
        if named: rc=CODES[named[-1]]
        want=open(lp,'rb').read() if rc==0 and os.path.exists(lp) and os.path.getsize(lp)>0 else None
    except subprocess.TimeoutExpired: want=None; rc='timeout'
    try:
        img=JpegImage(b); sg=img.plan(); streams,_=ob.oracle_encode(img.desc,sg); got=img.write_lep(streams); code=0
    except LeptonError as e: got=None; code=e.code
    except RuntimeError as e:
        got=None; code=str(e)
        if 'exit code' in code: code=int(code.rsplit(' ',1)[1])
    if got!=want or (got is None and code!=rc and isinstance(rc,int) and rc>=0):
        bad+=1; print('DIFF',trial,name,'ref',rc,None if want is None else len(want),'ours',code,None if got is None else len(got),flush=True); open('/tmp/jsdiff_%d_%d.jpg'%(seed,trial),'wb').write(b)
    elif got is None: refused+=1
    else: same+=1
print('same',same,'refused',refused,'bad',bad)
