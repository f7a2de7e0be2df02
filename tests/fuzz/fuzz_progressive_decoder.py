#!/usr/bin/env python3
"""Mutation fuzz of the GPU progressive scan decoders (lep_huffprogdec_win.h and lep_huffprogdec.h, lane-loop emulations) against each other and the host parser: a damaged
progressive file is either flagged irregular (-> host parser) or decoded to exactly what the host parser decodes and finished to
the same .lep header.  A tool, run by hand:  python tests/fuzz/fuzz_progressive_decoder.py <seed> <trials>"""
import sys, random, ctypes as C
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import test_core_emulation as t
from conftest import golden, golden_cases
from lepton_amd import abi, corpus
from lepton_amd.codec import JpegImage, LeptonError
emu=C.CDLL(t.EMU_SO)
rnd=random.Random(int(sys.argv[1])); N=int(sys.argv[2])
seeds=[golden(n)[0] for n in golden_cases() if n.startswith('prog_') and 'trunc' not in n and len(golden(n)[0])<30000]
acc=irr=inel=bad=0
for trial in range(N):
    j=bytearray(rnd.choice(seeds))
    sos=j.find(b'\xff\xda')
    k=rnd.choice(['flip','byte','del','ins','hdr'])
    pos=rnd.randrange(sos, len(j)-2) if k!='hdr' else rnd.randrange(2,sos)
    if k=='flip': j[pos]^=1<<rnd.randrange(8)
    elif k=='byte': j[pos]=rnd.randrange(256)
    elif k=='del': del j[pos:pos+rnd.choice([1,2,5])]
    elif k=='ins': j[pos:pos]=bytes(rnd.randrange(256) for _ in range(rnd.choice([1,2])))
    else: j[pos]^=1<<rnd.randrange(8)
    j=bytes(j)
    try:
        rows_w=[]
        h,planes,st=t._progressive_decode_on_the_emulation(emu,j,pipelined=bool(trial&1),win=True,rows_out=rows_w)   # lep_huffprogdec_win.h; level by level / one pipelined launch, in turn
    except AssertionError as e:
        inel+=1; continue   # open_gpu failed / sequential
    if st is None: inel+=1; continue
    # the uniform-code form (lep_huffprogdec.h) on the same bytes: same verdict, same frame, same records
    rows_o=[]
    h2,planes2,st2=t._progressive_decode_on_the_emulation(emu,j,pipelined=bool(trial&1),rows_out=rows_o)
    if st2 is not None: abi.lib().lep_jpeg_close(h2)
    if st2!=st or (st==0 and ([p.raw for p in planes]!=[p.raw for p in planes2] or rows_w!=rows_o)):
        bad+=1; print('FORMS DIFFER',trial,k,st,st2); open('/tmp/pdw_%d.jpg'%trial,'wb').write(j)
    if st==-1: irr+=1; abi.lib().lep_jpeg_close(h); continue
    # accepted by the GPU path: the host parser must accept too and agree
    try:
        t._same_as_the_host_parser(j,h,planes); acc+=1
    except LeptonError as e:
        bad+=1; print('GPU accepted, host refused',trial,k,e.code); open('/tmp/pd_%d.jpg'%trial,'wb').write(j)
    except AssertionError as e:
        bad+=1; print('MISMATCH',trial,k,str(e)[:80]); open('/tmp/pd_%d.jpg'%trial,'wb').write(j)
    abi.lib().lep_jpeg_close(h)
print('accepted+equal',acc,'irregular',irr,'ineligible',inel,'bad',bad)
