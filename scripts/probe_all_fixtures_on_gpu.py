"""Which of the committed fixtures and of the reference's own images take the GPU scan kernels in the batch pipelines, per direction
(gpu_huffman_files of a one-file batch): what is left with the host parser / re-coder, by name."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import golden, golden_cases, ref_cases, ref_golden
from lepton_amd.codec import GpuCodec
c = GpuCodec(0)
files = [(n, golden(n)) for n in golden_cases()] + [("ref:" + n, ref_golden(n)) for n in ref_cases()]
host_c, host_d = [], []
for name, (jpg, lep) in files:
    got, st, cs = c.compress_batch([jpg])
    back, st2, ds = c.decompress_batch([lep])
    ok = st == [0] and got[0] == lep and st2 == [0] and back[0] == jpg
    if not cs["gpu_huffman_files"]: host_c.append(name)
    if not ds["gpu_huffman_files"]: host_d.append(name)
    if not ok: print("NOT THE SAME:", name, st, st2)
print(len(files), "files")
print("compress, host parser:", host_c)
print("decompress, host re-coder:", host_d)
