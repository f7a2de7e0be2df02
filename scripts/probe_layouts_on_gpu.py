import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import golden, ref_golden
from lepton_amd.codec import GpuCodec
c = GpuCodec(0)
files = [("grayscale", ref_golden("grayscale")), ("gray2sf", ref_golden("gray2sf")), ("gray_120x88", golden("gray_120x88")), ("lay_gray22_80x56", golden("lay_gray22_80x56")),
         ("seq_y_cbcr_420_rst_97x50", golden("seq_y_cbcr_420_rst_97x50")), ("seq_ycb_cr_422_640x480_2seg", golden("seq_ycb_cr_422_640x480_2seg")), ("rst_rows_gray_64x96", golden("rst_rows_gray_64x96"))]
for name, (jpg, lep) in files:
    got, st, cs = c.compress_batch([jpg])
    back, st2, ds = c.decompress_batch([lep])
    print("%-32s compress st %s same %s gpu_huffman %d | decompress st %s same %s gpu_huffman %d" % (name, st, got[0] == lep, cs["gpu_huffman_files"], st2, back[0] == jpg, ds["gpu_huffman_files"]))
