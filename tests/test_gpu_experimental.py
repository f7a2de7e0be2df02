"""Opt-in code paths that have been checked in the lane-loop emulation but not yet on hardware.  NOT part of `-m gpu`:
run by hand, under `timeout`, from scripts/gpu_huffpar.sh; a test moves to test_gpu_parity.py once it has passed there."""
import pytest

from conftest import golden, golden_cases
from lepton_amd import corpus


@pytest.mark.gpu_experimental
@pytest.mark.parametrize("nsub", ["4", "16"])
def test_parallel_huffman_decode_in_the_compress_pipeline(gpu_codec, monkeypatch, nsub):
    """LEP_HUFFDEC_PAR=<n>: n wavefronts per image decode the JPEG scan (lep_huffdec_par.h) -- same .lep bytes"""
    names = golden_cases()
    jpgs = [golden(n)[0] for n in names] + [corpus.synth_jpeg(1280, 720, 61), corpus.synth_jpeg(640, 480, 62, quality=97)]
    want, st0, _ = gpu_codec.compress_batch(jpgs)
    monkeypatch.setenv("LEP_HUFFDEC_PAR", nsub)
    got, st1, _ = gpu_codec.compress_batch(jpgs)
    assert st0 == st1 and got == want
    assert got[: len(names)] == [golden(n)[1] for n in names]
