"""GPU tests written after a round's last hardware visit go into this file, which sorts last: under `pytest -x` they cannot hide
the results of the tests that have been through hardware.  (The one below has since passed on the MI355X:
profiles/r03d_pytest_gpu_final.log, 113 of 113.)"""
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


def test_gpu_round_trip_failures_and_non_canonical_scans_are_refused(gpu_codec):
    """the reference's default run (verification on) on files it cannot restore: exit 41 -- here lep_compress refuses them, and
    lep_compress_batch with verify=1 does (without verification the .lep is the reference's -skipverify one, and decodes to
    the same wrong bytes); legal but non-canonical Huffman layers (ZRL + EOB, mixed pad bits, bytes left over behind the last MCU: given up by the GPU scan
    decoder, refused by the host parser it falls back to) end in UNSUPPORTED_JPEG; their neighbours in the batch are untouched"""
    import hashlib
    import numpy as np
    import jpeg_writer as jw
    from conftest import roundtrip_failure_cases
    from lepton_amd.codec import LeptonError

    (name, restored_md5), = roundtrip_failure_cases()
    bad, bad_lep = golden(name)
    with pytest.raises(LeptonError) as e:
        gpu_codec.compress(bad)
    assert e.value.code == 41
    assert hashlib.md5(gpu_codec.decompress(bad_lep)).hexdigest() == restored_md5
    comps = [(0, 2, 2, 0, 0, 0), (200, 1, 1, 1, 1, 1), (7, 1, 1, 1, 1, 1)]
    quirky = [jw.write_baseline(160, 96, comps, np.random.default_rng(4), restart_interval=4, quirks=(q,))[0] for q in ("trailing_zrl", "mixed_pad", "scan_tail")]
    good = [golden("lay_mixed_200x120"), golden("c420_160x120"), golden("lay_gray22_80x56")]
    jpgs = [good[0][0], bad, quirky[0], good[1][0], quirky[1], quirky[2], good[2][0]]
    for verify in (True, False):
        got, status, _ = gpu_codec.compress_batch(jpgs, verify=verify)
        assert status == [0, 41 if verify else 0, 42, 0, 42, 42, 0]
        assert got[0] == good[0][1] and got[3] == good[1][1] and got[6] == good[2][1]
        assert got[1] == (None if verify else bad_lep) and got[2] is None and got[4] is None and got[5] is None
    for q in quirky:
        with pytest.raises(LeptonError) as e:
            gpu_codec.compress(q)
        assert e.value.code == 42
