"""Format versions >= 2 of the .lep container: brotli-compressed headers (`lepton -brotliheader`,
jpgcoder.cc:1116-1119, 4038, 4172), the packet end marker, every thread segment bound by its size (recoder.cc:598-613),
chained streams (`cat a.lep b.lep | lepton -`, jpgcoder.cc:1868-1897, test_suite/test_concat.sh) and `-lepcat` files with a
merged header (concat.cc).  Fixtures: tests/golden/v2/, written by the real reference (make_golden_v2.py), plus the
reference's own known-answer file narrowrst.lep (test_suite/test_future_compat.sh).  Decode direction only: the encoder
writes version 1, like the reference's default."""
import hashlib
import json
import os

import pytest

import oracle_binding as ob
from conftest import GOLDEN, golden
from lepton_amd.codec import LepFile, LeptonError, lep_stream

V2 = os.path.join(GOLDEN, "v2")
MAN = json.load(open(os.path.join(V2, "manifest.json")))


def v2(name):
    return open(os.path.join(V2, name + ".lep"), "rb").read()


def restore_on_the_cpu(blob):
    out = b""
    for f in lep_stream(blob):
        ob.oracle_decode(f.desc, f.segments, f.streams)
        out += f.recode()
    return out


@pytest.mark.parametrize("name", sorted(MAN))
def test_v2_files_restore_what_the_reference_restores(name):
    got = restore_on_the_cpu(v2(name))
    assert hashlib.md5(got).hexdigest() == MAN[name]["restored_md5"]
    if MAN[name].get("restored_equals_input"):
        assert got == golden(name)[0]
    for key in ("chain", "concatenate"):
        if key in MAN[name] and all(MAN[p]["restored_equals_input"] for p in MAN[name][key]):
            assert got == b"".join(golden(p)[0] for p in MAN[name][key])


def test_v2_container_details():
    f = LepFile(v2("c420_160x120"))
    assert f.data[2] == 2 and not f.more and f.consumed == len(f.data)
    files = lep_stream(v2("chain_3"))
    assert len(files) == 3 and [x.consumed for x in files[:2]] == [len(v2(n)) for n in MAN["chain_3"]["chain"][:2]]
    assert len(lep_stream(v2("concat_3"))) == 3
    assert LepFile(v2("narrowrst")).data[2] == 4
    # a version-1 file has no end marker: whatever follows it belongs to it (the reference decodes the first file only)
    one = golden("c420_160x120")[1]
    assert not LepFile(one + one).more
    # ANS-coded streams (version 3) need a build option the default reference does not have either
    ans = bytearray(v2("c420_160x120")); ans[2] = 3
    with pytest.raises(LeptonError) as e:
        LepFile(bytes(ans))
    assert e.value.code == 1
    bad = bytearray(v2("c420_160x120")); bad[2] = 9
    with pytest.raises(LeptonError) as e:
        LepFile(bytes(bad))
    assert e.value.code == 13
    # a chained stream cut inside the second file's fixed header is SHORT_READ for the whole stream, as in the reference;
    # cut inside its packets it restores what the reference restores (the coder reads zero bits past the end)
    with pytest.raises(LeptonError) as e:
        lep_stream(v2("chain_2")[: len(v2("c420_160x120")) + 10])
    assert e.value.code == 3
    assert hashlib.md5(restore_on_the_cpu(v2("chain_2")[:-2000])).hexdigest() == "30b733935e1187832bffafbf8c68ee0a"


@pytest.mark.gpu
def test_gpu_v2_files_and_chained_streams(gpu_codec):
    names = sorted(MAN)
    for n in names:
        assert hashlib.md5(gpu_codec.decompress(v2(n))).hexdigest() == MAN[n]["restored_md5"], n
    back, st, _ = gpu_codec.decompress_batch([v2(n) for n in names] + [golden("c420_160x120")[1]])
    assert st == [0] * (len(names) + 1)
    assert [hashlib.md5(b).hexdigest() for b in back[:-1]] == [MAN[n]["restored_md5"] for n in names]
    assert back[-1] == golden("c420_160x120")[0]


def _write_v2(name, streams_of):
    from lepton_amd import abi
    from lepton_amd.codec import JpegImage

    L = abi.lib()
    img = JpegImage(golden(name)[0])
    assert L.lep_jpeg_set_container_version(img.handle, 2) == 0
    return img.write_lep(streams_of(img))


WRITTEN_BY_THE_REFERENCE = [n for n in sorted(MAN) if not n.startswith(("chain", "concat", "narrowrst"))]


@pytest.mark.parametrize("name", WRITTEN_BY_THE_REFERENCE)
def test_v2_writer_equals_the_reference(name):
    """`lepton -brotliheader` (jpgcoder.cc:1116-1119, 4038): format-2 files written here -- the header through the reference's own
    brotli 1.0.0 encoder with BrotliCodec::Compress's parameters (src/io/BrotliCompression.cc:45-98), the packets closed with
    FF FE FF -- are byte-equal to the files the reference binary wrote from the same JPEGs (tests/golden/v2, make_golden_v2.py)"""
    from lepton_amd import abi

    if not abi.lib().lep_container_can_write_version(2):
        pytest.skip("this build of the library has no brotli 1.0.0 encoder (build() compiles it where /root/reference is)")
    assert _write_v2(name, lambda img: ob.oracle_encode(img.desc, img.plan())[0]) == v2(name)


def test_v2_writer_refuses_what_it_cannot_write():
    from lepton_amd import abi
    from lepton_amd.codec import JpegImage

    L = abi.lib()
    img = JpegImage(golden("c420_160x120")[0])
    assert L.lep_jpeg_set_container_version(img.handle, 3) == 13      # ANS coding: VERSION_UNSUPPORTED, like the default reference build
    assert L.lep_jpeg_set_container_version(img.handle, 1) == 0
    assert L.lep_container_can_write_version(1) == 1 and L.lep_container_can_write_version(4) == 0


@pytest.mark.gpu
def test_gpu_v2_writer_equals_the_reference(gpu_codec):
    from lepton_amd import abi

    if not abi.lib().lep_container_can_write_version(2):
        pytest.skip("no brotli 1.0.0 encoder in this build")
    for name in WRITTEN_BY_THE_REFERENCE:
        got = _write_v2(name, lambda img: gpu_codec.encode([img], [img.plan()])[0])
        assert got == v2(name), name
        assert gpu_codec.decompress(got) == golden(name)[0]
