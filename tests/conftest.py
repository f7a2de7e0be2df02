import glob
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_IMAGES = "/root/reference/images"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    import __graft_entry__ as g

    g.build()


def golden_cases():
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return sorted(k for k, v in man.items() if v.get("encode_exit") == 0 and "slice" not in v and "embedding" not in v and "permissive" not in v and "roundtrip_failure" not in v)


def roundtrip_failure_cases():
    """files the reference only compresses with -skipverify (its default run: ROUNDTRIP_FAILURE): [(name, md5 of what the reference restores)]"""
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return sorted((k, v["restored_md5"]) for k, v in man.items() if v.get("roundtrip_failure"))


def embedded_cases():
    """`lepton -embedding=<n>` fixtures: [(name, n)]; the .jpg is the whole blob, which the .lep restores"""
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return sorted((k, v["embedding"]) for k, v in man.items() if v.get("encode_exit") == 0 and "embedding" in v)


def slice_cases():
    """`lepton -startbyte -trunc` fixtures: [(name, start_byte, trunc)]; the .jpg is the whole input, the .lep restores [start, trunc)"""
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    return sorted((k, v["slice"][0], v["slice"][1]) for k, v in man.items() if v.get("encode_exit") == 0 and "slice" in v and v.get("restored_equals_input"))


def golden(name):
    return (open(os.path.join(GOLDEN, name + ".jpg"), "rb").read(), open(os.path.join(GOLDEN, name + ".lep"), "rb").read())


def reference_jpegs():
    """baseline fixtures of the reference's own test-suite (present only in the build container)"""
    skip = {"arithmetic", "badzerorun"}
    return sorted(p for p in glob.glob(os.path.join(REF_IMAGES, "*.jpg")) if os.path.basename(p)[:-4] not in skip)


@pytest.fixture(scope="session")
def gpu_codec():
    from lepton_amd.codec import GpuCodec

    return GpuCodec(0)


REF_GOLDEN = os.path.join(GOLDEN, "ref")   # the reference's own images/ + what its binary writes for them (tests/golden/make_golden_ref.py)


def ref_manifest():
    return json.load(open(os.path.join(REF_GOLDEN, "manifest.json")))


def ref_cases(progressive=None):
    """the reference's own images/*.jpg that its binary compresses (exit 0 with -skipverify): names"""
    return sorted(k for k, v in ref_manifest()["jpegs"].items() if v["encode_exit"] == 0 and (progressive is None or v["progressive"] == progressive))


def ref_refused_cases():
    """[(name, exit code)] of the images the reference refuses (arithmetic.jpg: 42; badzerorun.jpg: an assertion)"""
    return sorted((k, v["encode_exit"]) for k, v in ref_manifest()["jpegs"].items() if v["encode_exit"] != 0)


def ref_golden(name):
    return (open(os.path.join(REF_GOLDEN, name + ".jpg"), "rb").read(), open(os.path.join(REF_GOLDEN, name + ".lep"), "rb").read())
