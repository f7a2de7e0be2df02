#!/usr/bin/env python3
"""Puts the reference's OWN test corpus under tests/golden/ref/ so that it travels to the GPU box (which has no /root/reference):

 * every images/*.jpg of the reference checkout (test DATA: photographs from phones / SLRs / crops / truncations, the inputs of
   Makefile.am:277-353's test_iphone, test_SLR, test_misc, test_truncate, test_odd_rst, test_truncated_zero_run, test_trailing_rst,
   test_progressive, test_arithmetic_failfast, test_bad_zero_run, ... -- no reference SOURCE is copied);
 * for each, the .lep the REAL reference binary (oracle/_ref/lepton, built by oracle/Makefile.ref) writes for it with
   `-unjailed -skipverify` (+ `-allowprogressive` for the progressive ones), the exit code of that run, the exit code / failure name
   of the reference's DEFAULT (verifying) run, and the md5 of what the reference restores from its own .lep;
 * the three known-answer files images/{iphone16,gold-legacy,narrowrst}.lep with the md5s test_suite/test_16threads.sh,
   test_legacy.sh and test_future_compat.sh expect.

Run here (where /root/reference exists): `python tests/golden/make_golden_ref.py`.  Writes tests/golden/ref/manifest.json."""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "ref")
REF = os.path.join(ROOT, "oracle", "_ref", "lepton")
IMAGES = "/root/reference/images"

PROGRESSIVE = {"iphoneprogressive", "iphoneprogressive2", "androidprogressive"}
KNOWN_ANSWERS = {   # test_suite/test_16threads.sh:2, test_legacy.sh:3, test_future_compat.sh:2
    "iphone16.lep": "8ea9fcf1b2c24877aa838dd6ac1df413",
    "gold-legacy.lep": "9ffbfc24d1157d0b1ed7a9b53bef4c23",
    "narrowrst.lep": "07e9021d35114bd69f44f5bc1c3788e3",
}
EXIT_NAMES = {1: "ASSERTION_FAILURE", 3: "SHORT_READ", 8: "PROGRESSIVE_UNSUPPORTED", 41: "ROUNDTRIP_FAILURE", 42: "UNSUPPORTED_JPEG",   # src/vp8/util/memory.hh:13-40
              -6: "ASSERTION_FAILURE (this build of the reference aborts on always_assert)"}


def failure_name(code):
    return EXIT_NAMES.get(code, str(code))


def main():
    os.makedirs(OUT, exist_ok=True)
    manifest = {"jpegs": {}, "known_answers": {}}
    for path in sorted(glob.glob(os.path.join(IMAGES, "*.jpg"))):
        name = os.path.basename(path)[:-4]
        jpg = open(path, "rb").read()
        shutil.copyfile(path, os.path.join(OUT, name + ".jpg"))
        lp = os.path.join(OUT, name + ".lep")
        if os.path.exists(lp):
            os.unlink(lp)
        flags = ["-allowprogressive"] if name in PROGRESSIVE else []
        r = subprocess.run([REF, "-unjailed", "-skipverify"] + flags + [path, lp], capture_output=True)
        entry = {"jpg_md5": hashlib.md5(jpg).hexdigest(), "jpg_size": len(jpg), "progressive": name in PROGRESSIVE,
                 "encode_exit": r.returncode, "encode_failure": failure_name(r.returncode) if r.returncode else None}
        # the verifying run (the reference's default, what test_harness.cc drives): 0, or e.g. 41 ROUNDTRIP_FAILURE
        d = subprocess.run([REF, "-unjailed"] + flags + [path, "/tmp/_golden_ref_default.lep"], capture_output=True)
        entry.update(default_exit=d.returncode, default_failure=failure_name(d.returncode) if d.returncode else None)
        if r.returncode == 0 and os.path.getsize(lp) > 0:
            lep = open(lp, "rb").read()
            back = subprocess.run([REF, "-unjailed", lp, "/tmp/_golden_ref_back.jpg"], capture_output=True)
            restored = open("/tmp/_golden_ref_back.jpg", "rb").read() if back.returncode == 0 else b""
            entry.update(lep_md5=hashlib.md5(lep).hexdigest(), lep_size=len(lep), segments=lep[4], flag=chr(lep[3]), decode_exit=back.returncode,
                         restored_md5=hashlib.md5(restored).hexdigest(), restored_size=len(restored), restored_equals_input=restored == jpg)
        elif os.path.exists(lp):
            os.unlink(lp)
        manifest["jpegs"][name] = entry
        print(name, entry)
    for lep, md5 in KNOWN_ANSWERS.items():
        shutil.copyfile(os.path.join(IMAGES, lep), os.path.join(OUT, "known_" + lep))   # (images/narrowrst.lep must not land on the .lep written for narrowrst.jpg)
        data = open(os.path.join(IMAGES, lep), "rb").read()
        back = subprocess.run([REF, "-unjailed", os.path.join(IMAGES, lep), "/tmp/_golden_ref_back.jpg"], capture_output=True)
        restored = open("/tmp/_golden_ref_back.jpg", "rb").read() if back.returncode == 0 else b""
        assert hashlib.md5(restored).hexdigest() == md5, (lep, back.returncode)
        manifest["known_answers"][lep] = {"restored_md5": md5, "restored_size": len(restored), "lep_size": len(data), "version": data[2], "flag": chr(data[3]), "segments": data[4]}
        print(lep, manifest["known_answers"][lep])
    json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    sys.exit(main())
