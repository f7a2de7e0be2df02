#!/usr/bin/env python3
"""Format-version-2 fixtures (tests/golden/v2/): what the REAL reference (oracle/_ref/lepton) writes with `-brotliheader`
for a few of the golden JPEGs, chained streams (`cat a.lep b.lep`, test_suite/test_concat.sh) and a `-lepcat` file with
its merged "CNT" header (src/lepton/concat.cc), each with the bytes the reference restores from it; plus a copy of the
reference's own known-answer file images/narrowrst.lep (format version 4, test_suite/test_future_compat.sh: its decode must
hash to 07e9021d35114bd69f44f5bc1c3788e3).  Run where /root/reference exists; the fixtures travel to the GPU box."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.path.join(ROOT, "oracle", "_ref", "lepton")
OUT = os.path.join(HERE, "v2")
SINGLES = ["c420_160x120", "gray_120x88", "rst_c420_176x112", "q30_256x256_4seg", "truncated", "trailing_garbage", "prog_c420_320x240", "lay_mixed_rst_104x72"]
CHAINS = {"chain_2": ["c420_160x120", "rst_c420_176x112"], "chain_3": ["q30_256x256_4seg", "gray_120x88", "trailing_garbage"],
          "chain_prog": ["prog_c420_320x240", "c420_160x120"]}
CONCATS = {"concat_2": ["c420_160x120", "rst_c420_176x112"], "concat_3": ["gray_120x88", "c420_160x120", "lay_mixed_rst_104x72"]}


def run(args, **kw):
    return subprocess.run([REF, "-unjailed"] + args, capture_output=True, **kw)


def main():
    os.makedirs(OUT, exist_ok=True)
    man = {}
    tmp = tempfile.mkdtemp()
    leps = {}
    for n in SINGLES:
        jp = os.path.join(HERE, n + ".jpg")
        lp = os.path.join(OUT, n + ".lep")
        flags = ["-brotliheader", "-skipverify"] + ([] if n.startswith("prog") else ["-rejectprogressive"])
        r = run(flags + [jp, lp])
        assert r.returncode == 0, (n, r.returncode, r.stderr[-300:])
        back = os.path.join(tmp, n + ".back")
        r = run([lp, back])
        assert r.returncode == 0, (n, r.returncode)
        leps[n] = open(lp, "rb").read()
        man[n] = {"version": leps[n][2], "restored_md5": hashlib.md5(open(back, "rb").read()).hexdigest(),
                  "restored_equals_input": open(back, "rb").read() == open(jp, "rb").read()}
    for name, parts in CHAINS.items():
        blob = b"".join(leps[p] for p in parts)
        lp = os.path.join(OUT, name + ".lep")
        open(lp, "wb").write(blob)
        r = subprocess.run([REF, "-unjailed", "-"], input=blob, capture_output=True)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-300:])
        open(os.path.join(OUT, name + ".restored"), "wb").write(r.stdout)
        man[name] = {"chain": parts, "restored_md5": hashlib.md5(r.stdout).hexdigest()}
    for name, parts in CONCATS.items():
        lp = os.path.join(OUT, name + ".lep")
        with open(lp, "wb") as f:
            r = subprocess.run([REF, "-unjailed", "-lepcat"] + [os.path.join(OUT, p + ".lep") for p in parts], stdout=f, stderr=subprocess.PIPE)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-300:])
        r = subprocess.run([REF, "-unjailed", "-"], input=open(lp, "rb").read(), capture_output=True)
        assert r.returncode == 0, (name, r.returncode, r.stderr[-300:])
        open(os.path.join(OUT, name + ".restored"), "wb").write(r.stdout)
        man[name] = {"concatenate": parts, "restored_md5": hashlib.md5(r.stdout).hexdigest()}
    src = "/root/reference/images/narrowrst.lep"
    shutil.copy(src, os.path.join(OUT, "narrowrst.lep"))
    r = subprocess.run([REF, "-unjailed", "-"], input=open(src, "rb").read(), capture_output=True)
    assert hashlib.md5(r.stdout).hexdigest() == "07e9021d35114bd69f44f5bc1c3788e3"
    man["narrowrst"] = {"version": open(src, "rb").read()[2], "restored_md5": "07e9021d35114bd69f44f5bc1c3788e3",
                        "source": "dropbox/lepton images/narrowrst.lep (known-answer vector of test_suite/test_future_compat.sh)"}
    json.dump(man, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(man, indent=1))


if __name__ == "__main__":
    main()
