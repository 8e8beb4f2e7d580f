"""`bgbinary reference` (make_ref.cpp -> build_reference_task.cpp), host part, on the CPU (--no-bwt): the reference's own
lambda reference directory is reproduced byte for byte from its source.fasta (reference.fasta, karyotype.json,
reference.ref with the md5 of the index), and a hand-made FASTA exercises fasta_ref_importer's rules (N runs below /
from min_n_run, IUPAC codes, lower case, U, scaffolds without bases, CRLF) against a Python restatement."""
import hashlib
import json
import os
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BGBINARY = os.path.join(ROOT, "biograph_amd", "bgbinary")
G = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.skipif(not os.path.exists(BGBINARY), reason="bgbinary not built")


def build(tmp_path, fasta, *extra, ok=True):
    d = str(tmp_path / "ref")
    r = subprocess.run([BGBINARY, "reference", "--in", str(fasta), "--refdir", d, "--no-bwt"] + list(extra), capture_output=True,
                       text=True)
    assert (r.returncode == 0) == ok, r.stderr
    return d, r


def test_lambda_reference_directory_byte_for_byte(tmp_path):
    d, r = build(tmp_path, os.path.join(G, "ref_lambda_full", "source.fasta"))
    assert "Results saved to" in r.stdout
    for name, gold in (("reference.fasta", "ref_lambda_full/reference.fasta"), ("karyotype.json", "ref_lambda/karyotype.json"),
                       ("reference.ref", "ref_lambda/reference.ref"), ("source.fasta", "ref_lambda_full/source.fasta")):
        assert open(os.path.join(d, name), "rb").read() == open(os.path.join(G, gold), "rb").read(), name
    # overwrite rules of make_ref.cpp:93-104
    d2, r2 = build(tmp_path, os.path.join(G, "ref_lambda_full", "source.fasta"), ok=False)
    assert "Refusing to overwrite" in r2.stderr
    build(tmp_path, os.path.join(G, "ref_lambda_full", "source.fasta"), "--force")
    os.makedirs(str(tmp_path / "other"))
    r3 = subprocess.run([BGBINARY, "reference", "--in", os.path.join(G, "ref_lambda_full", "source.fasta"), "--refdir",
                         str(tmp_path / "other"), "--force", "--no-bwt"], capture_output=True, text=True)
    assert r3.returncode != 0 and "is not a BioGraph reference" in r3.stderr


def restate(fasta_text, min_n_run=50):
    """fasta_ref_importer.cpp:30-167 -> (karyotype dict, reference.fasta text, [(name, seq)] of the flat records)."""
    translate = "ANCN..GN..N.NN...NNTTNNNN."
    order, scaffolds, data = [], {}, []
    st = {"name": "", "seq": "", "pos": 0, "start": 0, "end": 0, "had": False}

    def finish_supercontig():
        if not st["seq"]:
            st["start"] = st["end"] = st["pos"]
            return
        st["had"] = True
        idx = order.index(st["name"]) if st["name"] in order else len(order)
        data.append((idx, st["name"], st["seq"], st["start"]))
        st["start"] = st["end"] = st["pos"]
        st["seq"] = ""

    def finish_scaffold():
        if st["name"]:
            finish_supercontig()
            if st["had"]:
                if st["name"] not in order:
                    order.append(st["name"])
                scaffolds.setdefault(st["name"], (st["pos"], order.index(st["name"])))
        st["pos"] = st["start"] = st["end"] = 0
        st["had"] = False

    for line in fasta_text.replace("\r\n", "\n").split("\n"):
        if not line:
            continue
        if line[0] == ">":
            finish_scaffold()
            st["name"] = line[1:].split()[0]
            continue
        for ch in line:
            c = translate[ord(ch.upper()) - 65]
            assert c != "."
            if c != "N":
                if st["end"] != st["pos"]:
                    if st["pos"] - st["end"] < min_n_run:
                        st["seq"] += c * (st["pos"] - st["end"])
                        st["end"] = st["pos"]
                    else:
                        finish_supercontig()
                st["seq"] += c
                st["end"] += 1
            st["pos"] += 1
    finish_scaffold()
    first = {}
    for _, n, q, s in data:
        first.setdefault((n, s), len(q))  # std::set<supercontig>::insert: the first one of a (name, offset) stays
    scs = sorted(first.items())
    kt = {"chr_order": order,
          "chromosomes": [{"index": scaffolds[n][1], "len": scaffolds[n][0], "name": n} for n in sorted(scaffolds)],
          "supercontigs": [{"chr": n, "len": ln, "name": "%s:%d" % (n, s), "offset": s} for (n, s), ln in scs]}
    data.sort(key=lambda d: (d[0], d[3]))
    fa = ""
    for _, n, q, s in data:
        fa += ">%s:%d" % (n, s)
        for j in range(0, len(q), 80):
            fa += "\n" + q[j:j + 80]
        fa += "\n"
    return kt, fa, [("%s:%d" % (n, s), q) for _, n, q, s in data]


def parse_ref(blob):
    assert blob[:6] == b"SPR000"
    n = struct.unpack(">I", blob[-4:])[0]
    return blob[6:len(blob) - 4 - n], blob[len(blob) - 4 - n:len(blob) - 4]


def test_importer_rules_on_a_hand_made_fasta(tmp_path):
    import random
    rnd = random.Random(5)
    s = lambda n: "".join(rnd.choice("ACGT") for _ in range(n))
    text = (">chrB some description\n" + s(70) + "\n" + "N" * 49 + s(30) + "\n" + "n" * 50 + s(90) + "rykmswbdhvU" + s(12) + "\n" +
            ">empty only N\n" + "N" * 120 + "\n" +
            ">chrA\r\n" + "NNN" + s(161).lower() + "\r\n" + "N" * 200 + "\r\n" + s(5) + "\r\n")
    src = tmp_path / "in.fa"
    src.write_text(text)
    d, _ = build(tmp_path, src, "--min-n-run", "50")
    kt, fa, recs = restate(text, 50)
    assert json.load(open(os.path.join(d, "karyotype.json"))) == kt
    assert open(os.path.join(d, "karyotype.json")).read() == json.dumps(kt, separators=(",", ":"), sort_keys=True)
    assert open(os.path.join(d, "reference.fasta")).read() == fa
    assert "empty" not in kt["chr_order"] and kt["chr_order"] == ["chrB", "chrA"]
    assert [c["name"] for c in kt["supercontigs"]] == ["chrA:0", "chrA:364", "chrB:0", "chrB:199"]  # 3 leading N: filled
    # reference.ref: 2 bits per base in reference.fasta order, padded with A; index = names, md5s, sizes, extents
    packed, index = parse_ref(open(os.path.join(d, "reference.ref"), "rb").read())
    flat = "".join(q for _, q in recs)
    bits = "".join("ACGT"[(b >> sh) & 3] for b in packed for sh in (6, 4, 2, 0))
    assert bits[:len(flat)] == flat and set(bits[len(flat):]) <= {"A"} and len(packed) == (len(flat) + 3) // 4
    for name, q in recs:
        assert name.encode() in index and hashlib.md5(q.encode()).hexdigest().encode() in index
    # the library's own reader agrees
    import biograph_amd as bg
    ref = bg.Reference(d)
    assert ref.scaffolds == ["chrB", "chrA"]
    got = {(name, a): ref.sequence(name, a, b) for name, a, b in ref.supercontigs}  # (scaffold, start, end)
    assert got == {(n.rsplit(":", 1)[0], int(n.rsplit(":", 1)[1])): q for n, q in recs}
    # a different min_n_run moves the cuts
    d2 = str(tmp_path / "ref2")
    subprocess.run([BGBINARY, "reference", "--in", str(src), "--refdir", d2, "--no-bwt", "--min-n-run", "10"], check=True,
                   capture_output=True)
    assert json.load(open(os.path.join(d2, "karyotype.json"))) == restate(text, 10)[0]


def test_rejects_bad_fasta(tmp_path):
    for bad in (">x\nAC-GT\n", ">x\nACGJ\n", "ACGT\n"):
        src = tmp_path / "bad.fa"
        src.write_text(bad)
        d, r = build(tmp_path, src, ok=False)
        assert r.stderr.strip()
        subprocess.run(["rm", "-rf", d])
