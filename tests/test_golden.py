"""The oracle against the committed golden vectors (minted from the reference by tests/golden/make_golden.py).
This test needs neither the reference nor a GPU."""
import hashlib
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_small_lac_files_byte_for_byte(pkg, oracle):
    with open(os.path.join(GOLDEN, "small", "index.json")) as f:
        index = json.load(f)
    assert len(index) >= 10
    for ent in index:
        g = ent["gen"]
        left, right = pkg.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"],
                                          seed=g["seed"], kind=g["kind"], stereo=g["stereo"])
        with open(os.path.join(GOLDEN, "small", ent["name"] + ".lac"), "rb") as f:
            want = f.read()
        assert hashlib.sha256(want).hexdigest() == ent["lac_sha256"]
        got = oracle.encode(left, right, g["sample_rate"], g["bit_depth"], ent["stereo_mode"])
        assert got == want, ent["name"]
        l2, r2, hdr = oracle.decode(want)
        assert np.array_equal(l2, left), ent["name"]
        if right is not None:
            assert np.array_equal(r2, right), ent["name"]
        assert hdr["sample_rate"] == g["sample_rate"] and hdr["bit_depth"] == g["bit_depth"]


def test_digests(pkg, oracle):
    with open(os.path.join(GOLDEN, "digests.json")) as f:
        entries = json.load(f)
    ran = 0
    for ent in entries:
        if not ent.get("cpu_test"):
            continue
        g = ent["gen"]
        left, right = pkg.synth.synth_pcm(g["frames"], g["channels"], g["bit_depth"], g["sample_rate"],
                                          seed=g["seed"], kind=g["kind"], stereo=g["stereo"], start=g.get("start", 0))
        got = oracle.encode(left, right, g["sample_rate"], g["bit_depth"], ent["stereo_mode"], threads=8)
        assert len(got) == ent["lac_bytes"], ent["name"]
        assert hashlib.sha256(got).hexdigest() == ent["lac_sha256"], ent["name"]
        ran += 1
    assert ran >= 3
